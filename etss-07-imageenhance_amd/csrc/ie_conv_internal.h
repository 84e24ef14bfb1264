// Shared between ie_ops.hip (forward) and ie_train.hip (data gradient = a forward conv with transposed, flipped weights).
#pragma once
#include "ie_common.h"

struct ConvNParams {
    const float* src[IE_CONV_MAX_SRC];
    int cs[IE_CONV_MAX_SRC];
    int nsrc;
    const float* w; const float* bias; const float* res; float* dst;
    int N, H, W, Cin, Cout, kh, kw, stride, pad, pad_mode, act, Ho, Wo, clamp;
    float slope, lo, hi;
    int w_ld;          // row pitch of the HWIO weight matrix (>= Cout; > Cout when only a column slice is used)
    int accumulate;    // dst += result (data gradients landing on a tensor with several consumers)
};

// Launch the fp32 convolution described by p (fields Ho/Wo already set): LDS-tiled GEMM kernel for wide layers,
// direct kernel for thin ones.  Returns an ie_status.
int ie_convn_launch(const ConvNParams& p, hipStream_t stream);

// Matrix-core weight gradient (ie_wgrad_mfma.hip): 16-bit NHWC, 3x3 / stride 1 / pad 1 / zero padding or 1x1 / pad 0, channel counts
// multiples of 8.
bool ie_wgrad_mfma_applicable(int kh, int kw, int stride, int pad, int pad_mode, int cs, int cout);
int ie_wgrad_mfma_launch(const void* x, const void* dy, int dtype, float* dw_hwio, int N, int H, int W, int cs, int cout, int cin_total, int cbase,
                         int ksize, hipStream_t stream);
// deterministic mode (ie_core.hip): library-owned scratch of at least `bytes` for the split-K partial slabs
extern "C" int ie_get_deterministic(void);
int ie_det_workspace(size_t bytes, void** out);
