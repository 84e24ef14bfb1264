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
                         int ksize, int pad, int stride, int in_h, int in_w, hipStream_t stream);
bool ie_wgrad_mfma_applicable_blocks(int kh, int kw, int stride, int pad, int pad_mode, int cs, int cout);      // 9x9 / pad 0 and 4x4 / stride 2 / pad 1 as launches of the 3x3 form
// deterministic mode (ie_core.hip): library-owned scratch of at least `bytes` for the split-K partial slabs
extern "C" int ie_get_deterministic(void);
int ie_det_workspace(size_t bytes, void** out);

// Strip form of the narrow 3x3 layers (ie_conv_strip.hip): Cin 32 / 64 / 32+32, Cout 32 / 64, stride 1, zero / replicate padding 1, bias + ReLU / LeakyReLU / none,
// reading ie_mconv_pack_weights' image.  ie_mconv_nhwc asks `applicable` first.
bool ie_sconv3_applicable(int nsrc, const int* cs, int Cout, int kh, int kw, int stride, int pad, int pad_mode, int act, bool residual, bool clamp,
                          bool out_f32, int W);
int ie_sconv3_first_launch(const void* img8, const void* w1, const float* b1, const void* wpk2, const float* b2, void* dst, void* pool_dst, int dtype,
                           int N, int H, int W, int C, hipStream_t st);
struct ie_sconv_tail {      // 1x1 conv of the ReLU'd rows, fused into the strip kernel (the conv's own output is not written)
    const void* w;          // [16][Cout] 16-bit, rows >= cout zero
    const float* b;         // [16]
    float* dst;             // NCHW fp32 [N, cout, H, W]
    int cout, clamp;
    float lo, hi;
};
bool ie_sconv3_fused_ok(int nsrc, const int* cs);
int ie_sconv3_launch(const void* const* srcs, const int* cs, int nsrc, const void* wpk, const float* bias, void* dst, int dtype, int N, int H, int W,
                     int Cout, int act, float slope, int pad_mode, const int* src_h, const int* src_w, void* pool_dst, const ie_sconv_tail* tail, int stride, hipStream_t st);
// pool_dst: null, or the 2x2 max-pooled copy of dst written by the same launch (ReLU layers only)
// 3x3 conv over <= 3 sources of 64 channels (each at its own resolution, nearest up-sampling folded in) with 1 .. 4 output channels, fp32 out (ie_conv_thin3.hip)
bool ie_sconv3_thin3_applicable(int nsrc, const int* cs, int Cout, int kh, int kw, int stride, int pad, int pad_mode, int act, bool residual, bool clamp,
                                bool out_f32, bool shuffle, int W);
int ie_sconv3_thin3_launch(const void* const* srcs, int nsrc, const int* src_h, const int* src_w, const void* wpk, const float* bias, float* dst, int dtype, int N,
                           int H, int W, int Cout, int act, int pad_mode, hipStream_t st);
// 1x1 convolution as a streaming GEMM over 128-pixel rows (ie_conv1x1.hip)
bool ie_conv1x1_applicable(int nsrc, const int* cs, int Cout, int kh, int kw, int stride, int pad, int act, bool residual, bool clamp, bool out_f32,
                           bool pixel_shuffle, bool any_up);
int ie_conv1x1_launch(const void* const* srcs, const int* cs, int nsrc, const void* packed_w, const float* bias, void* dst, int dtype, long long npx, int Cout, int nf,
                      int cout_tiles, int act, float slope, hipStream_t st);
// two stacked Conv3x3(64 -> 64) + bias + ReLU layers, the intermediate in LDS only (ie_conv_pair.hip)
int ie_sconv3x2_launch(const void* src, const void* w_a, const float* bias_a, const void* w_b, const float* bias_b, void* dst, int dtype, int N, int H, int W,
                       int pad_mode, int cout_b, int act_b, void* dst16, int dst16_c, void* pool_dst, hipStream_t st);
