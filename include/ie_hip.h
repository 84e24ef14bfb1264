/* ie_hip.h -- C ABI of the MI355X (gfx950) image-enhancement hot path.
 *
 * Drop-in boundary: the reference is pure Python and its hot
 * path bottoms out in ATen calls (torch.nn.Conv2d / F.conv2d / tanh / pow ...)
 * made from `forward(datapoint)`; there is no reference FFI.  The entry points
 * below are what a reference-side binding (ctypes / cffi)
 * calls instead of those ATen ops.  Each declaration cites the reference
 * interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - plain C types only; every pointer named *_dev is a device pointer owned
 *     by the caller (e.g. torch.Tensor.data_ptr()); *_host pointers are host
 *     memory read during the call.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream). All
 *     work is enqueued on it; no entry point synchronises the device except
 *     ie_plan_destroy / ie_zdce_reserve (allocation).
 *   - return value: 0 (IE_OK) or a negative ie_status; ie_last_error() gives a
 *     thread-local message.  Nothing throws or aborts.
 *   - activations between layers are NHWC ("channels last"), element type
 *     ie_dtype; images at the boundary are NCHW fp32 in [0,1] like the
 *     reference's datapoint["image"].
 */
#ifndef IE_HIP_H
#define IE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a public struct, enum or the meaning of an existing argument changes (2: ie_mconv_desc gained src_h / src_w /
 * pixel_shuffle).  The Python loader refuses a library whose ie_abi_version() differs from the one it was written against. */
#define IE_ABI_VERSION 2

typedef enum ie_status {
    IE_OK = 0,
    IE_ERR_INVALID = -1,     /* bad argument / shape */
    IE_ERR_HIP = -2,         /* a HIP runtime call failed */
    IE_ERR_UNSUPPORTED = -3, /* valid request this build has no kernel for */
    IE_ERR_STATE = -4,       /* plan not finalised / weight missing */
    IE_ERR_NOMEM = -5
} ie_status;

typedef enum ie_dtype { IE_F32 = 0, IE_F16 = 1, IE_BF16 = 2, IE_U8 = 3 } ie_dtype;
typedef enum ie_act { IE_ACT_NONE = 0, IE_ACT_RELU = 1, IE_ACT_LRELU = 2, IE_ACT_TANH = 3, IE_ACT_SIGMOID = 4 } ie_act;
typedef enum ie_pad_mode { IE_PAD_ZEROS = 0, IE_PAD_REPLICATE = 1 } ie_pad_mode;

/* ---- library ---------------------------------------------------------- */
int ie_abi_version(void);
const char* ie_last_error(void);
/* Fills name (<= cap bytes) with the gcnArchName of `device`; *cu = CU count. */
int ie_device_info(int device, char* name, int cap, int* cu, size_t* hbm_bytes);

/* ---- layout / boundary ops -------------------------------------------- */
/* NCHW fp32 -> NHWC dtype, channels zero-padded C -> Cp.
 * Replaces the implicit layout of datapoint["image"] (vision/model.py:97-109). */
int ie_nchw_f32_to_nhwc(const float* src_dev, void* dst_dev, ie_dtype dt, int N, int C, int H, int W, int Cp, void* stream);
/* NHWC dtype (Cp channels, first C used) -> NCHW fp32. */
int ie_nhwc_to_nchw_f32(const void* src_dev, ie_dtype dt, float* dst_dev, int N, int C, int H, int W, int Cp, void* stream);
/* float NCHW [0,1] -> uint8 NHWC.  mode 0: torchvision save_image rule
 * clamp(x*255+0.5,0,255) (core/image/io.py:139-140); mode 1: truncating
 * (uint8)(x*255) of the clamped value (core/image/utils.py:795). */
int ie_nchw_f32_to_u8_nhwc(const float* src_dev, uint8_t* dst_dev, int N, int C, int H, int W, int mode, void* stream);
/* uint8 NHWC -> float NCHW /255 (core/image/utils.py:801-847 to_image_tensor(normalize)). */
int ie_u8_nhwc_to_nchw_f32(const uint8_t* src_dev, float* dst_dev, int N, int C, int H, int W, void* stream);

/* Bilinear resize NCHW fp32 with F.interpolate semantics.  align_corners=0 is what
 * core.resize -> kornia resize does inside VisionModel.infer (vision/model.py:104-106,119-123;
 * core/image/geometry/transform.py:216-223); align_corners=1 is nn.Upsample(bilinear,
 * align_corners=True) (llunetpp.py:182, unet.py:82). */
int ie_resize_bilinear_nchw_f32(const float* src_dev, float* dst_dev, int N, int C, int Hi, int Wi, int Ho, int Wo,
                                int align_corners, void* stream);

/* ---- generic conv2d (fp32, NHWC) -------------------------------------- */
/* Replaces torch.nn.Conv2d / F.conv2d as aliased by mon.nn.Conv2d
 * (src/mon/nn/modules/conv/base.py:45) for every shape on the path:
 * k in {1,3,4,9}, stride {1,2}, zero or replicate padding
 * (mon_extra/.../retinexnet/model.py:26-66), groups in {1, Cin} (depthwise,
 * nn/modules/conv/dsconv.py:122-134), fused bias + activation, and a virtual
 * channel concat of two sources (torch.cat(dim=1), zero_dce.py:180-182).
 * Weights: HWIO fp32 [kh][kw][Cin/groups][Cout] on the device (repack with
 * ie_pack_weight_hwio_f32 from the checkpoint's OIHW). */
typedef struct ie_conv2d_desc {
    int N, H, W;        /* input spatial size */
    int C0, C1;         /* channels of source 0 / source 1 (C1 = 0: single source) */
    int Cout;
    int kh, kw, stride, pad;
    int pad_mode;       /* ie_pad_mode */
    int groups;         /* 1 or C0 (depthwise; C1 must be 0) */
    int act;            /* ie_act */
    float slope;        /* LeakyReLU slope */
    int Ho, Wo;         /* out: filled by the call */
} ie_conv2d_desc;
int ie_conv2d_nhwc_f32(ie_conv2d_desc* d, const float* src0_dev, const float* src1_dev,
                       const float* w_hwio_dev, const float* bias_dev, float* dst_dev, void* stream);
/* OIHW fp32 host -> HWIO fp32 device. */
int ie_pack_weight_hwio_f32(const float* w_oihw_host, int Cout, int Cin, int kh, int kw, float* dst_dev, void* stream);

/* ---- generator building blocks, fp32 NHWC (parity path of U-Net / LLUnet++ / RetinexNet / Zero-DCE++) ---- */
#define IE_CONV_MAX_SRC 8
/* conv2d over the virtual channel concat of nsrc tensors (torch.cat(dim=1): llunetpp.py:224-239 concatenates
 * up to 5 maps; unet.py:97; retinexnet/model.py:79-86) with fused bias + activation, then optional residual
 * add (llunetpp.py:134 `x3 += x2_3`; zero_ig.py:343 `fea + conv(fea)`), then optional clamp (llunetpp.py:241,
 * zero_ig.py:346).  Weights HWIO over the concatenated Cin. */
typedef struct ie_convn_desc {
    int N, H, W, nsrc, Cout, kh, kw, stride, pad;
    int pad_mode;          /* ie_pad_mode */
    int act;               /* ie_act */
    float slope;
    int clamp;             /* 0/1 */
    float clamp_lo, clamp_hi;
    int Ho, Wo;            /* out */
} ie_convn_desc;
int ie_convn_nhwc_f32(ie_convn_desc* d, const float* const* srcs_dev, const int* channels, const float* w_hwio_dev,
                      const float* bias_dev, const float* residual_dev, float* dst_dev, void* stream);
/* nn.InstanceNorm2d(affine) on channels [c_begin, c_begin + c_count) (biased variance over H*W per (n,c), eps;
 * llunetpp.py:107; half-instance-norm hinet.py:75-77), other channels copied; fused activation.
 * stats_ws_dev: >= 3*N*c_count doubles of scratch.  gamma/beta may both be NULL (affine=False). */
int ie_instnorm_nhwc_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev, double* stats_ws_dev,
                         int N, int H, int W, int C, int c_begin, int c_count, float eps, int act, float slope, void* stream);
/* nn.MaxPool2d(2,2), floor mode (llunetpp.py:181, unet.py:65). */
int ie_maxpool2x2_nhwc_f32(const float* x_dev, float* y_dev, int N, int H, int W, int C, void* stream);
/* mode 0: F.interpolate default nearest (retinexnet/model.py:78-86); 1: bilinear align_corners=False;
 * 2: bilinear align_corners=True (nn.Upsample, llunetpp.py:182, unet.py:82). */
int ie_resize_nhwc_f32(const float* x_dev, float* y_dev, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, void* stream);
/* nn.PixelShuffle(r) (nn/modules/shuffle.py:17-18, encoder/edsr.py:103): x (N,H,W,Cout*r*r) -> y (N,H*r,W*r,Cout). */
int ie_pixel_shuffle_nhwc_f32(const float* x_dev, float* y_dev, int N, int H, int W, int Cout, int r, void* stream);
/* nn.ConvTranspose2d(k=2, s=2) (hinet.py:101-107, unet.py:85); weights repacked [i][j][Cin][Cout]. */
int ie_conv_transpose2x2_nhwc_f32(const float* x_dev, const float* w_ijco_dev, const float* bias_dev, float* y_dev, int N, int H,
                                  int W, int Cin, int Cout, void* stream);
/* F.pad zero padding (unet.py:90-93). */
int ie_pad_nhwc_f32(const float* x_dev, float* y_dev, int N, int H, int W, int C, int top, int bottom, int left, int right, void* stream);
/* op 0: a + b, 1: a * b (b with C or 1 channels), 2: clamp(a, lo, hi), 3: y(C ch) = a[:, :C] * b where a has
 * C+1 channels and b one (S = R * I_delta with R = first 3 of DecomNet's 4 outputs, retinexnet/model.py:49-50,238),
 * 4: a - b, 5: clamp(a / b, lo, hi) (zero_ig.py:441-442, rrdnet.py:277), 6: a * lo + hi (zero_ig.py:438 `image + eps`),
 * 7: pow(a, lo) (rrdnet.py:276), 8: a / b[0] with b ONE device float (zero_dce_v.py:205 `x / torch.max(x)`),
 * 9: (a / b[0] > lo) ? 1 : 0 (nn/modules/prior/image.py:206-208). */
int ie_eltwise_nhwc_f32(int op, const float* a_dev, const float* b_dev, float* y_dev, size_t npix, int C, int Cb, float lo, float hi, void* stream);
/* [max_c(x), x] channel concat (retinexnet/model.py:45-46). */
int ie_max_concat_nhwc_f32(const float* x_dev, float* y_dev, size_t npix, int C, void* stream);

/* ---- 16-bit matrix-core convolution for the generators (fast path of nn.Conv2d on the U-Net / RetinexNet / LLUnet++ /
 * HINet graphs; same semantics as ie_convn_nhwc_f32) ---------------------------------------------------------------
 * Activations NHWC fp16 or bf16 (channel counts multiples of 8), fp32 accumulate on v_mfma_f32_16x16x32; stride 1, or
 * stride 2 computed as the stride-1 result sub-sampled in the epilogue; zero / replicate padding; virtual concat;
 * fused bias + activation, then residual add (residual in in_dtype, stored shape), then clamp; output in_dtype or fp32.
 * Weights are repacked once per layer AND per source split (K is chunked per source) by ie_mconv_pack_weights. */
typedef struct ie_mconv_desc {
    int N, H, W, nsrc, Cout, kh, kw, stride, pad;
    int pad_mode;          /* ie_pad_mode */
    int act;               /* ie_act */
    float slope;
    int clamp;             /* 0/1 */
    float clamp_lo, clamp_hi;
    int in_dtype;          /* IE_F16 | IE_BF16 */
    int out_dtype;         /* in_dtype | IE_F32 */
    int Ho, Wo;            /* out */
    int src_h[IE_CONV_MAX_SRC], src_w[IE_CONV_MAX_SRC];   /* stored size of source s, or 0 = (H, W).  A smaller source is read through a
                              nearest-neighbour up-sampling to (H, W) -- F.interpolate(x, size=(H, W)) in front of the conv (retinexnet/model.py:79-86) --
                              folded into the kernel's addressing: src = min(floor(dst * in / out), in - 1) */
    int pixel_shuffle;     /* 0, or 2: fused nn.PixelShuffle(2) of the Upsampler (edsr.py:88-123) -- the weights must have been packed with
                              their output rows in [i][j][c] order (row i*2C + j*C + c <- checkpoint row 4c + 2i + j); dst is (N, 2Ho', 2Wo', Cout/4) */
} ie_mconv_desc;
size_t ie_mconv_packed_bytes(const int* channels, int nsrc, int Cout, int kh, int kw);
int ie_mconv_pack_weights(const float* w_oihw_host, const int* channels, int nsrc, int Cout, int kh, int kw, int dtype,
                          void* dst_dev, void* stream);
/* Device-side packing from the trainer's HWIO fp32 master weights (they change every step).  mode 0: the forward conv over the
 * sources `channels`; mode 1: the DATA-GRADIENT conv of source `src_index` -- a conv of dY (one source of Cout channels) with the
 * flipped, transposed weights, output channels = channels[src_index]; its packed size is
 * ie_mconv_packed_bytes({Cout}, 1, channels[src_index], kh, kw).  8-channel (thin) sources are not supported here. */
int ie_mconv_pack_weights_dev(const float* w_hwio_dev, const int* channels, int nsrc, int Cout, int kh, int kw, int dtype, int mode,
                              int src_index, void* dst_dev, void* stream);
int ie_mconv_nhwc(ie_mconv_desc* d, const void* const* srcs_dev, const int* channels, const void* packed_w_dev,
                  const float* bias_dev, const void* residual_dev, void* dst_dev, void* stream);
/* Fused DepthwiseSeparableConv2d (nn/modules/conv/dsconv.py:106-149; zero_dcepp.py:140-146): depthwise 3x3 (zero padding 1,
 * bias) of the virtual concat, then pointwise 1x1 (bias, activation) on the matrix cores; the depthwise result stays in LDS.
 * d: kh = kw = 3, stride 1, pad 1; dw_w_dev: fp32 [Cin_total][9] (the checkpoint's (Cin,1,3,3)); packed_pw_dev: the 1x1 weights
 * from ie_mconv_pack_weights(kh = kw = 1) for the same channel split.  A single 8-channel source is refused (use
 * ie_dwconv3x3_nhwc + ie_mconv_nhwc). */
int ie_dsconv_nhwc(ie_mconv_desc* d, const void* const* srcs_dev, const int* channels, const float* dw_w_dev, const float* dw_b_dev,
                   const void* packed_pw_dev, const float* pw_bias_dev, void* dst_dev, void* stream);

/* ie_mconv_nhwc (stride 1, same-type output, no residual / clamp / pixel shuffle) that ALSO writes the nn.MaxPool2d(2) of its output,
 * pool_dst_dev = NHWC [N, Ho/2, Wo/2, Cout]: the U-Net encoder's `Down` = MaxPool2d(2) -> DoubleConv (reference unet.py:70-72) reads the
 * pooled copy, the skip connection the full one.  One launch for the narrow ReLU layers the strip kernel covers; the conv followed by
 * ie_maxpool2x2_nhwc otherwise (same results). */
int ie_mconv_pool_nhwc(ie_mconv_desc* d, const void* const* srcs_dev, const int* channels, const void* packed_w_dev, const float* bias_dev,
                       void* dst_dev, void* pool_dst_dev, void* stream);

/* The first DoubleConv of a U-Net and the MaxPool2d(2) behind it in one launch (reference unet.py:43-48 DoubleConvBlock, :70-72 Down, :152 `inc`):
 * conv1 = 3x3 / zero padding 1 / bias / ReLU on the 8-channel 16-bit image (img8_dev: NHWC, 16 bytes per pixel; packed_w1_dev from
 * ie_mconv_pack_weights with channels = {8}, Cout = C), conv2 = 3x3 C -> C / bias / ReLU (packed_w2_dev: channels = {C}).  C = 32 or 64; the
 * C-channel intermediate is computed row by row inside the kernel and never reaches HBM.  dst_dev: NHWC [N,H,W,C]; pool_dst_dev: [N,H/2,W/2,C].
 * ie_mconv_first_applicable says (1 / 0) whether a shape is covered; the caller otherwise runs ie_mconv_nhwc + ie_mconv_pool_nhwc. */
int ie_mconv_first_applicable(int C, int H, int W);
int ie_mconv_first_pool_nhwc(int dtype, int N, int H, int W, int C, const void* img8_dev, const void* packed_w1_dev, const float* bias1_dev,
                             const void* packed_w2_dev, const float* bias2_dev, void* dst_dev, void* pool_dst_dev, void* stream);

/* 3x3 conv (+ bias + ReLU) -> 1x1 conv (+ bias, optional clamp) -> NCHW fp32 in one launch: the U-Net head `outc` behind the last DoubleConv
 * (reference unet.py:186-195, 163).  d describes the 3x3 layer (packed_w_dev / bias_dev as for ie_mconv_nhwc; its output is never written);
 * tail_w_dev: the 1x1 weights as 16-bit [16][d->Cout] (rows >= tail_cout zero), tail_bias_dev: fp32 [16]; dst: [N, tail_cout, H, W] fp32.
 * Covered: the strip-kernel layers (k 3, stride 1, pad 1, Cout 32 / 64, ReLU) with W % 4 == 0 and tail_cout <= 16 -- ie_mconv_tail_applicable
 * says (1 / 0); anything else is IE_ERR_UNSUPPORTED (run ie_mconv_nhwc twice). */
int ie_mconv_tail_applicable(const ie_mconv_desc* d, const int* channels, int tail_cout);
int ie_mconv_tail_nhwc(ie_mconv_desc* d, const void* const* srcs_dev, const int* channels, const void* packed_w_dev, const float* bias_dev,
                       const void* tail_w_dev, const float* tail_bias_dev, int tail_cout, int clamp, float clamp_lo, float clamp_hi,
                       float* dst_nchw_dev, void* stream);

/* Two stacked 3x3 convolutions C -> C -> C (stride 1, padding 1, + bias + ReLU each) in one launch, the intermediate tensor never in HBM:
 * replaces two consecutive torch.nn.Conv2d + ReLU of a plain conv stack (reference: DecomNet's `net1_convs`, retinexnet/model.py:28-33; replicate padding
 * pads the INTERMEDIATE of the second layer, as the reference does).  packed_w_a / packed_w_b: ie_mconv_pack_weights images of the two layers (one source
 * of C channels each); src / dst: NHWC, C channels, dtype IE_F16 / IE_BF16.  Bit-identical to two ie_mconv_nhwc calls.  Covered: C = 64, W >= 64 --
 * ie_mconv_pair_applicable says (1 / 0); anything else is IE_ERR_UNSUPPORTED. */
int ie_mconv_pair_applicable(int cin, int cmid, int cout, int kh, int kw, int stride, int pad, int pad_mode, int act_a, int act_b, int W);
int ie_mconv_pair_nhwc(const void* src_dev, const void* packed_w_a_dev, const float* bias_a_dev, const void* packed_w_b_dev, const float* bias_b_dev,
                       void* dst_dev, int dtype, int N, int H, int W, int pad_mode, void* stream);
/* The same with a THIN second layer: Conv3x3(C -> C) + ReLU, then Conv3x3(C -> cout_b) + act_b, cout_b in {4, 8, 12, 16}, act_b in none / relu / sigmoid /
 * tanh -- the end of a plain conv stack (reference: DecomNet's last `net1_convs` layer + `net1_recon` + sigmoid, retinexnet/model.py:33-36,52-56).
 * dst_f32: fp32 NHWC with cout_b channels.  dst16_dev: null, or a second copy of the output as dtype NHWC with dst16_channels = 4 channels or zero-padded to 8
 * (cout_b == 4 only) -- what the next 16-bit network reads (RelightNet's input: 4 for ie_conv3x3_c4_nhwc, 8 for ie_mconv_nhwc), written by the same launch.  fp32 accumulation; agrees with ie_mconv_nhwc x 2 to fp32 rounding. */
int ie_mconv_pair_thin_applicable(int cin, int cmid, int cout, int kh, int kw, int stride, int pad, int pad_mode, int act_a, int act_b, int W);
int ie_mconv_pair_thin_nhwc(const void* src_dev, const void* packed_w_a_dev, const float* bias_a_dev, const void* packed_w_b_dev, const float* bias_b_dev,
                            float* dst_f32_dev, void* dst16_dev, int dst16_channels, int dtype, int N, int H, int W, int cout_b, int act_b, int pad_mode,
                            void* stream);
/* The pair whose first layer has 32 input channels, with the MaxPool2d(2) of the next encoder level in the same launch (reference: U-Net's `down1` =
 * DoubleConv(32, 64) followed by `down2`'s MaxPool2d(2), unet.py:43-48,70-72 for features = 32).  src: NHWC 32 channels; dst: NHWC 64 channels (the skip
 * tensor); pool_dst: NHWC 64 channels, N x floor(H / 2) x floor(W / 2).  packed_w_a: ie_mconv_pack_weights image of the 32 -> 64 layer.  Bit-identical to
 * ie_mconv_nhwc + ie_mconv_nhwc(pool).  Covered: 32 -> 64 -> 64, W >= 64, H >= 2 (ie_mconv_pair_pool_applicable). */
int ie_mconv_pair_pool_applicable(int cin, int cmid, int cout, int kh, int kw, int stride, int pad, int pad_mode, int act_a, int act_b, int H, int W);
int ie_mconv_pair_pool_nhwc(const void* src_dev, const void* packed_w_a_dev, const float* bias_a_dev, const void* packed_w_b_dev, const float* bias_b_dev,
                            void* dst_dev, void* pool_dst_dev, int dtype, int N, int H, int W, int pad_mode, void* stream);

/* Conv2d(4, 64, kernel_size 9, padding 4, padding_mode="replicate") on a 4-channel 16-bit NHWC image (reference: DecomNet's `net1_conv0`,
 * retinexnet/model.py:26, whose input is cat(max_c(x), x)): a strip kernel with K = 8 kx taps x 4 channels per MFMA and nine rotating accumulator sets.
 * pack_weights: (64, 4, 9, 9) fp32 OIHW on the host -> the kernel's fragment image (packed_bytes() bytes on the device).  src: NHWC 4 channels dtype;
 * dst: NHWC 64 channels dtype; no activation; fp32 accumulation. */
size_t ie_conv9x9_c4_packed_bytes(void);
int ie_conv9x9_c4_pack_weights(const float* w_oihw_host, int dtype, void* dst_dev, void* stream);
int ie_conv9x9_c4_nhwc(const void* src4_dev, const void* packed_w_dev, const float* bias_dev, void* dst_dev, int dtype, int N, int H, int W, void* stream);
/* The same kernel at 3x3: Conv2d(4, 64, 3, padding=1, padding_mode="replicate") on a 4-channel image (reference: RelightNet's `net2_conv0_1` on cat(R, L),
 * retinexnet/model.py:58); weights (64, 4, 3, 3). */
size_t ie_conv3x3_c4_packed_bytes(void);
int ie_conv3x3_c4_pack_weights(const float* w_oihw_host, int dtype, void* dst_dev, void* stream);
int ie_conv3x3_c4_nhwc(const void* src4_dev, const void* packed_w_dev, const float* bias_dev, void* dst_dev, int dtype, int N, int H, int W, void* stream);

/* The FIRST level of a 32-channel U-Net in one launch, from the caller's fp32 NCHW image: replaces, in reference terms (unet.py:43-48 `inc` = DoubleConv(3, 32),
 * :70-72 the MaxPool2d(2) of `down1`), x1 = relu(bn(conv3x3(relu(bn(conv3x3(image)))))) and p1 = max_pool2d(x1, 2), BatchNorm folded into the weights.
 * packed_w1: ie_unet_first_pack_weights of the folded (32, 3, 3, 3) weights (ie_unet_first_packed_bytes() bytes); packed_w2: ie_mconv_pack_weights(channels {32},
 * Cout 32).  dst: NHWC 32 channels dtype (N, H, W); pool_dst: (N, H / 2, W / 2).  Agrees with the layout pass + ie_mconv_first_pool_nhwc to fp32 rounding. */
size_t ie_unet_first_packed_bytes(void);
int ie_unet_first_pack_weights(const float* w_oihw_host, int dtype, void* dst_dev, void* stream);
int ie_unet_first_applicable(int C, int H, int W);
int ie_unet_first_pool_nchw(const float* img_nchw_dev, const void* packed_w1_dev, const float* bias1_dev, const void* packed_w2_dev, const float* bias2_dev, void* dst_dev,
                            void* pool_dst_dev, int dtype, int N, int H, int W, void* stream);

/* The last decoder level of a C-channel U-Net + its head in one launch: replaces, in reference terms (unet.py:75-112 `Up.forward`, :186-195 `outc`),
 *   x = cat([skip, up], 1); x = relu(bn(conv3x3(x)))  [2C -> C];  x = relu(bn(conv3x3(x)))  [C -> C];  y = conv1x1(x)  [C -> tail_cout]  (+ clamp)
 * with BatchNorm folded into the weights.  skip: NHWC C channels dtype.  up: the second source of the concatenation, NHWC C channels -- at the skip's
 * resolution (up_h == H, up_w == W), or at HALF resolution (H == 2 up_h, W == 2 up_w): then the kernel also does the reference's bilinear x2 up-sampling
 * (nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True), unet.py:82-83) while it fills its LDS rows, and the up-sampled tensor never exists.
 * packed_w_a: ie_mconv_pack_weights(channels {C, C}, Cout C); packed_w_b: channels {C}, Cout C; tail_w: 16-bit [16][C] (rows >= tail_cout zero), tail_bias
 * fp32 [16].  dst: NCHW fp32 (N, tail_cout, H, W).  Covered: C = 32, W >= 64 -- ie_unet_up_head_applicable says (1 / 0).  Agrees with the separate
 * launches to fp32 rounding (same 16-bit roundings of both intermediates; the in-kernel up-sampling rounds once, like ie_resize_nhwc). */
int ie_unet_up_head_applicable(int C, int tail_cout, int H, int W, int up_h, int up_w);
int ie_unet_up_head_nhwc(const void* skip_dev, const void* up_dev, int up_h, int up_w, const void* packed_w_a_dev, const float* bias_a_dev,
                         const void* packed_w_b_dev, const float* bias_b_dev, const void* tail_w_dev, const float* tail_bias_dev, int tail_cout, int clamp,
                         float clamp_lo, float clamp_hi, float* dst_nchw_dev, int dtype, int N, int H, int W, void* stream);

/* ---- 16-bit (IE_F16 / IE_BF16) NHWC streaming ops between the matrix-core convolutions; IE_F32 forwards to the *_f32
 * entry points above.  On the 16-bit path channel counts are multiples of 8 (a thread moves 8 channels = 16 bytes). */
int ie_maxpool2x2_nhwc(const void* x_dev, void* y_dev, int dtype, int N, int H, int W, int C, void* stream);
int ie_resize_nhwc(const void* x_dev, void* y_dev, int dtype, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, void* stream);
int ie_pad_nhwc(const void* x_dev, void* y_dev, int dtype, int N, int H, int W, int C, int top, int bottom, int left, int right, void* stream);
/* F.pad(up(x), [left, right, top, bottom]) in one pass (reference: `Up.forward`, unet.py:95-103 -- the up-sampled tensor padded to the skip tensor's
 * size): y is (N, Ho + top + bottom, Wo + left + right, C) with the resized image at (top, left) and zeros around it.  16-bit dtypes only; the bits of
 * ie_resize_nhwc followed by ie_pad_nhwc. */
int ie_resize_pad_nhwc(const void* x_dev, void* y_dev, int dtype, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, int top, int bottom, int left, int right,
                       void* stream);
int ie_pixel_shuffle_nhwc(const void* x_dev, void* y_dev, int dtype, int N, int H, int W, int Cout, int r, void* stream);
/* op 0: a + b, 1: a * b; b has C channels or one (broadcast). */
int ie_eltwise_nhwc(int op, const void* a_dev, const void* b_dev, void* y_dev, int dtype, size_t npix, int C, int Cb, void* stream);
/* Depthwise 3x3 (groups = C, zero padding 1, stride 1) + bias + activation on a 16-bit tensor: the first half of
 * DepthwiseSeparableConv2d (nn/modules/conv/dsconv.py:122-134); the pointwise half is ie_mconv_nhwc with a 1x1 kernel.
 * w_tap_c_dev: fp32 [9][C] (tap-major); bias_dev may be NULL. */
int ie_dwconv3x3_nhwc(const void* x_dev, const float* w_tap_c_dev, const float* bias_dev, void* y_dev, int dtype, int N, int H, int W, int C,
                      int act, float slope, void* stream);
/* dst[p][c] = c < Cs ? (dst type) src[p][c] : 0 for c < Cd: dtype conversion and channel padding / cropping in one pass
 * (e.g. DecomNet's 4-channel fp32 input -> 8-channel bf16; a 16-bit map -> the fp32 tensor handed back to the caller). */
int ie_cast_nhwc(const void* src_dev, int src_dtype, int Cs, void* dst_dev, int dst_dtype, int Cd, size_t npix, void* stream);
/* torch.cat(srcs, dim=1) of 1 .. 4 fp32 NHWC tensors + the same conversion / zero padding to Cd channels, one pass: the input of a network whose first layer
 * reads a concatenation (reference: GCENet's EnhanceNet, gcenet.py:283-290: cat([image, depth, edge])) as ONE 8-channel 16-bit tensor. */
int ie_concat_cast_nhwc(const float* const* srcs_dev, const int* channels, int nsrc, void* dst_dev, int dst_dtype, int Cd, size_t npix, void* stream);
/* ie_instnorm_nhwc_f32 for 16-bit tensors (statistics in fp64 from the 16-bit values, one rounding on store);
 * stats_ws_dev: >= 3*N*c_count doubles; c_begin and c_count multiples of 8. */
int ie_instnorm_nhwc(const void* x_dev, const float* gamma_dev, const float* beta_dev, void* y_dev, int dtype, double* stats_ws_dev,
                     int N, int H, int W, int C, int c_begin, int c_count, float eps, int act, float slope, void* stream);

/* ---- training path, fp32 NHWC: what torch.autograd does behind Model.training_step ---------
 * (src/mon/nn/model.py:701-746) for the layers above, the L1 + SSIM losses and Adam with gradient-norm clipping. */
/* Forward of nn.InstanceNorm2d (per_sample=1) or of nn.BatchNorm2d in training mode (per_sample=0: batch
 * statistics over N*H*W, unet.py:43-48 under model.train()) on channels [c_begin, c_begin+c_count), fused
 * affine + activation.  ws_dev: >= 2*G*c_count doubles; save_mean/save_invstd: G*c_count floats kept for the
 * backward pass (G = N if per_sample else 1); running_mean/var (batch mode, may be NULL) are updated with
 * `momentum` and the unbiased variance like torch. */
int ie_norm_forward_nhwc_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev, double* ws_dev,
                             float* save_mean_dev, float* save_invstd_dev, float* running_mean_dev, float* running_var_dev,
                             float momentum, int per_sample, int N, int H, int W, int C, int c_begin, int c_count, float eps,
                             int act, float slope, void* stream);
/* Backward of the above: g = dy * act'(y); dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) (channels outside the
 * range: dx = g); dgamma = sum g*xhat, dbeta = sum g (assigned, or added when accumulate_param_grads).
 * y_dev (the layer's output) is needed when act != IE_ACT_NONE.  ws_dev: >= 2*G*c_count doubles. */
int ie_norm_backward_nhwc_f32(const float* x_dev, const float* dy_dev, const float* y_dev, const float* gamma_dev,
                              const float* save_mean_dev, const float* save_invstd_dev, double* ws_dev, float* dx_dev,
                              float* dgamma_dev, float* dbeta_dev, int accumulate_param_grads, int per_sample, int N, int H, int W,
                              int C, int c_begin, int c_count, int act, float slope, void* stream);
/* Conv2d weight gradient dW (HWIO over the concatenated Cin, like the forward weights) and optional bias gradient
 * (ws_dev: >= Cout doubles when dbias_dev != NULL).  d->act etc. are ignored: dy is the gradient at the conv output. */
int ie_convn_wgrad_nhwc_f32(ie_convn_desc* d, const float* const* srcs_dev, const int* channels, const float* dy_dev,
                            float* dw_hwio_dev, float* dbias_dev, int accumulate, double* ws_dev, void* stream);
/* Conv2d data gradient, one tensor per source of the virtual concat (NULL entry = not needed); zero padding, any stride. */
int ie_convn_dgrad_nhwc_f32(ie_convn_desc* d, const float* dy_dev, const float* w_hwio_dev, float* const* dsrcs_dev,
                            const int* channels, const int* accumulate, void* stream);
/* MaxPool2d(2,2) backward (gradient to the first maximum in scan order, like torch's saved indices). */
int ie_maxpool2x2_backward_nhwc_f32(const float* x_dev, const float* dy_dev, float* dx_dev, int N, int H, int W, int C,
                                    int accumulate, void* stream);
/* Adjoint of ie_resize_nhwc_f32 (same modes): dy (N,Ho,Wo,C) -> dx (N,Hi,Wi,C), deterministic gather form. */
int ie_resize_backward_nhwc_f32(const float* dy_dev, float* dx_dev, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode,
                                int accumulate, void* stream);
/* dst (N,H,W,C) (+)= src[:, top:top+H, left:left+W, :] of an (N,Hs,Ws,C) tensor: adjoint of ie_pad_nhwc_f32. */
int ie_crop_nhwc_f32(const float* src_dev, float* dst_dev, int N, int Hs, int Ws, int C, int top, int left, int H, int W,
                     int accumulate, void* stream);
/* Adjoint of ie_pixel_shuffle_nhwc_f32: dy (N,H*r,W*r,Cout) -> dx (N,H,W,Cout*r*r). */
int ie_pixel_unshuffle_nhwc_f32(const float* dy_dev, float* dx_dev, int N, int H, int W, int Cout, int r, int accumulate, void* stream);
int ie_pixel_unshuffle_nhwc(const void* dy_dev, void* dx_dev, int dtype, int N, int H, int W, int Cout, int r, int accumulate, void* stream);
/* dx (+)= dy * act'(ref): act = ie_act with ref = the activation's OUTPUT, or act = 5: clamp with ref = its INPUT
 * (gradient passes where lo <= ref <= hi, like torch.clamp). */
int ie_act_backward_f32(const float* dy_dev, const float* ref_dev, float* dx_dev, size_t n, int act, float slope, float lo, float hi,
                        int accumulate, void* stream);
/* dtype-generic twin (IE_F32 / IE_F16 / IE_BF16; all three tensors share the dtype; dx may alias dy). */
int ie_act_backward(const void* dy_dev, const void* ref_dev, void* dx_dev, int dtype, size_t n, int act, float slope, float lo, float hi,
                    int accumulate, void* stream);
/* L1Loss (nn/loss/base.py:164-178): *sum_dev = sum |pred - target| (caller divides by n for reduction="mean");
 * grad_dev (may be NULL) (+)= grad_scale * sign(pred - target). */
int ie_l1_loss_f32(const float* pred_dev, const float* target_dev, size_t n, double* sum_dev, float* grad_dev, float grad_scale,
                   int accumulate, void* stream);
/* custom_ssim (nn/metric/image.py:143-237): 11-tap Gaussian window (sigma), separable valid filtering, c1=(k1*R)^2,
 * c2=(k2*R)^2.  Element (n,c,h,w) of both images is at n*sn + c*sc + h*sh + w*sw floats (NCHW or NHWC).
 * sum_dev[N*C] = sum of the SSIM map per (n,c) (mean = sum / ((H-10)*(W-10))).  d_ws_dev (may be NULL):
 * 3*N*C*(H-10)*(W-10) floats of per-pixel partial derivatives kept for ie_ssim_backward_f32. */
int ie_ssim_forward_f32(const float* x_dev, const float* y_dev, int N, int C, int H, int W, int64_t sn, int64_t sc, int64_t sh,
                        int64_t sw, float data_range, float k1, float k2, float sigma, double* sum_dev, float* d_ws_dev, void* stream);
/* grad_dev (same strides as x) (+)= scale * d(sum of all SSIM-map pixels)/dx; SSIMLoss = 1 - mean(...) (nn/loss/image.py:862-865)
 * means scale = -loss_weight / (N*C*(H-10)*(W-10)).  non_negative: planes whose SSIM sum is negative get no gradient (relu). */
int ie_ssim_backward_f32(const float* x_dev, const float* y_dev, const float* d_ws_dev, const double* sum_dev, float* grad_dev, int N,
                         int C, int H, int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float sigma, float scale,
                         int non_negative, int accumulate, void* stream);
/* Deterministic mode (process-wide switch, default off).  On: the conv weight gradients (ie_convn_wgrad_nhwc[_f32]) replace their
 * split-K fp32 atomics by a two-stage reduce -- every split parks its partial tile in a library-owned scratch slab and a second
 * kernel adds the slabs in a fixed order with fp64 accumulation -- so the same inputs give the same bits of dW on every run.  The
 * scratch grows on demand (hipMalloc + device synchronisation on growth: not inside a graph capture).  The remaining reductions of
 * the training path (norm statistics, bias gradients, loss sums) accumulate in fp64 with atomics and round to fp32 ONCE: their
 * result changes only when an fp64 sum falls within ~1e-16 relative of an fp32 rounding boundary. */
int ie_set_deterministic(int on);
int ie_get_deterministic(void);
/* *acc_dev (+)= sum (scale*a_i)^2 -- global gradient norm for torch.nn.utils.clip_grad_norm_ (Lightning gradient_clip_val).
 * Deterministic (no atomics): identical inputs give the identical bits on every data-parallel replica.
 * ws_dev: IE_SUMSQ_WS_DOUBLES doubles of scratch. */
#define IE_SUMSQ_WS_DOUBLES 2048
int ie_sumsq_f32(const float* a_dev, size_t n, float scale, double* acc_dev, double* ws_dev, int zero_first, void* stream);
/* torch.optim.Adam step t=`step` (>=1) on a flat buffer: g = grad*grad_scale*min(1, max_norm/(sqrt(*sumsq_dev)+1e-6))
 * (+ weight_decay*p); sumsq_dev NULL or max_norm <= 0 disables clipping.  A non-finite *sumsq_dev (overflowed fp16 gradients under
 * loss scaling) leaves parameters and moments untouched -- torch.cuda.amp.GradScaler's skipped step. */
int ie_adam_step_f32(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, size_t n, int step, float lr,
                     float beta1, float beta2, float eps, float weight_decay, float grad_scale, const double* sumsq_dev,
                     float max_norm, void* stream);
/* The same step with the step counter ON THE DEVICE: state_dev = {int step, int applied, float bc1, float bc2_sqrt} (16 bytes,
 * zero-initialised by the caller).  A step with a non-finite *sumsq_dev is skipped AND does not advance `step`, so the bias
 * correction and a checkpoint's global_step count the updates actually applied (GradScaler skips optimizer.step() entirely);
 * `applied` reports the outcome of the last call.  Pass sumsq_dev with max_norm <= 0 for the overflow check without clipping. */
int ie_adam_step_counted_f32(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, size_t n, int* state_dev,
                             float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, const double* sumsq_dev,
                             float max_norm, void* stream);

/* ---- dtype-generic twins of the training entry points (dtype = IE_F32 / IE_F16 / IE_BF16 of the activation tensors; parameter
 * gradients, statistics and saved mean / invstd stay fp32 / fp64): the 16-bit training path keeps activations and their gradients in
 * 16 bits between the matrix-core convolutions (forward ie_mconv_nhwc, data gradient ie_mconv_nhwc with ie_mconv_pack_weights_dev
 * mode 1) and fp32 master weights. */
int ie_norm_forward_nhwc(const void* x_dev, const float* gamma_dev, const float* beta_dev, void* y_dev, int dtype, double* ws_dev,
                         float* save_mean_dev, float* save_invstd_dev, float* running_mean_dev, float* running_var_dev, float momentum,
                         int per_sample, int N, int H, int W, int C, int c_begin, int c_count, float eps, int act, float slope, void* stream);
int ie_norm_backward_nhwc(const void* x_dev, const void* dy_dev, const void* y_dev, int dtype, const float* gamma_dev, const float* save_mean_dev,
                          const float* save_invstd_dev, double* ws_dev, void* dx_dev, float* dgamma_dev, float* dbeta_dev,
                          int accumulate_param_grads, int per_sample, int N, int H, int W, int C, int c_begin, int c_count, int act, float slope,
                          void* stream);
/* The same backward with the activation's derivative taken from the RECOMPUTED pre-activation v = gamma * xhat + beta (channels outside the range: v = x)
 * instead of the stored layer output: for act = none / ReLU / LeakyReLU the derivative is a function of sign(v) = sign(y), and neither pass reads y -- two
 * tensor reads per pass instead of three (what torch's fused BatchNorm + ReLU backward saves the output for: nn.BatchNorm2d + nn.ReLU, unet.py:42-47). */
int ie_norm_backward_pre_nhwc(const void* x_dev, const void* dy_dev, int dtype, const float* gamma_dev, const float* beta_dev, const float* save_mean_dev,
                              const float* save_invstd_dev, double* ws_dev, void* dx_dev, float* dgamma_dev, float* dbeta_dev,
                              int accumulate_param_grads, int per_sample, int N, int H, int W, int C, int c_begin, int c_count, int act, float slope,
                              void* stream);
int ie_convn_wgrad_nhwc(ie_convn_desc* d, const void* const* srcs_dev, const int* channels, const void* dy_dev, int dtype, float* dw_hwio_dev,
                        float* dbias_dev, int accumulate, double* ws_dev, void* stream);
int ie_maxpool2x2_backward_nhwc(const void* x_dev, const void* dy_dev, void* dx_dev, int dtype, int N, int H, int W, int C, int accumulate, void* stream);
int ie_resize_backward_nhwc(const void* dy_dev, void* dx_dev, int dtype, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, int accumulate,
                            void* stream);
int ie_crop_nhwc(const void* src_dev, void* dst_dev, int dtype, int N, int Hs, int Ws, int C, int top, int left, int H, int W, int accumulate,
                 void* stream);
/* one tile of a spatial torch.cat (the patch hierarchy of mprnet.py:472-505): dst[:, top:top+H, left:left+W, :] = src (N,H,W,C) */
int ie_paste_nhwc(const void* src_dev, void* dst_dev, int dtype, int N, int H, int W, int C, int Hd, int Wd, int top, int left, void* stream);

/* ---- RetinexNet's training forward (src/mon_extra/vision/enhance/llie/retinexnet/model.py:199-260) ------------------
 * Each call ADDS weight * term to *loss_dev (double) and weight * d term to the gradient tensors it is given (NULL = not wanted); the
 * caller zeroes them once per step.  R / I (and dR / dI) are fp32 pixel-major with a stride in floats: 3 and 1 for separate tensors,
 * 4 and 4 for the (R, L) output of DecomNet (I = base + 3).  T: the target image, NHWC 3 channels.
 *   recon    mean |R * I - T|                      (model.py:216-221: recon_loss_*, recon_loss_mutal_*, relight_loss)
 *   l1_detached  mean |A - B|, gradient to A only  (model.py:220: equal_R_loss against R_high.detach())
 *   smooth   mean over the (H+1) x (W+1) grid of |gx I| exp(-10 avg3x3 |gx gray(R)|) + the same along y   (model.py:240-260) */
/* RetinexNet's inference ends as single passes.  decom_input: fp32 NCHW RGB -> DecomNet's input cat(max_c(x), x) (model.py:45-46) as dtype (IE_F16 /
 * IE_BF16) NHWC with cpad = 4 channels (what ie_conv9x9_c4_nhwc reads) or padded to cpad = 8 (what ie_mconv_nhwc reads).  outputs: rl = DecomNet's (R, L) output, fp32 pixel-major 4 floats, i_delta fp32 one float per pixel ->
 * R (N,3,H,W), L (N,1,H,W) (either may be NULL) and S = R * I_delta (N,3,H,W) (model.py:72-74, 229-231), NCHW fp32. */
int ie_retinex_decom_input(const float* x_nchw_dev, void* dst_nhwc_dev, int dtype, int cpad, int N, int H, int W, void* stream);
int ie_retinex_outputs_f32(const float* rl_nhwc4_dev, const float* i_delta_dev, float* R_nchw_dev, float* L_nchw_dev, float* S_nchw_dev, int N, int H, int W,
                           void* stream);
int ie_retinex_recon_f32(const float* R_dev, int r_stride, const float* I_dev, int i_stride, const float* T_dev, float weight, float* dR_dev,
                         float* dI_dev, double* loss_dev, int N, int H, int W, void* stream);
int ie_l1_detached_f32(const float* A_dev, const float* B_dev, int stride, int C, float weight, float* dA_dev, double* loss_dev, size_t npx, void* stream);
size_t ie_retinex_smooth_ws_floats(int N, int H, int W);
int ie_retinex_smooth_f32(const float* I_dev, int i_stride, const float* R_dev, int r_stride, float weight, float* dI_dev, float* dR_dev, double* loss_dev,
                          float* ws_dev, int N, int H, int W, void* stream);
/* replicate padding as its own op and its adjoint (every padded position adds its gradient to the edge pixel it copied): the backward
 * pass of a padding_mode="replicate" convolution (model.py:26-66) is the zero-padding one on the explicitly padded input */
int ie_pad_replicate_nhwc_f32(const float* x_dev, float* y_dev, int N, int H, int W, int C, int pad, void* stream);
int ie_pad_replicate_backward_nhwc_f32(const float* dy_dev, float* dx_dev, int N, int H, int W, int C, int pad, int accumulate, void* stream);
/* The same pair for IE_F16 / IE_BF16 tensors (C a multiple of 8; IE_F32 forwards to the entry points above), and the zero insertion in front of the data gradient
 * of a stride-2 convolution (dst[n, 2h, 2w, :] = src[n, h, w, :], zeros elsewhere; dst (N, Ho, Wo, C), Ho >= 2 H - 1): with them RetinexNet's replicate-padded and
 * strided layers (retinexnet/model.py:26-66) train on the 16-bit matrix-core graph -- dX = conv(zero_insert(dY), W flipped, padding k - 1 - p). */
int ie_pad_replicate_nhwc(const void* x_dev, void* y_dev, int dtype, int N, int H, int W, int C, int pad, void* stream);
int ie_pad_replicate_backward_nhwc(const void* dy_dev, void* dx_dev, int dtype, int N, int H, int W, int C, int pad, int accumulate, void* stream);
int ie_zero_insert2_nhwc(const void* x_dev, void* y_dev, int dtype, int N, int H, int W, int C, int Ho, int Wo, void* stream);

/* ---- zero-reference training of Zero-DCE: the losses of zero_dce.py:34-79 and the LE-curve backward --------- */
/* floats of scratch ie_zref_image_losses_f32 needs for an (N,3,H,W) batch. */
size_t ie_zref_workspace_floats(int N, int H, int W);
/* SpatialConsistencyLoss(num_regions=4, patch 4) (nn/loss/image.py:905-1270), ExposureControlLoss(patch 16, mean exp_mean)
 * (:534-568) and ColorConstancyLoss (:191-217) of `enhanced` against `image`, both NCHW fp32 (N,3,H,W), H,W >= 16.
 * sums_dev (>= 3 + 3N doubles): [0] = sum over pooled cells of the spatial term (mean = /(N*(H/4)*(W/4))), [1] = sum of the
 * exposure term (/(N*(H/16)*(W/16))), [2] = sum over frames of the colour term (/N).  grad_enhanced_dev (may be NULL)
 * (+)= d(w_spa*L_spa + w_exp*L_exp + w_col*L_col)/d enhanced. */
int ie_zref_image_losses_f32(const float* enhanced_dev, const float* image_dev, int N, int H, int W, float exp_mean, float w_spa,
                             float w_exp, float w_col, double* sums_dev, float* ws_dev, float* grad_enhanced_dev, int accumulate,
                             void* stream);
/* TotalVariationLoss (nn/loss/image.py:1355-1392) on NHWC fp32 maps (the 24 curve maps): sums_dev[0] / [1] = sums of squared
 * vertical / horizontal differences; loss = weight * 2 * (s0/(C*(H-1)*W) + s1/(C*H*(W-1))) / N; grad_dev (may be NULL). */
int ie_tv_loss_nhwc_f32(const float* a_dev, int N, int H, int W, int C, float weight, double* sums_dev, float* grad_dev, int accumulate,
                        void* stream);
/* Backward of ie_le_curve with respect to the curve maps: grad_adjust (NHWC, Cp channels) (+)= d enhanced / d adjust applied to
 * grad_enhanced (NCHW).  The image receives no gradient (it is the network input). */
int ie_le_curve_backward(const float* img_dev, const float* adjust_nhwc_dev, int Cp, const float* grad_enhanced_dev,
                         float* grad_adjust_nhwc_dev, int N, int H, int W, int iters, int accumulate, void* stream);

/* ---- colour space, guided filter and the generalised LE curve (ie_filter.hip): the V-channel / guided curve models --------
 * ZeroDCEV (src/mon/vision/enhance/llie/zero_dce/zero_dce_v.py:172-230), GCENet (.../gcenet/gcenet.py:460-502),
 * filters src/mon/vision/filtering/guided_filter.py:99-204 and box_filter.py:24-110.  fp32 NHWC.  The kornia functions the
 * reference calls there (rgb_to_hsv, hsv_to_rgb, median_blur, sobel) are restated from kornia's published algorithms. */
/* core.rgb_to_hsv (core/image/color.py:171-195 -> kornia.color.rgb_to_hsv): h in [0, 2 pi), s, v.  hsv_dev (3 ch) and / or
 * v_dev (1 ch, core.rgb_to_v, :198-218) may be NULL. */
int ie_rgb_to_hsv_nhwc_f32(const float* rgb_dev, float* hsv_dev, float* v_dev, size_t npix, float eps, void* stream);
/* core.hsv_to_rgb (:274-291 -> kornia.color.hsv_to_rgb); v_new_dev (1 ch, may be NULL) replaces the V channel first
 * (ZeroDCEV.replace_v_component, zero_dce_v.py:225-230). */
int ie_hsv_to_rgb_nhwc_f32(const float* hsv_dev, const float* v_new_dev, float* rgb_dev, size_t npix, void* stream);
/* core.RGBToHVI (src/mon/core/image/color.py:313-434, the HVI-CIDNet colour transform LLLINetHVI wraps its network in, lllinet.py:360-385): rgb_to_hvi and
 * hvi_to_rgb on fp32 NHWC 3-channel tensors; density_k = the module's learnable `density_k` (0.2), eps 1e-8; clamp01: the clamp(0, 1) the model applies behind it. */
int ie_rgb_to_hvi_nhwc_f32(const float* x_dev, float* hvi_dev, size_t npix, float density_k, float eps, void* stream);
int ie_hvi_to_rgb_nhwc_f32(const float* hvi_dev, float* rgb_dev, size_t npix, float density_k, float eps, int clamp01, void* stream);
/* y <- y + r_k (y^2 - y), k < iters, on a C-channel NHWC image with Ca = C*iters curve maps (Ca == C: one shared map);
 * bam_dev (1 ch, may be NULL): GCENet's brightness-attention form, gcenet.py:478-484 (bright = y (1 - bam), dark = y bam,
 * y = bright + dark + r (dark^2 - dark)). */
int ie_le_curve_nhwc_f32(const float* x_dev, const float* adjust_dev, const float* bam_dev, float* y_dev, size_t npix, int C, int Ca,
                         int iters, void* stream);
/* Un-normalised box filter: window sums over (2r+1)^2 clipped to the image == box_filter.py:24-110 (cumsum differences). */
int ie_box_filter_nhwc_f32(const float* x_dev, float* y_dev, int N, int H, int W, int C, int radius, void* stream);
/* GuidedFilter(radius, eps)(image = x, guide = y) (guided_filter.py:99-147): out = mean_A * x + mean_b.  Cx = 1 or Cy.
 * ws_dev: 2 * N*H*W*Cy floats. */
int ie_guided_filter_nhwc_f32(const float* x_dev, const float* y_dev, float* out_dev, float* ws_dev, int N, int H, int W, int Cx, int Cy,
                              int radius, float eps, void* stream);
/* FastGuidedFilter(radius, eps)(x_lr, y_lr, x_hr) (guided_filter.py:150-204): coefficients at (h, w), bicubic (align_corners)
 * interpolation to (H, W), out = A * x_hr + b, clipped to [lo, hi] when clip (ZeroDCEV.filter_up).  ws_dev: 2 * N*h*w*Cy floats. */
int ie_fast_guided_filter_nhwc_f32(const float* x_lr_dev, const float* y_lr_dev, const float* x_hr_dev, float* out_dev, float* ws_dev, int N,
                                   int h, int w, int H, int W, int Cx, int Cy, int radius, float eps, int clip, float lo, float hi,
                                   void* stream);
/* F.interpolate(mode="bicubic", align_corners=True) (guided_filter.py:201-202). */
int ie_resize_bicubic_nhwc_f32(const float* x_dev, float* y_dev, int N, int h, int w, int H, int W, int C, void* stream);
/* *out_dev = max over n floats (two deterministic stages; ws_dev: 1024 floats) -- `x / torch.max(x)` (zero_dce_v.py:205). */
int ie_reduce_max_f32(const float* a_dev, size_t n, float* out_dev, float* ws_dev, void* stream);
/* kornia.filters.median_blur (k x k, zero padding; nn/modules/prior/image.py:271) and kornia.filters.sobel magnitude
 * (replicate padding, un-normalised kernel; prior/image.py:205). */
int ie_median_blur_nhwc_f32(const float* x_dev, float* y_dev, int N, int H, int W, int C, int ksize, void* stream);
int ie_sobel_nhwc_f32(const float* x_dev, float* y_dev, int N, int H, int W, int C, float eps, void* stream);

/* ---- per-(sample, channel) statistics and the layers built on them (ie_attn.hip) -------------------------------
 * stats_dev: N*C*2 doubles {sum, sum of squares over H*W} (zeroed by the call). */
int ie_channel_stats_nhwc_f32(const float* x_dev, double* stats_dev, int N, int H, int W, int C, void* stream);
/* LearnableInstanceNorm2d (src/mon/nn/modules/normalization.py:177-224): y = IN(x; gamma, beta) * r + x * (1 - r) == x * scale + shift
 * per (n, c); gamma / beta / r (C floats each) may be NULL (1, 0, 1).  scale_dev / shift_dev: N*C floats. */
int ie_lin_coeffs_f32(const double* stats_dev, const float* gamma_dev, const float* beta_dev, const float* r_dev, float* scale_dev,
                      float* shift_dev, int N, int C, int HW, float eps, void* stream);
/* mean_dev[n][c] = sum / HW: nn.AdaptiveAvgPool2d(1) of the channel-attention layer (mprnet.py:60, 76). */
int ie_stats_mean_f32(const double* stats_dev, float* mean_dev, int N, int C, int HW, void* stream);
/* y = act(x * scale[n][c] + shift[n][c]) (shift_dev may be NULL): the apply pass of the learnable instance norm (+ LeakyReLU,
 * lllinet.py:140-142) and the per-channel rescale of channel attention (mprnet.py:78). */
int ie_channel_affine_nhwc_f32(const float* x_dev, const float* scale_dev, const float* shift_dev, float* y_dev, int N, int H, int W, int C,
                               int act, float slope, void* stream);
/* the same two for fp32 / fp16 / bf16 activations (statistics in fp64, coefficients in fp32): channel attention on the 16-bit path */
int ie_channel_stats_nhwc(const void* x_dev, int dtype, double* stats_dev, int N, int H, int W, int C, void* stream);
int ie_channel_affine_nhwc(const void* x_dev, const float* scale_dev, const float* shift_dev, void* y_dev, int dtype, int N, int H, int W, int C,
                           int act, float slope, void* stream);
/* y = act(x * scale + shift) + res in one pass, one rounding of the fp32 sum (res_dev may be NULL): the channel-attention block's gate and its
 * skip connection (mprnet.py:156-161 `y = self.CA(y); y += x`). */
int ie_channel_affine_add_nhwc(const void* x_dev, const float* scale_dev, const float* shift_dev, const void* res_dev, void* y_dev, int dtype, int N,
                               int H, int W, int C, int act, float slope, void* stream);
int ie_simam_nhwc(const void* x_dev, const double* stats_dev, void* y_dev, int dtype, int N, int H, int W, int C, float lambda, void* stream);
/* SimAM (src/mon/nn/modules/attention.py:1156-1182) from the statistics of its own input. */
int ie_simam_nhwc_f32(const float* x_dev, const double* stats_dev, float* y_dev, int N, int H, int W, int C, float lambda, void* stream);

/* ---- LE curve (zero_dce.py:186-188) ------------------------------------ */
/* y <- y + r_i (y^2 - y), i < iters; img NCHW fp32 (N,3,H,W); adjust NHWC fp32
 * with Cp channels (first 3*iters used, already tanh'ed; 3 <= Cp < 3*iters: one
 * shared 3-channel map for every iteration, zero_dcepp.py:201-203); out NCHW fp32. */
int ie_le_curve(const float* img_dev, const float* adjust_nhwc_dev, int Cp, float* enhanced_dev,
                int N, int H, int W, int iters, void* stream);

/* ---- Zero-DCE plan: the fused hot path --------------------------------- */
/* Replaces ZeroDCE_RE.forward (src/mon/vision/enhance/llie/zero_dce/zero_dce.py:173-191)
 * == enhance_net_nopool.forward (src/mon_extra/vision/enhance/llie/zero_dce/model.py:28-50). */
typedef struct ie_zdce_plan ie_zdce_plan;

typedef struct ie_zdce_config {
    int in_channels;   /* 3 */
    int num_channels;  /* 32 */
    int num_iters;     /* 8 */
    int compute_dtype; /* ie_dtype: IE_F32 (parity mode, VALU fp32), IE_F16 or IE_BF16 (MFMA, fp32 accumulate) */
    int device;        /* HIP device ordinal */
    int variant;       /* kernel variant bitmask, 0 = default; see IE_ZDCE_VAR_* */
} ie_zdce_config;

#define IE_ZDCE_VAR_SYNC_STAGE 1 /* drain the LDS-DMA ring every row (debug / A-B) */
#define IE_ZDCE_VAR_NARROW_STORE 2 /* 8-byte epilogue stores instead of permlane-widened 16-byte */
#define IE_ZDCE_VAR_NO_DMA 4 /* register-staged global->LDS copies instead of LDS-DMA (debug / A-B) */
#define IE_ZDCE_VAR_UNFUSED 8 /* one kernel per layer instead of the fused conv1+2 / conv4+5 / conv6+7 kernels */
#define IE_ZDCE_VAR_NO_K345 32 /* e_conv3 as its own kernel + the two-layer conv4+5 kernel instead of the three-layer conv3+4+5 kernel (A-B) */
#define IE_ZDCE_VAR_K67_LEGACY 64 /* round-2 form of the last kernel: e_conv7 rows in channel order, tanh + curve on fp32 rows parked in LDS (A-B) */
#define IE_ZDCE_VAR_P_SCHEDULE 128
#define IE_ZDCE_VAR_K345_LEGACY 256      /* e_conv3 + e_conv4 + e_conv5 in round 2's (cout fragment, strip half) wave mapping instead of K345X (A-B runs) */ /* round-3 schedule (A-B): the first kernel also computes P = W7a * x1 + b7 (64 B/px written, read back by the last kernel) instead of
                                      the last kernel recomputing x1 from the image and contracting e_conv7 over both of its sources */
#define IE_ZDCE_VAR_PURE_BF16 16 /* compute_dtype IE_BF16 only: every 16-bit tensor and weight in bf16.  Without it the
                                    "bf16" schedule holds the tensors that carry the rounding error (x1, x2, x6, the partial
                                    sum P and the weights contracted with them) in fp16 and x3/x4/x5 (+ theirs) in bf16 --
                                    same bytes, same MFMA rate, 6e-4 instead of 2e-3 of the fp32 result */

int ie_zdce_create(const ie_zdce_config* cfg, ie_zdce_plan** out);
/* layer in 1..7 (e_conv{layer}); w OIHW fp32 host, b fp32 host; shapes per zero_dce.py:133-139. */
int ie_zdce_set_layer(ie_zdce_plan* p, int layer, const float* w_oihw_host, const float* b_host, int Cout, int Cin);
/* Uploads the repacked weights; all 7 layers must be set. */
int ie_zdce_finalize(ie_zdce_plan* p, void* stream);
/* Ensure workspace for (N,H,W); may hipMalloc (synchronises). Called implicitly by forward. */
int ie_zdce_reserve(ie_zdce_plan* p, int N, int H, int W);
/* image_dev: NCHW fp32 (N,3,H,W).  Any of the outputs may be NULL:
 *   enhanced_dev  NCHW fp32 (N,3,H,W)        outputs["enhanced"]
 *   adjust_dev    NCHW fp32 (N,3*iters,H,W)  outputs["adjust"] (split views give adjust_i)
 *   u8_dev        NHWC uint8 (N,H,W,3)       save_image rule applied to enhanced */
int ie_zdce_forward(ie_zdce_plan* p, const float* image_dev, float* enhanced_dev, float* adjust_dev,
                    uint8_t* u8_dev, int N, int H, int W, void* stream);
/* uint8 in, uint8 (and / or fp32) out -- the frames as the reference's loader hands them over (HWC uint8, core/image/io.py:82-89; /255
 * like to_image_tensor, core/image/utils.py:801-847) and as `write_image` wants them back (save_image rule): replaces
 * read -> to_image_tensor -> forward -> save_image of project/runml/predict.py:76-105 for this model.  image_u8_dev: (N,H,W,3). */
int ie_zdce_forward_u8(ie_zdce_plan* plan, const uint8_t* image_u8_dev, float* enhanced_dev, float* adjust_dev, uint8_t* u8_dev, int N, int H,
                       int W, void* stream);
/* Debug/unit parity: copy intermediate feature x{idx} (idx 1..6) of the last forward to
 * NCHW fp32 (N,num_channels,H,W). */
int ie_zdce_get_feature(ie_zdce_plan* p, int idx, float* dst_dev, void* stream);
size_t ie_zdce_workspace_bytes(const ie_zdce_plan* p);
int ie_zdce_destroy(ie_zdce_plan* p);

/* Timing helper for bench.py: runs `iters` forwards back to back on `stream`
 * bracketed by hipEvents recorded on that same stream; *ms_total = elapsed ms. */
int ie_zdce_time_forward(ie_zdce_plan* p, const float* image_dev, float* enhanced_dev, float* adjust_dev,
                         uint8_t* u8_dev, int N, int H, int W, int iters, void* stream, float* ms_total);
/* Per-kernel timing of one forward (hipEvents around each launch, same stream):
 * per layer (cap >= 7): ms[] = summed kernel time, launches[] = launch count (may be NULL),
 * names[] = static strings; *n = 7. */
int ie_zdce_profile_forward(ie_zdce_plan* p, const float* image_dev, float* enhanced_dev, float* adjust_dev,
                            uint8_t* u8_dev, int N, int H, int W, void* stream,
                            float* ms, const char** names, int* launches, int cap, int* n);
/* Schedule knobs (0 = automatic): rows per workgroup, rows per band, frames per group.
 * Also read from IE_ZDCE_RB / IE_ZDCE_BAND / IE_ZDCE_GROUP at create time. */
int ie_zdce_set_tuning(ie_zdce_plan* p, int rows_per_block, int band_rows, int group_frames);

#ifdef __cplusplus
}
#endif
#endif /* IE_HIP_H */
