/* Oracle (TEST INFRASTRUCTURE ONLY): plain-C CPU restatement of the hot path's
 * leaf arithmetic.  Never linked into the product library.
 *
 * Follows torch.nn.Conv2d semantics as used by the reference
 *   src/mon/vision/enhance/llie/zero_dce/zero_dce.py:133-139 (Conv2d 3x3 s1 p1, bias)
 *   src/mon/vision/enhance/llie/zero_dce/zero_dce.py:173-191 (forward + LE curve)
 *   src/mon_extra/vision/enhance/llie/retinexnet/model.py:26-66 (replicate pad, k=9, stride 2)
 * i.e. cross-correlation  y[n,o,h,w] = b[o] + sum_{c,i,j} W[o,c,i,j] x[n,c,h*s+i-p,w*s+j-p].
 *
 * Sums are accumulated in double and rounded once to float, so this is an
 * order-independent anchor (<= 1 ulp from the exact value) against which both
 * the torch restatement (oracle/zero_dce.py) and the HIP path are compared.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { IEO_ACT_NONE = 0, IEO_ACT_RELU = 1, IEO_ACT_LRELU = 2, IEO_ACT_TANH = 3, IEO_ACT_SIGMOID = 4 };
enum { IEO_PAD_ZERO = 0, IEO_PAD_REPLICATE = 1 };

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float act_f(double v, int act, float slope) {
    switch (act) {
    case IEO_ACT_RELU:    return (float)(v > 0 ? v : 0);
    case IEO_ACT_LRELU:   return (float)(v > 0 ? v : v * (double)slope);
    case IEO_ACT_TANH:    return (float)tanh(v);
    case IEO_ACT_SIGMOID: return (float)(1.0 / (1.0 + exp(-v)));
    default:              return (float)v;
    }
}

/* Generic NCHW fp32 conv2d; groups in {1, C}.  Output size follows torch:
 * Ho = (H + 2p - k)/s + 1. */
int ieo_conv2d_nchw_f32(const float* x, int N, int C, int H, int W,
                        const float* w, const float* b, int Co, int kh, int kw,
                        int stride, int pad, int pad_mode, int groups,
                        int act, float slope, float* y) {
    if (groups != 1 && !(groups == C && Co == C)) return -1;
    const int Ho = (H + 2 * pad - kh) / stride + 1;
    const int Wo = (W + 2 * pad - kw) / stride + 1;
    const int cig = C / groups; /* input channels per group */
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Co; ++o) {
            const int c0 = (groups == 1) ? 0 : o;
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    double acc = b ? (double)b[o] : 0.0;
                    for (int ci = 0; ci < cig; ++ci)
                        for (int i = 0; i < kh; ++i)
                            for (int j = 0; j < kw; ++j) {
                                int hi = ho * stride + i - pad, wi = wo * stride + j - pad;
                                if (pad_mode == IEO_PAD_REPLICATE) {
                                    hi = clampi(hi, 0, H - 1);
                                    wi = clampi(wi, 0, W - 1);
                                } else if (hi < 0 || hi >= H || wi < 0 || wi >= W) {
                                    continue;
                                }
                                acc += (double)w[((size_t)(o * cig + ci) * kh + i) * kw + j] *
                                       (double)x[(((size_t)n * C + c0 + ci) * H + hi) * W + wi];
                            }
                    y[(((size_t)n * Co + o) * Ho + ho) * Wo + wo] = act_f(acc, act, slope);
                }
        }
    return 0;
}

/* LE curve, zero_dce.py:186-188: y <- y + r_i*(y*y - y), fp32 in this order. */
void ieo_le_curve_f32(const float* x, const float* r, int N, int H, int W, int iters, float* y) {
    const size_t P = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < 3; ++c)
            for (size_t p = 0; p < P; ++p) {
                float v = x[((size_t)n * 3 + c) * P + p];
                for (int i = 0; i < iters; ++i) {
                    const float ri = r[((size_t)n * 3 * iters + 3 * i + c) * P + p];
                    v = v + ri * (v * v - v);
                }
                y[((size_t)n * 3 + c) * P + p] = v;
            }
}

static void cat2(const float* a, const float* b, int N, int Ca, int Cb, size_t P, float* o) {
    for (int n = 0; n < N; ++n) {
        memcpy(o + (size_t)n * (Ca + Cb) * P, a + (size_t)n * Ca * P, sizeof(float) * Ca * P);
        memcpy(o + ((size_t)n * (Ca + Cb) + Ca) * P, b + (size_t)n * Cb * P, sizeof(float) * Cb * P);
    }
}

/* Whole Zero-DCE forward (zero_dce.py:173-191).  wts[k]/bias[k]: OIHW fp32 of
 * e_conv{k+1}.  adjust: (N,3*iters,H,W); enhanced: (N,3,H,W). */
int ieo_zero_dce_forward_f32(const float* x, int N, int H, int W, int nc, int iters,
                             const float* const* wts, const float* const* bias,
                             float* adjust, float* enhanced) {
    const size_t P = (size_t)H * W;
    const size_t S = (size_t)N * nc * P;
    float* buf = (float*)malloc(sizeof(float) * S * 8);
    if (!buf) return -2;
    float *x1 = buf, *x2 = buf + S, *x3 = buf + 2 * S, *x4 = buf + 3 * S, *x5 = buf + 4 * S,
          *x6 = buf + 5 * S, *cat = buf + 6 * S;
    const int R = IEO_ACT_RELU;
    ieo_conv2d_nchw_f32(x, N, 3, H, W, wts[0], bias[0], nc, 3, 3, 1, 1, 0, 1, R, 0, x1);
    ieo_conv2d_nchw_f32(x1, N, nc, H, W, wts[1], bias[1], nc, 3, 3, 1, 1, 0, 1, R, 0, x2);
    ieo_conv2d_nchw_f32(x2, N, nc, H, W, wts[2], bias[2], nc, 3, 3, 1, 1, 0, 1, R, 0, x3);
    ieo_conv2d_nchw_f32(x3, N, nc, H, W, wts[3], bias[3], nc, 3, 3, 1, 1, 0, 1, R, 0, x4);
    cat2(x3, x4, N, nc, nc, P, cat);
    ieo_conv2d_nchw_f32(cat, N, 2 * nc, H, W, wts[4], bias[4], nc, 3, 3, 1, 1, 0, 1, R, 0, x5);
    cat2(x2, x5, N, nc, nc, P, cat);
    ieo_conv2d_nchw_f32(cat, N, 2 * nc, H, W, wts[5], bias[5], nc, 3, 3, 1, 1, 0, 1, R, 0, x6);
    cat2(x1, x6, N, nc, nc, P, cat);
    ieo_conv2d_nchw_f32(cat, N, 2 * nc, H, W, wts[6], bias[6], 3 * iters, 3, 3, 1, 1, 0, 1,
                        IEO_ACT_TANH, 0, adjust);
    ieo_le_curve_f32(x, adjust, N, H, W, iters, enhanced);
    free(buf);
    return 0;
}

/* torchvision save_image rule (core/image/io.py:139-140): NCHW float -> NHWC u8 */
void ieo_to_u8_round(const float* x, int N, int C, int H, int W, uint8_t* o) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    float v = x[(((size_t)n * C + c) * H + h) * W + w] * 255.0f + 0.5f;
                    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
                    o[(((size_t)n * H + h) * W + w) * C + c] = (uint8_t)v;
                }
}
