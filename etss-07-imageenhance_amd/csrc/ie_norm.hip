// Normalisation layers on NHWC fp32, forward and backward: InstanceNorm2d (statistics per (n, c)) and
// training-mode BatchNorm2d (statistics per c over N*H*W), optional channel sub-range (HINet's half
// instance norm), fused affine + activation.  All passes are HBM-bound streams; the per-channel sums are
// double-precision block partials + atomics (wave64 lanes walk channels, so loads stay coalesced).
// Reference call sites are cited at the entry points in include/ie_hip.h.
#include "ie_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

// thread = (pixel slot, channel); accumulates sum / sum of squares of x, or of g and g*xhat (backward)
// PRE (backward passes): the activation's derivative is taken from the RECOMPUTED pre-activation v = gamma * xhat + beta instead of the stored layer output
// (ReLU / LeakyReLU: the sign of v is the sign of the output) -- one tensor read less in both passes of the backward; pg / pb = gamma / beta of the range or null.
__device__ __forceinline__ float ie_act_grad_from_pre(float v, int act, float slope) { return v > 0.f ? 1.f : (act == IE_ACT_LRELU ? slope : 0.f); }

template <bool BWD, typename T>
__global__ void norm_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ yact,
                                   const float* __restrict__ mean, const float* __restrict__ inv, double* __restrict__ stats,
                                   int HW, int C, int c0, int cn, int cn_total, int c_off, int n_mul, int act, float slope,
                                   const float* __restrict__ pg = nullptr, const float* __restrict__ pb = nullptr, int pre = 0) {
    const int n = blockIdx.y;
    const int tpc = max(1, 256 / cn);
    const int c = threadIdx.x % cn, ps = threadIdx.x / cn;
    const int g = n * n_mul;
    double s = 0.0, ss = 0.0;
    if (ps < tpc) {
        const size_t base = (size_t)n * HW * C + c0 + c;
        float m = 0.f, iv = 1.f, ga = 1.f, be = 0.f;
        if (BWD) { m = mean[(size_t)g * cn_total + c_off + c]; iv = inv[(size_t)g * cn_total + c_off + c]; }
        if (BWD && pre && pg) { ga = pg[c_off + c]; be = pb[c_off + c]; }
        for (int pix = blockIdx.x * tpc + ps; pix < HW; pix += gridDim.x * tpc) {
            const size_t i = base + (size_t)pix * C;
            if (BWD) {
                float gr = (float)dy[i];
                const float xh = ((float)x[i] - m) * iv;
                if (act != IE_ACT_NONE) gr *= pre ? ie_act_grad_from_pre(fmaf(xh, ga, be), act, slope) : ie_act_grad_from_output((float)yact[i], act, slope);
                s += gr; ss += (double)gr * xh;
            } else {
                const float v = (float)x[i];
                s += v; ss += (double)v * v;
            }
        }
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = ss;
    __syncthreads();
    if (ps == 0) {
        for (int k = 1; k < tpc; ++k) { s += sh[0][k * cn + c]; ss += sh[1][k * cn + c]; }
        atomicAdd(&stats[((size_t)g * cn_total + c_off + c) * 2 + 0], s);
        atomicAdd(&stats[((size_t)g * cn_total + c_off + c) * 2 + 1], ss);
    }
}

// mean / 1/sqrt(var+eps) per (group, channel); BatchNorm running statistics (unbiased variance, nn.BatchNorm2d)
__global__ void norm_finalize_kernel(const double* __restrict__ stats, float* __restrict__ mean, float* __restrict__ inv, int total,
                                     double count, float eps, float* run_mean, float* run_var, float momentum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double m = stats[2 * (size_t)i] / count;
    const double var = fmax(stats[2 * (size_t)i + 1] / count - m * m, 0.0);
    mean[i] = (float)m;
    inv[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        run_mean[i] = (1.f - momentum) * run_mean[i] + momentum * (float)m;
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        run_var[i] = (1.f - momentum) * run_var[i] + momentum * (float)unb;
    }
}

template <typename T>
__global__ void norm_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ inv,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, int HW,
                                  int C, int c0, int cn, int n_mul, int act, float slope, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    float v = (float)x[i];
    if (c >= c0 && c < c0 + cn) {
        const size_t g = (i / ((size_t)HW * C)) * n_mul;
        v = (v - mean[g * cn + (c - c0)]) * inv[g * cn + (c - c0)];
        if (gamma) v = v * gamma[c - c0] + beta[c - c0];
    }
    y[i] = (T)ie_act_apply(v, act, slope);
}

// dx = gamma * inv * (g - mean(g) - xhat * mean(g * xhat)); channels outside the range: dx = g
template <typename T>
__global__ void norm_backward_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ yact,
                                           const float* __restrict__ mean, const float* __restrict__ inv,
                                           const float* __restrict__ gamma, const double* __restrict__ stats, T* __restrict__ dx,
                                           int HW, int C, int c0, int cn, int n_mul, double count, int act, float slope, size_t total,
                                           const float* __restrict__ pb = nullptr, int pre = 0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    float gr = (float)dy[i];
    const bool in = c >= c0 && c < c0 + cn;
    if (act != IE_ACT_NONE) {
        if (!pre) gr *= ie_act_grad_from_output((float)yact[i], act, slope);
        else if (!in) gr *= ie_act_grad_from_pre((float)x[i], act, slope);      // outside the range the layer is the bare activation
    }
    if (in) {
        const size_t k = (i / ((size_t)HW * C)) * n_mul * cn + (c - c0);
        const float iv = inv[k];
        const float xh = ((float)x[i] - mean[k]) * iv;
        if (act != IE_ACT_NONE && pre) gr *= ie_act_grad_from_pre(fmaf(xh, gamma ? gamma[c - c0] : 1.f, pb ? pb[c - c0] : 0.f), act, slope);
        const float m1 = (float)(stats[2 * k] / count), m2 = (float)(stats[2 * k + 1] / count);
        gr = (gamma ? gamma[c - c0] : 1.f) * iv * (gr - m1 - xh * m2);
    }
    dx[i] = (T)gr;
}

// dgamma[c] = sum over groups of sum(g * xhat), dbeta[c] = sum over groups of sum(g)
__global__ void norm_param_grad_kernel(const double* __restrict__ stats, float* dgamma, float* dbeta, int groups, int cn, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cn) return;
    double s1 = 0.0, s2 = 0.0;
    for (int g = 0; g < groups; ++g) { s1 += stats[((size_t)g * cn + c) * 2]; s2 += stats[((size_t)g * cn + c) * 2 + 1]; }
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
}

// ---- 16-bit fast path: one thread = 8 consecutive channels (16 bytes) of one pixel -------------------------------------
// Used when C, c_begin and c_count are multiples of 8, so every 16-byte chunk is wholly inside or outside the
// normalised range.  Per-thread partial sums are fp32 over a bounded number of pixels (<= ~512, see launch_reduce_v8),
// combined in double like the scalar kernels.
template <typename T> struct ie_vec8 { typedef __attribute__((ext_vector_type(8))) T type; };

template <bool BWD, typename T>
__global__ __launch_bounds__(256) void norm_reduce_v8_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ yact,
                                                             const float* __restrict__ mean, const float* __restrict__ inv,
                                                             double* __restrict__ stats, int HW, int C, int c0, int nch, int cn_total, int c_off,
                                                             int n_mul, int act, float slope, const float* __restrict__ pg = nullptr,
                                                             const float* __restrict__ pb = nullptr, int pre = 0) {
    typedef typename ie_vec8<T>::type V;
    const int n = blockIdx.y;
    const int tpc = 256 / nch;
    const int ch = threadIdx.x % nch, ps = threadIdx.x / nch;
    const int g = n * n_mul;
    float s[8], ss[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; ss[k] = 0.f; }
    if (ps < tpc) {
        const size_t base = (size_t)n * HW * C + c0 + ch * 8;
        float m[8], iv[8], ga[8], be[8];
        if (BWD) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                m[k] = mean[(size_t)g * cn_total + c_off + ch * 8 + k];
                iv[k] = inv[(size_t)g * cn_total + c_off + ch * 8 + k];
                ga[k] = (pre && pg) ? pg[c_off + ch * 8 + k] : 1.f;
                be[k] = (pre && pg) ? pb[c_off + ch * 8 + k] : 0.f;
            }
        }
        for (int pix = blockIdx.x * tpc + ps; pix < HW; pix += gridDim.x * tpc) {
            const size_t i = base + (size_t)pix * C;
            const V xv = *(const V*)(x + i);
            if (BWD) {
                const V dv = *(const V*)(dy + i);
                V yv = xv;
                if (act != IE_ACT_NONE && !pre) yv = *(const V*)(yact + i);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float gr = (float)dv[k];
                    const float xh = ((float)xv[k] - m[k]) * iv[k];
                    if (act != IE_ACT_NONE) gr *= pre ? ie_act_grad_from_pre(fmaf(xh, ga[k], be[k]), act, slope) : ie_act_grad_from_output((float)yv[k], act, slope);
                    s[k] += gr;
                    ss[k] = fmaf(gr, xh, ss[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float v = (float)xv[k]; s[k] += v; ss[k] = fmaf(v, v, ss[k]); }
            }
        }
    }
    __shared__ float sh[16][256];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sh[k][threadIdx.x] = s[k]; sh[8 + k][threadIdx.x] = ss[k]; }
    __syncthreads();
    for (int o = threadIdx.x; o < nch * 16; o += 256) {
        const int stat = o / (nch * 8), cc = o % (nch * 8);
        double a = 0.0;
        for (int q = 0; q < tpc; ++q) a += (double)sh[stat * 8 + (cc & 7)][q * nch + (cc >> 3)];
        atomicAdd(&stats[((size_t)g * cn_total + c_off + cc) * 2 + stat], a);
    }
}

// Apply passes: thread = (pixel slot, 8-channel chunk) walking down the image, so the per-channel coefficients are
// fetched once per thread and the loop body is one 16-byte load (three in the backward pass), a few FMAs and one
// 16-byte store.  Grid: (pixel chunks, N, channel blocks of <= 256 chunks); nchb = chunks per channel block.
template <typename T>
__global__ __launch_bounds__(256) void norm_apply_v8_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ inv,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                                            int HW, int C, int c0, int cn, int n_mul, int act, float slope, int nchb) {
    typedef typename ie_vec8<T>::type V;
    const int n = blockIdx.y;
    const int tpc = 256 / nchb;
    const int c = (blockIdx.z * nchb + threadIdx.x % nchb) * 8, ps = threadIdx.x / nchb;
    if (ps >= tpc || c >= C) return;
    const bool in = c >= c0 && c < c0 + cn;
    float m[8], iv[8], ga[8], be[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        m[k] = 0.f; iv[k] = 1.f; ga[k] = 1.f; be[k] = 0.f;
        if (in) {
            const size_t idx = (size_t)n * n_mul * cn + (c - c0) + k;
            m[k] = mean[idx]; iv[k] = inv[idx];
            if (gamma) { ga[k] = gamma[c - c0 + k]; be[k] = beta[c - c0 + k]; }
        }
    }
    const size_t base = (size_t)n * HW * C + c;
    for (int pix = blockIdx.x * tpc + ps; pix < HW; pix += gridDim.x * tpc) {
        const size_t i = base + (size_t)pix * C;
        const V xv = *(const V*)(x + i);
        V out;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = (float)xv[k];
            if (in) { v = (v - m[k]) * iv[k]; if (gamma) v = v * ga[k] + be[k]; }
            out[k] = (T)ie_act_apply(v, act, slope);
        }
        *(V*)(y + i) = out;
    }
}

template <typename T>
__global__ __launch_bounds__(256, 4) void norm_backward_apply_v8_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ yact,
                                                                     const float* __restrict__ mean, const float* __restrict__ inv,
                                                                     const float* __restrict__ gamma, const double* __restrict__ stats,
                                                                     T* __restrict__ dx, int HW, int C, int c0, int cn, int n_mul, double count,
                                                                     int act, float slope, int nchb, const float* __restrict__ pb = nullptr, int pre = 0) {
    typedef typename ie_vec8<T>::type V;
    const int n = blockIdx.y;
    const int tpc = 256 / nchb;
    const int c = (blockIdx.z * nchb + threadIdx.x % nchb) * 8, ps = threadIdx.x / nchb;
    if (ps >= tpc || c >= C) return;
    const bool in = c >= c0 && c < c0 + cn;
    float m[8], iv[8], gi[8], m1[8], m2[8], ga[8], be[8];
    const double rcount = 1.0 / count;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        m[k] = 0.f; iv[k] = 1.f; gi[k] = 1.f; m1[k] = 0.f; m2[k] = 0.f; ga[k] = 1.f; be[k] = 0.f;
        if (in) {
            const size_t idx = (size_t)n * n_mul * cn + (c - c0) + k;
            m[k] = mean[idx]; iv[k] = inv[idx];
            ga[k] = gamma ? gamma[c - c0 + k] : 1.f;
            be[k] = (pre && pb) ? pb[c - c0 + k] : 0.f;
            gi[k] = ga[k] * iv[k];
            m1[k] = (float)(stats[2 * idx] * rcount); m2[k] = (float)(stats[2 * idx + 1] * rcount);
        }
    }
    const size_t base = (size_t)n * HW * C + c;
    for (int pix = blockIdx.x * tpc + ps; pix < HW; pix += gridDim.x * tpc) {
        const size_t i = base + (size_t)pix * C;
        const V dv = *(const V*)(dy + i);
        float gr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gr[k] = (float)dv[k];
        if (act != IE_ACT_NONE && !pre) {
            const V yv = *(const V*)(yact + i);
#pragma unroll
            for (int k = 0; k < 8; ++k) gr[k] *= ie_act_grad_from_output((float)yv[k], act, slope);
        }
        V out;
        if (in) {
            const V xv = *(const V*)(x + i);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh = ((float)xv[k] - m[k]) * iv[k];
                if (act != IE_ACT_NONE && pre) gr[k] *= ie_act_grad_from_pre(fmaf(xh, ga[k], be[k]), act, slope);
                out[k] = (T)(gi[k] * (gr[k] - m1[k] - xh * m2[k]));
            }
        } else {
            if (act != IE_ACT_NONE && pre) {      // outside the range the layer is the bare activation of x
                const V xv = *(const V*)(x + i);
#pragma unroll
                for (int k = 0; k < 8; ++k) gr[k] *= ie_act_grad_from_pre((float)xv[k], act, slope);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) out[k] = (T)gr[k];
        }
        *(V*)(dx + i) = out;
    }
}

// grid of the apply passes: ~8 pixels per thread, at most ~1024 workgroups
inline dim3 norm_apply_v8_grid(int N, int HW, int C, int& nchb) {
    const int C8 = C / 8;
    nchb = C8 < 256 ? C8 : 256;
    const int tpc = 256 / nchb;
    int chunks = ie_cdiv(HW, tpc * 8);
    static const int wg_cap = getenv("IE_NORM_GRID_CAP") ? atoi(getenv("IE_NORM_GRID_CAP")) : 1024;      // few, long-running workgroups (grid-stride) leave wave slots to the weight-gradient kernel on the side stream: 8192 -> 1024 = +1 % on the U-Net-64 step, 512 = -6 %
    const int cap = ie_cdiv(wg_cap, N);
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    return dim3(chunks, N, ie_cdiv(C8, nchb));
}

// the apply passes take the 8-channel form for fp32 tensors too (two 16-byte accesses per thread; same arithmetic as the scalar
// kernels); the statistics passes keep their per-element double accumulation for fp32 (the parity path)
inline bool norm_apply_v8_ok(int C, int c_begin, int c_count) { return (C & 7) == 0 && (c_begin & 7) == 0 && (c_count & 7) == 0; }
template <typename T> constexpr bool norm_v8_ok(int C, int c_begin, int c_count) {
    return sizeof(T) == 2 && (C & 7) == 0 && (c_begin & 7) == 0 && (c_count & 7) == 0;
}

template <bool BWD, typename T>
int launch_reduce_v8(const T* x, const T* dy, const T* yact, const float* mean, const float* inv, double* ws, int N, int HW,
                     int C, int c_begin, int c_count, int n_mul, int act, float slope, hipStream_t st, const float* pg = nullptr, const float* pb = nullptr, int pre = 0) {
    const int groups = n_mul ? N : 1;
    IE_HIP_CHECK(hipMemsetAsync(ws, 0, sizeof(double) * 2 * (size_t)groups * c_count, st));
    for (int off = 0; off < c_count; off += 2048) {      // <= 256 chunks of 8 channels per launch
        const int cn = (c_count - off) < 2048 ? (c_count - off) : 2048;
        const int nch = cn / 8, tpc = 256 / nch;
        int chunks = ie_cdiv(HW, tpc * 16);
        const int cap = n_mul ? 1024 : (N >= 16 ? 64 : 1024 / N);
        if (chunks > cap) chunks = cap;
        const int min_chunks = ie_cdiv(HW, tpc * 512);      // bound the fp32 partial sums to <= 512 terms per thread
        if (chunks < min_chunks) chunks = min_chunks;
        if (chunks < 1) chunks = 1;
        hipLaunchKernelGGL((norm_reduce_v8_kernel<BWD, T>), dim3(chunks, N), dim3(256), 0, st, x, dy, yact, mean, inv, ws, HW, C, c_begin + off,
                           nch, c_count, off, n_mul, act, slope, pg, pb, pre);
        IE_LAUNCH_CHECK();
    }
    return IE_OK;
}

template <bool BWD, typename T>
int launch_reduce(const T* x, const T* dy, const T* yact, const float* mean, const float* inv, double* ws, int N, int HW,
                  int C, int c_begin, int c_count, int n_mul, int act, float slope, hipStream_t st, const float* pg = nullptr, const float* pb = nullptr, int pre = 0) {
    if constexpr (sizeof(T) == 2) {
        if (norm_v8_ok<T>(C, c_begin, c_count)) return launch_reduce_v8<BWD, T>(x, dy, yact, mean, inv, ws, N, HW, C, c_begin, c_count, n_mul, act, slope, st, pg, pb, pre);
    }
    const int groups = n_mul ? N : 1;
    IE_HIP_CHECK(hipMemsetAsync(ws, 0, sizeof(double) * 2 * (size_t)groups * c_count, st));
    for (int off = 0; off < c_count; off += 256) {     // <= 256 channels per launch (one thread column each)
        const int cn = (c_count - off) < 256 ? (c_count - off) : 256;
        const int tpc = 256 / cn > 0 ? 256 / cn : 1;
        int chunks = ie_cdiv(HW, tpc * 64);
        const int cap = n_mul ? 1024 : (N >= 16 ? 64 : 1024 / N);
        if (chunks > cap) chunks = cap;
        if (chunks < 1) chunks = 1;
        hipLaunchKernelGGL((norm_reduce_kernel<BWD, T>), dim3(chunks, N), dim3(256), 0, st, x, dy, yact, mean, inv, ws, HW, C, c_begin + off,
                           cn, c_count, off, n_mul, act, slope, pg, pb, pre);
        IE_LAUNCH_CHECK();
    }
    return IE_OK;
}

}  // namespace

template <typename T>
static int norm_forward_impl(const T* x, const float* gamma, const float* beta, T* y, double* ws, float* save_mean, float* save_invstd,
                             float* running_mean, float* running_var, float momentum, int per_sample, int N, int H, int W, int C, int c_begin,
                             int c_count, float eps, int act, float slope, void* stream) {
    if (!x || !y || !ws || !save_mean || !save_invstd || N <= 0 || H <= 0 || W <= 0 || C <= 0 || c_begin < 0 || c_count <= 0 ||
        c_begin + c_count > C || (gamma == nullptr) != (beta == nullptr) || (running_mean == nullptr) != (running_var == nullptr) ||
        (per_sample && running_mean))
        IE_FAIL(IE_ERR_INVALID, "ie_norm_forward_nhwc: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W, n_mul = per_sample ? 1 : 0, groups = per_sample ? N : 1;
    int rc = launch_reduce<false, T>(x, nullptr, nullptr, nullptr, nullptr, ws, N, HW, C, c_begin, c_count, n_mul, 0, 0.f, st);
    if (rc != IE_OK) return rc;
    const double count = per_sample ? (double)HW : (double)N * HW;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(nblk((size_t)groups * c_count)), dim3(256), 0, st, ws, save_mean, save_invstd,
                       groups * c_count, count, eps, running_mean, running_var, momentum);
    IE_LAUNCH_CHECK();
    const size_t total = (size_t)N * HW * C;
    if (norm_apply_v8_ok(C, c_begin, c_count)) {
        int nchb;
        const dim3 grid = norm_apply_v8_grid(N, HW, C, nchb);
        hipLaunchKernelGGL(norm_apply_v8_kernel<T>, grid, dim3(256), 0, st, x, save_mean, save_invstd, gamma, beta, y, HW, C, c_begin, c_count,
                           n_mul, act, slope, nchb);
        IE_LAUNCH_CHECK();
        return IE_OK;
    }
    hipLaunchKernelGGL(norm_apply_kernel<T>, dim3(nblk(total)), dim3(256), 0, st, x, save_mean, save_invstd, gamma, beta, y, HW, C, c_begin,
                       c_count, n_mul, act, slope, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

template <typename T>
static int norm_backward_impl(const T* x, const T* dy, const T* y, const float* gamma, const float* save_mean, const float* save_invstd, double* ws,
                              T* dx, float* dgamma, float* dbeta, int accumulate_param_grads, int per_sample, int N, int H, int W, int C,
                              int c_begin, int c_count, int act, float slope, void* stream, const float* beta = nullptr, int pre = 0) {
    if (!x || !dy || !save_mean || !save_invstd || !ws || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || c_begin < 0 || c_count <= 0 ||
        c_begin + c_count > C || (act != IE_ACT_NONE && !y && !pre) || (dgamma == nullptr) != (dbeta == nullptr))
        IE_FAIL(IE_ERR_INVALID, "ie_norm_backward_nhwc: bad args");
    if (pre && ((gamma == nullptr) != (beta == nullptr) || (act != IE_ACT_NONE && act != IE_ACT_RELU && act != IE_ACT_LRELU)))
        IE_FAIL(IE_ERR_INVALID, "ie_norm_backward_pre_nhwc: gamma and beta go together; none / ReLU / LeakyReLU only (the derivative must be a function of the pre-activation's sign)");
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W, n_mul = per_sample ? 1 : 0, groups = per_sample ? N : 1;
    int rc = launch_reduce<true, T>(x, dy, y, save_mean, save_invstd, ws, N, HW, C, c_begin, c_count, n_mul, act, slope, st, gamma, beta, pre);
    if (rc != IE_OK) return rc;
    const double count = per_sample ? (double)HW : (double)N * HW;
    const size_t total = (size_t)N * HW * C;
    if (norm_apply_v8_ok(C, c_begin, c_count)) {
        int nchb;
        const dim3 grid = norm_apply_v8_grid(N, HW, C, nchb);
        hipLaunchKernelGGL(norm_backward_apply_v8_kernel<T>, grid, dim3(256), 0, st, x, dy, y, save_mean, save_invstd, gamma, ws, dx, HW, C,
                           c_begin, c_count, n_mul, count, act, slope, nchb, beta, pre);
    } else
        hipLaunchKernelGGL(norm_backward_apply_kernel<T>, dim3(nblk(total)), dim3(256), 0, st, x, dy, y, save_mean, save_invstd, gamma, ws, dx,
                           HW, C, c_begin, c_count, n_mul, count, act, slope, total, beta, pre);
    IE_LAUNCH_CHECK();
    if (dgamma) {
        hipLaunchKernelGGL(norm_param_grad_kernel, dim3(nblk(c_count)), dim3(256), 0, st, ws, dgamma, dbeta, groups, c_count,
                           accumulate_param_grads);
        IE_LAUNCH_CHECK();
    }
    return IE_OK;
}

IE_EXPORT int ie_norm_forward_nhwc(const void* x, const float* gamma, const float* beta, void* y, int dtype, double* ws, float* save_mean,
                                   float* save_invstd, float* running_mean, float* running_var, float momentum, int per_sample, int N, int H,
                                   int W, int C, int c_begin, int c_count, float eps, int act, float slope, void* stream) {
    switch (dtype) {
    case IE_F32: return norm_forward_impl((const float*)x, gamma, beta, (float*)y, ws, save_mean, save_invstd, running_mean, running_var, momentum, per_sample, N, H, W, C, c_begin, c_count, eps, act, slope, stream);
    case IE_F16: return norm_forward_impl((const _Float16*)x, gamma, beta, (_Float16*)y, ws, save_mean, save_invstd, running_mean, running_var, momentum, per_sample, N, H, W, C, c_begin, c_count, eps, act, slope, stream);
    case IE_BF16: return norm_forward_impl((const __bf16*)x, gamma, beta, (__bf16*)y, ws, save_mean, save_invstd, running_mean, running_var, momentum, per_sample, N, H, W, C, c_begin, c_count, eps, act, slope, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_norm_forward_nhwc: bad dtype");
    }
}

IE_EXPORT int ie_norm_backward_nhwc(const void* x, const void* dy, const void* y, int dtype, const float* gamma, const float* save_mean,
                                    const float* save_invstd, double* ws, void* dx, float* dgamma, float* dbeta, int accumulate_param_grads,
                                    int per_sample, int N, int H, int W, int C, int c_begin, int c_count, int act, float slope, void* stream) {
    switch (dtype) {
    case IE_F32: return norm_backward_impl((const float*)x, (const float*)dy, (const float*)y, gamma, save_mean, save_invstd, ws, (float*)dx, dgamma, dbeta, accumulate_param_grads, per_sample, N, H, W, C, c_begin, c_count, act, slope, stream);
    case IE_F16: return norm_backward_impl((const _Float16*)x, (const _Float16*)dy, (const _Float16*)y, gamma, save_mean, save_invstd, ws, (_Float16*)dx, dgamma, dbeta, accumulate_param_grads, per_sample, N, H, W, C, c_begin, c_count, act, slope, stream);
    case IE_BF16: return norm_backward_impl((const __bf16*)x, (const __bf16*)dy, (const __bf16*)y, gamma, save_mean, save_invstd, ws, (__bf16*)dx, dgamma, dbeta, accumulate_param_grads, per_sample, N, H, W, C, c_begin, c_count, act, slope, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_norm_backward_nhwc: bad dtype");
    }
}

IE_EXPORT int ie_norm_backward_pre_nhwc(const void* x, const void* dy, int dtype, const float* gamma, const float* beta, const float* save_mean,
                                        const float* save_invstd, double* ws, void* dx, float* dgamma, float* dbeta, int accumulate_param_grads,
                                        int per_sample, int N, int H, int W, int C, int c_begin, int c_count, int act, float slope, void* stream) {
    switch (dtype) {
    case IE_F32: return norm_backward_impl((const float*)x, (const float*)dy, (const float*)nullptr, gamma, save_mean, save_invstd, ws, (float*)dx, dgamma, dbeta, accumulate_param_grads, per_sample, N, H, W, C, c_begin, c_count, act, slope, stream, beta, 1);
    case IE_F16: return norm_backward_impl((const _Float16*)x, (const _Float16*)dy, (const _Float16*)nullptr, gamma, save_mean, save_invstd, ws, (_Float16*)dx, dgamma, dbeta, accumulate_param_grads, per_sample, N, H, W, C, c_begin, c_count, act, slope, stream, beta, 1);
    case IE_BF16: return norm_backward_impl((const __bf16*)x, (const __bf16*)dy, (const __bf16*)nullptr, gamma, save_mean, save_invstd, ws, (__bf16*)dx, dgamma, dbeta, accumulate_param_grads, per_sample, N, H, W, C, c_begin, c_count, act, slope, stream, beta, 1);
    default: IE_FAIL(IE_ERR_INVALID, "ie_norm_backward_pre_nhwc: bad dtype");
    }
}

IE_EXPORT int ie_norm_forward_nhwc_f32(const float* x, const float* gamma, const float* beta, float* y, double* ws, float* save_mean,
                                       float* save_invstd, float* running_mean, float* running_var, float momentum, int per_sample,
                                       int N, int H, int W, int C, int c_begin, int c_count, float eps, int act, float slope,
                                       void* stream) {
    return ie_norm_forward_nhwc(x, gamma, beta, y, IE_F32, ws, save_mean, save_invstd, running_mean, running_var, momentum, per_sample, N, H, W, C,
                                c_begin, c_count, eps, act, slope, stream);
}

IE_EXPORT int ie_instnorm_nhwc_f32(const float* x, const float* gamma, const float* beta, float* y, double* stats_ws, int N, int H,
                                   int W, int C, int c_begin, int c_count, float eps, int act, float slope, void* stream) {
    if (!stats_ws || N <= 0 || c_count <= 0) IE_FAIL(IE_ERR_INVALID, "ie_instnorm_nhwc_f32: bad args");
    float* mean = (float*)(stats_ws + 2 * (size_t)N * c_count);
    return ie_norm_forward_nhwc_f32(x, gamma, beta, y, stats_ws, mean, mean + (size_t)N * c_count, nullptr, nullptr, 0.f, 1, N, H, W, C,
                                    c_begin, c_count, eps, act, slope, stream);
}

IE_EXPORT int ie_norm_backward_nhwc_f32(const float* x, const float* dy, const float* y, const float* gamma, const float* save_mean,
                                        const float* save_invstd, double* ws, float* dx, float* dgamma, float* dbeta,
                                        int accumulate_param_grads, int per_sample, int N, int H, int W, int C, int c_begin,
                                        int c_count, int act, float slope, void* stream) {
    return ie_norm_backward_nhwc(x, dy, y, IE_F32, gamma, save_mean, save_invstd, ws, dx, dgamma, dbeta, accumulate_param_grads, per_sample, N, H, W,
                                 C, c_begin, c_count, act, slope, stream);
}
