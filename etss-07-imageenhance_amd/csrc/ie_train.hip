// Training path, fp32 NHWC: convolution weight / data gradients, backward of the
// streaming ops (max-pool, resize, pad, pixel-shuffle, activations), L1 and SSIM losses with their
// gradients, global gradient norm and the Adam update on a flat parameter buffer.
// Reference: torch.autograd over the layers cited in include/ie_hip.h; loss definitions
// src/mon/nn/loss/base.py:164-178 (L1Loss), src/mon/nn/loss/image.py:834-866 (SSIMLoss),
// src/mon/nn/metric/image.py:90-237 (custom_ssim); optimiser: torch.optim.Adam + clip_grad_norm_
// (src/mon/nn/model.py:498-615, zero_dce_re_sice_mix.py:46-51,106).
#include "ie_common.h"
#include "ie_conv_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

// ---- conv weight gradient -------------------------------------------------------------------------------------
// dW[tap][cbase+ci][co] += sum over output pixels of X[pixel shifted by tap][ci] * dY[pixel][co]: a GEMM with
// K = pixels.  Workgroup = 64 ci x 64 co tile of one tap over a chunk of output rows (split-K, fp32 atomics at
// the end); 256 threads x (4 ci x 4 co) register tile; 16 pixels per LDS stage.
constexpr int WG_T = 64, WG_K = 16;

struct WgradParams {
    const void* x; const void* dy; float* dw;
    int N, H, W, Cs, Cout, Cin_total, cbase, kh, kw, stride, pad, pad_mode, Ho, Wo, rows_per_wg, ci_tiles;
    float* slab;       // deterministic mode: [chunk][tile = blockIdx.y * gridDim.x + blockIdx.x][64 ci][64 co] partial sums, else NULL
};

// 4 consecutive channels of an fp32 / 16-bit NHWC tensor as fp32
template <typename T> __device__ __forceinline__ ie_f32x4 ld4(const T* q) {
    if constexpr (sizeof(T) == 4) return *(const ie_f32x4*)q;
    else {
        typedef __attribute__((ext_vector_type(4))) T t4;
        const t4 v = *(const t4*)q;
        return ie_f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    __shared__ float xs[WG_K][WG_T + 4];
    __shared__ float ds[WG_K][WG_T + 4];
    const int t = threadIdx.x;
    const int co_t = blockIdx.x * WG_T;
    const int ci_t = (blockIdx.y % p.ci_tiles) * WG_T;
    const int tap = blockIdx.y / p.ci_tiles;
    const int ti = tap / p.kw, tj = tap % p.kw;
    const int total_rows = p.N * p.Ho;
    const int r0 = blockIdx.z * p.rows_per_wg, r1 = min(r0 + p.rows_per_wg, total_rows);
    const int lk = t >> 4, lc = (t & 15) * 4;        // loader: pixel slot, first of 4 channels
    const int tci = (t >> 4) * 4, tco = (t & 15) * 4; // compute tile origin
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    const bool vec_x = (p.Cs & 3) == 0, vec_d = (p.Cout & 3) == 0;
    for (int r = r0; r < r1; ++r) {
        const int n = r / p.Ho, ho = r % p.Ho;
        int hi = ho * p.stride + ti - p.pad;
        if (p.pad_mode == IE_PAD_REPLICATE) hi = min(max(hi, 0), p.H - 1);
        else if (hi < 0 || hi >= p.H) continue;          // uniform over the workgroup
        const T* xrow = (const T*)p.x + ((size_t)n * p.H + hi) * p.W * p.Cs;
        const T* drow = (const T*)p.dy + ((size_t)n * p.Ho + ho) * p.Wo * p.Cout;
        for (int w0 = 0; w0 < p.Wo; w0 += WG_K) {
            const int wo = w0 + lk;
            int wi = wo * p.stride + tj - p.pad;
            bool okx = wo < p.Wo;
            if (p.pad_mode == IE_PAD_REPLICATE) wi = min(max(wi, 0), p.W - 1);
            else okx = okx && wi >= 0 && wi < p.W;
            ie_f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vd = {0.f, 0.f, 0.f, 0.f};
            if (okx) {
                const T* q = xrow + (size_t)wi * p.Cs + ci_t + lc;
                if (vec_x && ci_t + lc + 3 < p.Cs) vx = ld4<T>(q);
                else
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (ci_t + lc + k < p.Cs) vx[k] = (float)q[k];
            }
            if (wo < p.Wo) {
                const T* q = drow + (size_t)wo * p.Cout + co_t + lc;
                if (vec_d && co_t + lc + 3 < p.Cout) vd = ld4<T>(q);
                else
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (co_t + lc + k < p.Cout) vd[k] = (float)q[k];
            }
            __syncthreads();
            *(ie_f32x4*)&xs[lk][lc] = vx;
            *(ie_f32x4*)&ds[lk][lc] = vd;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < WG_K; ++kk) {
                const ie_f32x4 a = *(const ie_f32x4*)&xs[kk][tci];
                const ie_f32x4 b = *(const ie_f32x4*)&ds[kk][tco];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
            }
        }
    }
    if (p.slab) {       // deterministic mode: park the whole 64 x 64 partial tile, wgrad_reduce_kernel adds the chunks in order
        float* sl = p.slab + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (WG_T * WG_T);
#pragma unroll
        for (int i = 0; i < 4; ++i) *(ie_f32x4*)(sl + (tci + i) * WG_T + tco) = ie_f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = ci_t + tci + i;
        if (ci >= p.Cs) continue;
        float* row = p.dw + ((size_t)tap * p.Cin_total + p.cbase + ci) * p.Cout;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co_t + tco + j;
            if (co < p.Cout && acc[i][j] != 0.f) atomicAdd(row + co, acc[i][j]);
        }
    }
}

// deterministic mode, second stage: dW element += sum over the chunks' slabs in chunk order (fp64 accumulate, one rounding)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int chunks, int co_tiles, int ci_tiles, int taps,
                                    int Cs, int Cout, int Cin_total, int cbase) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t tiles = (size_t)co_tiles * ci_tiles * taps;
    if (e >= tiles * WG_T * WG_T) return;
    const int tile = (int)(e / (WG_T * WG_T)), r = (int)(e % (WG_T * WG_T));
    const int cot = tile % co_tiles, y = tile / co_tiles, cit = y % ci_tiles, tap = y / ci_tiles;
    const int ci = cit * WG_T + r / WG_T, co = cot * WG_T + r % WG_T;
    if (ci >= Cs || co >= Cout) return;
    double s = 0.0;
    for (int z = 0; z < chunks; ++z) s += (double)slab[((size_t)z * tiles + tile) * (WG_T * WG_T) + r];
    float* d = dw + ((size_t)tap * Cin_total + cbase + ci) * Cout + co;
    *d = (float)((double)*d + s);
}

// ---- conv data gradient ----------------------------------------------------------------------------------------
// dX[n,hi,wi,ci] = sum_{i,j,co} dY[n,ho,wo,co] * W[i][j][cbase+ci][co] with ho*stride + i - pad = hi (zero padding).
// lane = input pixel, wave = 8 input channels (weights wave-uniform -> scalar loads), like the forward kernel.
constexpr int DG_CT = 8;
struct DgradParams {
    const float* dy; const float* w; float* dx;
    int N, H, W, Cs, Cout, Cin_total, cbase, kh, kw, stride, pad, Ho, Wo, accumulate;
};

__global__ __launch_bounds__(256) void conv_dgrad_kernel(DgradParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.z;
    const int ipix = blockIdx.x * 64 + lane;
    const int ci0 = (blockIdx.y * 4 + wave) * DG_CT;
    if (ci0 >= p.Cs) return;
    const bool live = ipix < p.H * p.W;
    const int hi = live ? ipix / p.W : 0, wi = live ? ipix % p.W : 0;
    const int nci = min(DG_CT, p.Cs - ci0);
    float acc[DG_CT];
#pragma unroll
    for (int k = 0; k < DG_CT; ++k) acc[k] = 0.f;
    const bool vec = (p.Cout & 3) == 0;
    for (int i = 0; i < p.kh; ++i)
        for (int j = 0; j < p.kw; ++j) {
            const int hn = hi + p.pad - i, wn = wi + p.pad - j;
            const int ho = hn / p.stride, wo = wn / p.stride;
            const bool ok = live && hn >= 0 && wn >= 0 && hn % p.stride == 0 && wn % p.stride == 0 && ho < p.Ho && wo < p.Wo;
            const float* dp = p.dy + (((size_t)n * p.Ho + (ok ? ho : 0)) * p.Wo + (ok ? wo : 0)) * p.Cout;
            const float* wt = p.w + ((size_t)(i * p.kw + j) * p.Cin_total + p.cbase + ci0) * p.Cout;
            if (vec) {
                for (int co = 0; co < p.Cout; co += 4) {
                    const ie_f32x4 v = ok ? *(const ie_f32x4*)(dp + co) : ie_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < DG_CT; ++k)
                        if (k < nci) {
                            const float* wr = wt + (size_t)k * p.Cout + co;
                            acc[k] = fmaf(v[0], wr[0], acc[k]); acc[k] = fmaf(v[1], wr[1], acc[k]);
                            acc[k] = fmaf(v[2], wr[2], acc[k]); acc[k] = fmaf(v[3], wr[3], acc[k]);
                        }
                }
            } else {
                for (int co = 0; co < p.Cout; ++co) {
                    const float v = ok ? dp[co] : 0.f;
#pragma unroll
                    for (int k = 0; k < DG_CT; ++k) if (k < nci) acc[k] = fmaf(v, wt[(size_t)k * p.Cout + co], acc[k]);
                }
            }
        }
    if (!live) return;
    float* o = p.dx + (((size_t)n * p.H + hi) * p.W + wi) * p.Cs + ci0;
#pragma unroll
    for (int k = 0; k < DG_CT; ++k) if (k < nci) o[k] = (p.accumulate ? o[k] : 0.f) + acc[k];
}

// W'[(kh-1-i)*kw + (kw-1-j)][co][ci] = W[i*kw + j][ci][co]: the data gradient of a stride-1 conv is a conv of dY with W'
__global__ void flip_transpose_hwio_kernel(const float* __restrict__ w, float* __restrict__ wt, int taps, int Cin, int Cout, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % Cin);
    size_t r = i / Cin;
    const int co = (int)(r % Cout);
    const int tp = (int)(r / Cout);
    wt[i] = w[((size_t)(taps - 1 - tp) * Cin + ci) * Cout + co];
}

// per-channel sum over pixels (bias gradient): same thread layout as the norm statistics pass
template <typename T>
__global__ void channel_sum_kernel(const T* __restrict__ x, double* __restrict__ acc, size_t npix, int C, int c0, int cn) {
    const int tpc = max(1, 256 / cn);
    const int c = threadIdx.x % cn, ps = threadIdx.x / cn;
    double s = 0.0;
    if (ps < tpc)
        for (size_t pix = (size_t)blockIdx.x * tpc + ps; pix < npix; pix += (size_t)gridDim.x * tpc) s += (float)x[pix * C + c0 + c];
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (ps == 0) {
        for (int k = 1; k < tpc; ++k) s += sh[k * cn + c];
        atomicAdd(&acc[c0 + c], s);
    }
}
// 16-bit tensors with C % 8 == 0: thread = (pixel slot, 8-channel chunk), 16-byte loads; fp32 partial sums over a bounded number
// of pixels per thread (launch: <= ~512), combined in double
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_v8_kernel(const T* __restrict__ x, double* __restrict__ acc, size_t npix, int C, int c0, int nch) {
    typedef __attribute__((ext_vector_type(8))) T V;
    const int tpc = 256 / nch;
    const int ch = threadIdx.x % nch, ps = threadIdx.x / nch;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    if (ps < tpc)
        for (size_t pix = (size_t)blockIdx.x * tpc + ps; pix < npix; pix += (size_t)gridDim.x * tpc) {
            const V v = *(const V*)(x + pix * C + c0 + ch * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += (float)v[k];
        }
    __shared__ float sh[8][256];
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int o = threadIdx.x; o < nch * 8; o += 256) {
        double a = 0.0;
        for (int q = 0; q < tpc; ++q) a += (double)sh[o & 7][q * nch + (o >> 3)];
        atomicAdd(&acc[c0 + o], a);
    }
}
__global__ void store_f64_as_f32_kernel(const double* __restrict__ a, float* y, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (accumulate ? y[i] : 0.f) + (float)a[i];
}

// ---- streaming backward ops ---------------------------------------------------------------------------------------
// MaxPool2d(2,2): gradient to the first maximum of the window in scan order (torch keeps the first index).
// V consecutive channels per thread (8 for 16-bit tensors whose channel count allows: 16-byte accesses).
template <typename T, int V>
__global__ void maxpool2_backward_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int H,
                                         int W, int C, int Ho, int Wo, int accumulate, size_t total) {
    typedef __attribute__((ext_vector_type(V))) T TV;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int CV = C / V;
    const int c = (int)(i % CV) * V;
    size_t r = i / CV;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const size_t n = r / H;
    float g[V];
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = 0.f;
    const int ho = h >> 1, wo = w >> 1;
    if (ho < Ho && wo < Wo) {
        const T* p = x + ((n * H + 2 * ho) * W + 2 * wo) * C + c;
        TV v[4];
        v[0] = *(const TV*)p; v[1] = *(const TV*)(p + C); v[2] = *(const TV*)(p + (size_t)W * C); v[3] = *(const TV*)(p + (size_t)W * C + C);
        const TV d = *(const TV*)(dy + ((n * Ho + ho) * Wo + wo) * C + c);
        const int me = (h & 1) * 2 + (w & 1);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] = (float)v[k][e];
            int best = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k) if (f[k] > f[best] || (f[k] != f[k] && f[best] == f[best])) best = k;
            if (best == me) g[e] = (float)d[e];
        }
    }
    T* o = dx + ((n * H + h) * W + w) * C + c;
    TV out;
    if (accumulate) {
        const TV old = *(const TV*)o;
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (T)((float)old[e] + g[e]);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (T)g[e];
    }
    *(TV*)o = out;
}

// source coordinates of destination index o (same float expressions as the forward kernel in ie_ops.hip)
__device__ __forceinline__ void resize_src(int o, int In, int Out, int mode, int& i0, int& i1, float& l) {
    if (mode == 0) {
        i0 = i1 = min((int)floorf(o * ((float)In / (float)Out)), In - 1); l = 0.f; return;
    }
    float s;
    if (mode == 2) s = (Out > 1 ? (float)(In - 1) / (float)(Out - 1) : 0.f) * o;
    else s = fmaxf((float)In / (float)Out * (o + 0.5f) - 0.5f, 0.f);
    i0 = min((int)s, In - 1); i1 = min(i0 + 1, In - 1); l = s - (float)i0;
}
// outputs o that may read input i: those whose first source index i0(o) is i - 1 or i.  i0 is monotone in o, so they form one range; its ends
// come from inverting the source coordinate of the mode, padded by one output each side against rounding (membership is re-checked exactly).
// A x2 resize: 8 candidates per axis instead of the 11 of the mode-blind bound -- the adjoint is bound by this candidate arithmetic.
__device__ __forceinline__ void resize_candidates(int i, int In, int Out, int mode, int& lo, int& hi) {
    const float sc = (float)Out / (float)In;
    float flo, fhi;
    if (mode == 0) { flo = i * sc; fhi = (i + 1) * sc; }
    else if (mode == 2) {
        if (In <= 1 || Out <= 1) { lo = 0; hi = Out - 1; return; }
        const float inv = (float)(Out - 1) / (float)(In - 1);
        flo = (i - 1) * inv; fhi = (i + 1) * inv;
    } else { flo = (i - 0.5f) * sc - 0.5f; fhi = (i + 1.5f) * sc - 0.5f; }
    lo = max(0, (int)floorf(flo) - 1);
    hi = min(Out - 1, (int)ceilf(fhi) + 1);
}
// gather form of the resize adjoint: dx[i] = sum over the outputs that read input i of weight * dy[o]; V channels per thread
template <typename T, int V>
__global__ void resize_backward_kernel(const T* __restrict__ dy, T* __restrict__ dx, int Hi, int Wi, int C, int Ho, int Wo,
                                       int mode, int accumulate, size_t total) {
    typedef __attribute__((ext_vector_type(V))) T TV;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int CV = C / V;
    const int c = (int)(i % CV) * V;
    size_t r = i / CV;
    const int xi = (int)(r % Wi); r /= Wi;
    const int yi = (int)(r % Hi);
    const size_t n = r / Hi;
    int ylo, yhi, xlo, xhi;
    resize_candidates(yi, Hi, Ho, mode, ylo, yhi);
    resize_candidates(xi, Wi, Wo, mode, xlo, xhi);
    const T* p = dy + n * (size_t)Ho * Wo * C + c;
    float g[V];
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    for (int yo = ylo; yo <= yhi; ++yo) {
        int a0, a1; float ly;
        resize_src(yo, Hi, Ho, mode, a0, a1, ly);
        float wy = 0.f;
        if (mode == 0) wy = (a0 == yi) ? 1.f : 0.f;
        else { if (a0 == yi) wy += 1.f - ly; if (a1 == yi) wy += ly; }
        if (wy == 0.f) continue;
        float rowacc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) rowacc[e] = 0.f;
        for (int xo = xlo; xo <= xhi; ++xo) {
            int b0, b1; float lx;
            resize_src(xo, Wi, Wo, mode, b0, b1, lx);
            float wx = 0.f;
            if (mode == 0) wx = (b0 == xi) ? 1.f : 0.f;
            else { if (b0 == xi) wx += 1.f - lx; if (b1 == xi) wx += lx; }
            if (wx != 0.f) {
                const T* q = p + ((size_t)yo * Wo + xo) * C;
                const TV d = *(const TV*)q;
#pragma unroll
                for (int e = 0; e < V; ++e) rowacc[e] = fmaf(wx, (float)d[e], rowacc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = fmaf(wy, rowacc[e], g[e]);
    }
    T* o = dx + ((n * Hi + yi) * Wi + xi) * C + c;
    TV out;
    if (accumulate) {
        const TV old = *(const TV*)o;
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (T)((float)old[e] + g[e]);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (T)g[e];
    }
    *(TV*)o = out;
}

// adjoint of zero padding: dx[h,w] = dy[h+top, w+left]
template <typename T>
__global__ void crop_nhwc_kernel(const T* __restrict__ src, T* __restrict__ dst, int Hs, int Ws, int C, int top, int left,
                                 int H, int W, int accumulate, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int w = (int)(q % W); q /= W;
    const int h = (int)(q % H);
    const size_t n = q / H;
    dst[i] = (T)((accumulate ? (float)dst[i] : 0.f) + (float)src[((n * Hs + h + top) * Ws + w + left) * C + c]);
}

// a tile of a spatial torch.cat (mprnet.py:484-505: patches joined along H or W): dst[h+top, w+left] = src[h, w]
template <typename T>
__global__ void paste_nhwc_kernel(const T* __restrict__ src, T* __restrict__ dst, int Hd, int Wd, int C, int top, int left, int H, int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int w = (int)(q % W); q /= W;
    const int h = (int)(q % H);
    const size_t n = q / H;
    dst[((n * Hd + h + top) * Wd + w + left) * C + c] = src[i];
}

// adjoint of pixel shuffle: dx[n,h,w,c*r*r+i*r+j] = dy[n,h*r+i,w*r+j,c]
template <typename T>
__global__ void pixel_unshuffle_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int Co, int r,
                                       int accumulate, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % (Co * r * r));
    size_t q = i / (Co * r * r);
    const int w = (int)(q % W); q /= W;
    const int h = (int)(q % H);
    const size_t n = q / H;
    const int c = ch / (r * r), ii = (ch / r) % r, jj = ch % r;
    dx[i] = (T)((accumulate ? (float)dx[i] : 0.f) + (float)dy[((n * H * r + h * r + ii) * (size_t)(W * r) + w * r + jj) * Co + c]);
}

// dx = dy * act'(.) from the activation OUTPUT y; mode 5: clamp(x, lo, hi) from its INPUT x (pass where lo <= x <= hi).
// V elements per thread (8 for 16-bit tensors whose length allows: 16-byte accesses); dx may alias dy.
template <typename T, int V>
__global__ void act_backward_kernel(const T* dy, const T* __restrict__ ref, T* dx, int act, float slope, float lo, float hi, int accumulate,
                                    size_t total) {
    typedef __attribute__((ext_vector_type(V))) T TV;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const TV g = *(const TV*)(dy + i * V), r = *(const TV*)(ref + i * V);
    TV old;
    if (accumulate) old = *(const TV*)(dx + i * V);
    TV out;
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const float v = (float)r[e];
        const float d = act == 5 ? ((v >= lo && v <= hi) ? 1.f : 0.f) : ie_act_grad_from_output(v, act, slope);
        out[e] = (T)((accumulate ? (float)old[e] : 0.f) + (float)g[e] * d);
    }
    *(TV*)(dx + i * V) = out;
}

// ---- losses ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v) {
    __shared__ double sh[256];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    return sh[0];
}

// sum |a - b| (double accumulator); optional gradient grad (+)= gscale * sign(a - b)
__global__ void l1_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ acc, float* grad,
                          float gscale, int accumulate, size_t total) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        s += fabsf(d);
        if (grad) grad[i] = (accumulate ? grad[i] : 0.f) + (d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f));
    }
    s = block_sum(s);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}

// SSIM (metric/image.py:143-195): 11-tap Gaussian (sigma 1.5), separable VALID filtering of x, y, x^2, y^2, xy.
// Tile = 32 x 16 outputs of one (n, c) plane; element (n,c,h,w) lives at n*sn + c*sc + h*sh + w*sw (any layout).
constexpr int SS_TW = 32, SS_TH = 16, SS_R = 10, SS_K = 11;
struct SsimParams {
    const float* x; const float* y;
    long sn, sc, sh, sw;
    int N, C, H, W;
    float c1, c2;
    double* acc;            // [N*C] sum of the ssim map
    float* d1; float* d2; float* d3;   // optional (N*C planes of (H-10) x (W-10)): dS/dmu1, dS/dE[x^2], dS/dE[xy]
    float g[SS_K];
};

__global__ __launch_bounds__(256) void ssim_forward_kernel(SsimParams p) {
    __shared__ float sx[SS_TH + SS_R][SS_TW + SS_R + 1];
    __shared__ float sy[SS_TH + SS_R][SS_TW + SS_R + 1];
    __shared__ float hf[5][SS_TH + SS_R][SS_TW + 1];
    const int plane = blockIdx.z, n = plane / p.C, c = plane % p.C;
    const int Hv = p.H - SS_R, Wv = p.W - SS_R;
    const int ox = blockIdx.x * SS_TW, oy = blockIdx.y * SS_TH;
    const float* xb = p.x + n * p.sn + c * p.sc;
    const float* yb = p.y + n * p.sn + c * p.sc;
    for (int i = threadIdx.x; i < (SS_TH + SS_R) * (SS_TW + SS_R); i += 256) {
        const int r = i / (SS_TW + SS_R), q = i % (SS_TW + SS_R);
        const int h = oy + r, w = ox + q;
        const bool ok = h < p.H && w < p.W;
        sx[r][q] = ok ? xb[h * p.sh + w * p.sw] : 0.f;
        sy[r][q] = ok ? yb[h * p.sh + w * p.sw] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (SS_TH + SS_R) * SS_TW; i += 256) {      // filter along W
        const int r = i / SS_TW, q = i % SS_TW;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float u = sx[r][q + k], v = sy[r][q + k], gk = p.g[k];
            a = fmaf(gk, u, a); b = fmaf(gk, v, b); aa = fmaf(gk, u * u, aa); bb = fmaf(gk, v * v, bb); ab = fmaf(gk, u * v, ab);
        }
        hf[0][r][q] = a; hf[1][r][q] = b; hf[2][r][q] = aa; hf[3][r][q] = bb; hf[4][r][q] = ab;
    }
    __syncthreads();
    double s = 0.0;
    for (int i = threadIdx.x; i < SS_TH * SS_TW; i += 256) {                // filter along H, SSIM map
        const int r = i / SS_TW, q = i % SS_TW;
        const int h = oy + r, w = ox + q;
        if (h >= Hv || w >= Wv) continue;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float gk = p.g[k];
            m1 = fmaf(gk, hf[0][r + k][q], m1); m2 = fmaf(gk, hf[1][r + k][q], m2); e11 = fmaf(gk, hf[2][r + k][q], e11);
            e22 = fmaf(gk, hf[3][r + k][q], e22); e12 = fmaf(gk, hf[4][r + k][q], e12);
        }
        const float A1 = 2.f * m1 * m2 + p.c1, A2 = 2.f * (e12 - m1 * m2) + p.c2;
        const float B1 = m1 * m1 + m2 * m2 + p.c1, B2 = (e11 - m1 * m1) + (e22 - m2 * m2) + p.c2;
        const float inv = 1.f / (B1 * B2);
        const float S = A1 * A2 * inv;
        s += S;
        if (p.d1) {
            const size_t o = ((size_t)plane * Hv + h) * Wv + w;
            p.d1[o] = 2.f * m2 * (A2 - A1) * inv - 2.f * m1 * S * (B2 - B1) * inv;
            p.d2[o] = -S / B2;
            p.d3[o] = 2.f * A1 * inv;
        }
    }
    s = block_sum(s);
    if (threadIdx.x == 0) atomicAdd(&p.acc[plane], s);
}

// grad[n,c,h,w] (+)= scale * (F(d1) + 2 x F(d2) + y F(d3)), F = adjoint ("full") of the valid separable filter
struct SsimBwdParams {
    const float* x; const float* y; const float* d1; const float* d2; const float* d3; float* grad;
    const double* acc;      // ssim sums (for non_negative_ssim's relu mask)
    long sn, sc, sh, sw;
    int N, C, H, W, accumulate, non_negative;
    float scale;
    float g[SS_K];
};
__global__ __launch_bounds__(256) void ssim_backward_kernel(SsimBwdParams p) {
    __shared__ float sd[3][SS_TH + SS_R][SS_TW + SS_R + 1];
    __shared__ float hf[3][SS_TH + SS_R][SS_TW + 1];
    const int plane = blockIdx.z, n = plane / p.C, c = plane % p.C;
    const int Hv = p.H - SS_R, Wv = p.W - SS_R;
    const int ox = blockIdx.x * SS_TW, oy = blockIdx.y * SS_TH;       // tile origin in the image
    float scale = p.scale;
    if (p.non_negative && p.acc[plane] < 0.0) scale = 0.f;
    const size_t pb = (size_t)plane * Hv * Wv;
    for (int i = threadIdx.x; i < (SS_TH + SS_R) * (SS_TW + SS_R); i += 256) {
        const int r = i / (SS_TW + SS_R), q = i % (SS_TW + SS_R);
        const int h = oy + r - SS_R, w = ox + q - SS_R;               // map coordinates oy-10 .. oy+TH-1
        const bool ok = h >= 0 && h < Hv && w >= 0 && w < Wv;
        const size_t o = pb + (size_t)(ok ? h : 0) * Wv + (ok ? w : 0);
        sd[0][r][q] = ok ? p.d1[o] : 0.f; sd[1][r][q] = ok ? p.d2[o] : 0.f; sd[2][r][q] = ok ? p.d3[o] : 0.f;
    }
    __syncthreads();
    // out[w] = sum_k g[k] * d[w - k]  -> with the tile shifted by 10: d index q + (10 - k)
    for (int i = threadIdx.x; i < (SS_TH + SS_R) * SS_TW; i += 256) {
        const int r = i / SS_TW, q = i % SS_TW;
        float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float gk = p.g[k];
            a = fmaf(gk, sd[0][r][q + SS_R - k], a); b = fmaf(gk, sd[1][r][q + SS_R - k], b); cc = fmaf(gk, sd[2][r][q + SS_R - k], cc);
        }
        hf[0][r][q] = a; hf[1][r][q] = b; hf[2][r][q] = cc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SS_TH * SS_TW; i += 256) {
        const int r = i / SS_TW, q = i % SS_TW;
        const int h = oy + r, w = ox + q;
        if (h >= p.H || w >= p.W) continue;
        float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float gk = p.g[k];
            a = fmaf(gk, hf[0][r + SS_R - k][q], a); b = fmaf(gk, hf[1][r + SS_R - k][q], b); cc = fmaf(gk, hf[2][r + SS_R - k][q], cc);
        }
        const size_t o = n * p.sn + c * p.sc + h * p.sh + w * p.sw;
        const float gr = scale * (a + 2.f * p.x[o] * b + p.y[o] * cc);
        p.grad[o] = (p.accumulate ? p.grad[o] : 0.f) + gr;
    }
}

void gauss_window(float* g, float sigma) {       // _fspecial_gauss_1d (metric/image.py:90-108), fp32 like torch
    float s = 0.f;
    for (int k = 0; k < SS_K; ++k) { const float d = (float)(k - SS_K / 2); g[k] = expf(-(d * d) / (2.f * sigma * sigma)); s += g[k]; }
    for (int k = 0; k < SS_K; ++k) g[k] /= s;
}

// ---- optimiser ---------------------------------------------------------------------------------------------------------
// two deterministic stages (fixed block -> partial mapping, fixed-order final sum): every data-parallel replica gets the
// bit-identical norm from the bit-identical all-reduced gradients, so the replicas' weights cannot drift apart
__global__ void sumsq_partial_kernel(const float* __restrict__ a, double* __restrict__ partial, float scale, size_t total) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float v = a[i] * scale;
        s += (double)v * v;
    }
    s = block_sum(s);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void sumsq_final_kernel(const double* __restrict__ partial, int n, double* __restrict__ acc, int zero_first) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
    s = block_sum(s);
    if (threadIdx.x == 0) *acc = (zero_first ? 0.0 : *acc) + s;
}

// torch.optim.Adam (L2 weight decay, no amsgrad) preceded by clip_grad_norm_: g *= min(1, max_norm / (norm + 1e-6))
__global__ void adam_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                            const double* __restrict__ sumsq, float grad_scale, float max_norm, float lr, float b1, float b2, float eps,
                            float wd, float bc1, float bc2_sqrt, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float coef = 1.f;
    if (sumsq) {
        const double ss = *sumsq;
        if (!(ss < INFINITY)) return;            // overflowed 16-bit gradients (inf / NaN norm): skip the step, as AMP's GradScaler does
        if (max_norm > 0.f) coef = fminf(1.f, max_norm / ((float)sqrt(ss) + 1e-6f));
    }
    float g = grad[i] * grad_scale * coef;
    const float w = param[i];
    if (wd != 0.f) g = fmaf(wd, w, g);
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    param[i] = w - (lr / bc1) * (mi / denom);
}

// Step counter kept on the device (GradScaler semantics without a host sync): state = {int step, int applied, float bc1,
// float bc2_sqrt}.  A step whose gradient norm is not finite is not applied and does not advance the counter, so Adam's bias
// correction and the checkpointed global_step count the updates that really happened.
__global__ void adam_prepare_kernel(int* __restrict__ state, const double* __restrict__ sumsq, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool ok = !sumsq || (*sumsq < (double)INFINITY);
    int t = state[0];
    if (ok) { t += 1; state[0] = t; }
    state[1] = ok ? 1 : 0;
    ((float*)state)[2] = (float)(1.0 - pow((double)b1, (double)t));
    ((float*)state)[3] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
__global__ void adam_counted_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                                    const double* __restrict__ sumsq, const int* __restrict__ state, float grad_scale, float max_norm,
                                    float lr, float b1, float b2, float eps, float wd, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || state[1] == 0) return;
    const float bc1 = ((const float*)state)[2], bc2_sqrt = ((const float*)state)[3];
    float coef = 1.f;
    if (sumsq && max_norm > 0.f) coef = fminf(1.f, max_norm / ((float)sqrt(*sumsq) + 1e-6f));
    float g = grad[i] * grad_scale * coef;
    const float w = param[i];
    if (wd != 0.f) g = fmaf(wd, w, g);
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    param[i] = w - (lr / bc1) * (mi / denom);
}

}  // namespace

// ---- entry points ----------------------------------------------------------------------------------------------------
static int check_conv(const ie_convn_desc* d, const char* who) {
    if (!d) IE_FAIL(IE_ERR_INVALID, "%s: null descriptor", who);
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0 || d->nsrc < 1 ||
        d->nsrc > IE_CONV_MAX_SRC)
        IE_FAIL(IE_ERR_INVALID, "%s: bad shape", who);
    return IE_OK;
}

template <typename T>
static int wgrad_impl(ie_convn_desc* d, const void* const* srcs, const int* cs, const void* dy, float* dw_hwio, float* dbias, int accumulate,
                      double* ws, void* stream) {
    int rc = check_conv(d, "ie_convn_wgrad_nhwc");
    if (rc != IE_OK) return rc;
    if (!srcs || !cs || !dy || !dw_hwio || (dbias && !ws)) IE_FAIL(IE_ERR_INVALID, "ie_convn_wgrad_nhwc: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int Cin = 0;
    for (int s = 0; s < d->nsrc; ++s) { if (!srcs[s] || cs[s] <= 0) IE_FAIL(IE_ERR_INVALID, "ie_convn_wgrad_nhwc: source %d invalid", s); Cin += cs[s]; }
    d->Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    d->Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    if (d->Ho <= 0 || d->Wo <= 0) IE_FAIL(IE_ERR_INVALID, "ie_convn_wgrad_nhwc: empty output");
    const int taps = d->kh * d->kw;
    if (!accumulate) IE_HIP_CHECK(hipMemsetAsync(dw_hwio, 0, sizeof(float) * (size_t)taps * Cin * d->Cout, st));
    static const int force_fma = getenv("IE_WGRAD_FMA") ? atoi(getenv("IE_WGRAD_FMA")) : 0;      // A/B switch: skip the matrix-core kernel
    int cbase = 0;
    for (int s = 0; s < d->nsrc; ++s) {
        if constexpr (sizeof(T) == 2) {
            if (!force_fma && (ie_wgrad_mfma_applicable(d->kh, d->kw, d->stride, d->pad, d->pad_mode, cs[s], d->Cout) ||
                               ie_wgrad_mfma_applicable_blocks(d->kh, d->kw, d->stride, d->pad, d->pad_mode, cs[s], d->Cout))) {
                rc = ie_wgrad_mfma_launch(srcs[s], dy, ie_elem<T>::dtype, dw_hwio, d->N, d->Ho, d->Wo, cs[s], d->Cout, Cin, cbase, d->kh, d->pad, d->stride, d->H, d->W, st);
                if (rc != IE_OK) return rc;
                cbase += cs[s];
                continue;
            }
        }
        WgradParams p;
        p.x = srcs[s]; p.dy = dy; p.dw = dw_hwio;
        p.N = d->N; p.H = d->H; p.W = d->W; p.Cs = cs[s]; p.Cout = d->Cout; p.Cin_total = Cin; p.cbase = cbase; p.kh = d->kh; p.kw = d->kw;
        p.stride = d->stride; p.pad = d->pad; p.pad_mode = d->pad_mode; p.Ho = d->Ho; p.Wo = d->Wo;
        p.ci_tiles = ie_cdiv(cs[s], WG_T);
        const int co_tiles = ie_cdiv(d->Cout, WG_T);
        const int tiles = co_tiles * p.ci_tiles * taps;
        const int total_rows = d->N * d->Ho;
        int chunks = 4096 / tiles;
        if (chunks < 1) chunks = 1;
        if (chunks > total_rows) chunks = total_rows;
        p.rows_per_wg = ie_cdiv(total_rows, chunks);
        chunks = ie_cdiv(total_rows, p.rows_per_wg);
        p.slab = nullptr;
        if (ie_get_deterministic()) {
            void* ws_det = nullptr;
            rc = ie_det_workspace((size_t)chunks * tiles * WG_T * WG_T * sizeof(float), &ws_det);
            if (rc != IE_OK) return rc;
            p.slab = (float*)ws_det;
        }
        hipLaunchKernelGGL(conv_wgrad_kernel<T>, dim3(co_tiles, p.ci_tiles * taps, chunks), dim3(256), 0, st, p);
        IE_LAUNCH_CHECK();
        if (p.slab) {
            const size_t total = (size_t)tiles * WG_T * WG_T;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk(total)), dim3(256), 0, st, p.slab, dw_hwio, chunks, co_tiles, p.ci_tiles, taps, cs[s],
                               d->Cout, Cin, cbase);
            IE_LAUNCH_CHECK();
        }
        cbase += cs[s];
    }
    if (dbias) {
        IE_HIP_CHECK(hipMemsetAsync(ws, 0, sizeof(double) * d->Cout, st));
        const size_t npix = (size_t)d->N * d->Ho * d->Wo;
        const bool sum_v8 = sizeof(T) == 2 && (d->Cout & 7) == 0;
        for (int off = 0; sum_v8 && off < d->Cout; off += 2048) {
            const int cn = d->Cout - off < 2048 ? d->Cout - off : 2048;
            const int nch = cn / 8, tpc = 256 / nch;
            size_t blocks = (npix + (size_t)tpc * 16 - 1) / ((size_t)tpc * 16);
            if (blocks > 2048) blocks = 2048;
            const size_t min_blocks = (npix + (size_t)tpc * 512 - 1) / ((size_t)tpc * 512);
            if (blocks < min_blocks) blocks = min_blocks;
            if constexpr (sizeof(T) == 2)
                hipLaunchKernelGGL(channel_sum_v8_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)dy, ws, npix, d->Cout, off, nch);
            IE_LAUNCH_CHECK();
        }
        for (int off = 0; !sum_v8 && off < d->Cout; off += 256) {
            const int cn = d->Cout - off < 256 ? d->Cout - off : 256;
            size_t blocks = (npix + (size_t)(256 / cn) * 64 - 1) / ((size_t)(256 / cn) * 64);
            if (blocks > 1024) blocks = 1024;
            if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(channel_sum_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)dy, ws, npix, d->Cout, off, cn);
            IE_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(store_f64_as_f32_kernel, dim3(nblk(d->Cout)), dim3(256), 0, st, ws, dbias, d->Cout, accumulate);
        IE_LAUNCH_CHECK();
    }
    return IE_OK;
}

IE_EXPORT int ie_convn_wgrad_nhwc(ie_convn_desc* d, const void* const* srcs, const int* cs, const void* dy, int dtype, float* dw_hwio, float* dbias,
                                  int accumulate, double* ws, void* stream) {
    switch (dtype) {
    case IE_F32: return wgrad_impl<float>(d, srcs, cs, dy, dw_hwio, dbias, accumulate, ws, stream);
    case IE_F16: return wgrad_impl<_Float16>(d, srcs, cs, dy, dw_hwio, dbias, accumulate, ws, stream);
    case IE_BF16: return wgrad_impl<__bf16>(d, srcs, cs, dy, dw_hwio, dbias, accumulate, ws, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_convn_wgrad_nhwc: bad dtype");
    }
}

IE_EXPORT int ie_convn_wgrad_nhwc_f32(ie_convn_desc* d, const float* const* srcs, const int* cs, const float* dy, float* dw_hwio,
                                      float* dbias, int accumulate, double* ws, void* stream) {
    return wgrad_impl<float>(d, (const void* const*)srcs, cs, dy, dw_hwio, dbias, accumulate, ws, stream);
}

IE_EXPORT int ie_convn_dgrad_nhwc_f32(ie_convn_desc* d, const float* dy, const float* w_hwio, float* const* dsrcs, const int* cs,
                                      const int* accumulate, void* stream) {
    int rc = check_conv(d, "ie_convn_dgrad_nhwc_f32");
    if (rc != IE_OK) return rc;
    if (!dy || !w_hwio || !dsrcs || !cs) IE_FAIL(IE_ERR_INVALID, "ie_convn_dgrad_nhwc_f32: null pointer");
    if (d->pad_mode != IE_PAD_ZEROS) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_convn_dgrad_nhwc_f32: zero padding only");
    int Cin = 0;
    for (int s = 0; s < d->nsrc; ++s) { if (cs[s] <= 0) IE_FAIL(IE_ERR_INVALID, "ie_convn_dgrad_nhwc_f32: source %d invalid", s); Cin += cs[s]; }
    d->Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    d->Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    if (d->Ho <= 0 || d->Wo <= 0) IE_FAIL(IE_ERR_INVALID, "ie_convn_dgrad_nhwc_f32: empty output");
    hipStream_t st = (hipStream_t)stream;
    static const int force_gather = getenv("IE_DGRAD_GATHER") ? atoi(getenv("IE_DGRAD_GATHER")) : 0;     // A/B switch
    if (d->stride == 1 && d->kh - 1 - d->pad >= 0 && d->kw - 1 - d->pad >= 0 && d->kh - 1 - d->pad == d->kw - 1 - d->pad && !force_gather) {
        // stride 1: dX = conv(dY, W') with W' = W flipped and transposed -> the LDS-tiled forward kernel (3x the gather kernel)
        const int taps = d->kh * d->kw;
        const size_t wn = (size_t)taps * Cin * d->Cout;
        float* wt = nullptr;
        IE_HIP_CHECK(hipMallocAsync((void**)&wt, wn * sizeof(float), st));
        hipLaunchKernelGGL(flip_transpose_hwio_kernel, dim3(nblk(wn)), dim3(256), 0, st, w_hwio, wt, taps, Cin, d->Cout, wn);
        IE_LAUNCH_CHECK();
        int cb = 0;
        for (int s = 0; s < d->nsrc; ++s) {
            if (dsrcs[s]) {
                ConvNParams q;
                memset(&q, 0, sizeof(q));
                q.src[0] = dy; q.cs[0] = d->Cout; q.nsrc = 1; q.Cin = d->Cout;
                q.w = wt + cb; q.w_ld = Cin; q.Cout = cs[s]; q.dst = dsrcs[s];
                q.N = d->N; q.H = d->Ho; q.W = d->Wo; q.kh = d->kh; q.kw = d->kw; q.stride = 1; q.pad = d->kh - 1 - d->pad;
                q.pad_mode = IE_PAD_ZEROS; q.act = IE_ACT_NONE; q.Ho = d->H; q.Wo = d->W;
                q.accumulate = accumulate ? accumulate[s] : 0;
                rc = ie_convn_launch(q, st);
                if (rc != IE_OK) { hipFreeAsync(wt, st); return rc; }
            }
            cb += cs[s];
        }
        IE_HIP_CHECK(hipFreeAsync(wt, st));
        return IE_OK;
    }
    int cbase = 0;
    for (int s = 0; s < d->nsrc; ++s) {
        if (dsrcs[s]) {                    // NULL = this source needs no gradient (e.g. the input image)
            DgradParams p;
            p.dy = dy; p.w = w_hwio; p.dx = dsrcs[s];
            p.N = d->N; p.H = d->H; p.W = d->W; p.Cs = cs[s]; p.Cout = d->Cout; p.Cin_total = Cin; p.cbase = cbase; p.kh = d->kh;
            p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo; p.accumulate = accumulate ? accumulate[s] : 0;
            dim3 grid(ie_cdiv(d->H * d->W, 64), ie_cdiv(cs[s], 4 * DG_CT), d->N);
            hipLaunchKernelGGL(conv_dgrad_kernel, grid, dim3(256), 0, st, p);
            IE_LAUNCH_CHECK();
        }
        cbase += cs[s];
    }
    return IE_OK;
}

// ---- row forms of the two adjoints for 16-bit tensors (8 channels = 16 bytes per lane) -------------------------------------------------------------
// blockIdx.x = (frame, row pair | input row): the row's scalars are computed once per workgroup, the lanes run over the row's 16-byte pieces with 32-bit
// arithmetic.  Same operations in the same order as the flat kernels above (bit-identical results); what goes away is the 64-bit index split per 16 bytes,
// the four-fold re-read of every pooling window (one lane now owns the window's four elements) and the per-lane candidate arithmetic of the row direction.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_backward_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C, int Ho,
                                                                     int Wo, int accumulate, int c8_shift) {
    typedef __attribute__((ext_vector_type(8))) T TV;
    const unsigned HP = (unsigned)(H + 1) >> 1, WP = (unsigned)(W + 1) >> 1;
    const unsigned n = blockIdx.x / HP, hp = blockIdx.x - n * HP;
    const unsigned C8 = (unsigned)C >> 3, pieces = WP * C8;
    const int h0 = 2 * (int)hp;
    const bool two_rows = h0 + 1 < H, win_row = (int)hp < Ho;
    const T* xr = x + ((size_t)n * H + h0) * (size_t)W * C;
    T* dr = dx + ((size_t)n * H + h0) * (size_t)W * C;
    const T* dyr = dy + ((size_t)n * Ho + (win_row ? hp : 0)) * (size_t)Wo * C;
    const size_t rowe = (size_t)W * C;
    for (unsigned j = blockIdx.y * 256u + threadIdx.x; j < pieces; j += gridDim.y * 256u) {
        const unsigned wp = c8_shift >= 0 ? (j >> c8_shift) : (j / C8), c = (j - wp * C8) * 8u;
        const int w0 = 2 * (int)wp;
        const bool two_cols = w0 + 1 < W, win = win_row && (int)wp < Wo;
        const size_t o00 = (size_t)w0 * C + c;
        float g[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) g[k][e] = 0.f;
        if (win) {          // both rows and both columns exist
            TV v[4];
            v[0] = *(const TV*)(xr + o00); v[1] = *(const TV*)(xr + o00 + C); v[2] = *(const TV*)(xr + rowe + o00); v[3] = *(const TV*)(xr + rowe + o00 + C);
            const TV d = *(const TV*)(dyr + (size_t)wp * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) f[k] = (float)v[k][e];
                int best = 0;
#pragma unroll
                for (int k = 1; k < 4; ++k) if (f[k] > f[best] || (f[k] != f[k] && f[best] == f[best])) best = k;
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k][e] = best == k ? (float)d[e] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (((k & 2) && !two_rows) || ((k & 1) && !two_cols)) continue;
            T* o = dr + (k & 2 ? rowe : 0) + o00 + (k & 1 ? C : 0);
            TV out;
            if (accumulate) {
                const TV old = *(const TV*)o;
#pragma unroll
                for (int e = 0; e < 8; ++e) out[e] = (T)((float)old[e] + g[k][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) out[e] = (T)g[k][e];
            }
            *(TV*)o = out;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void resize_backward_rows_kernel(const T* __restrict__ dy, T* __restrict__ dx, int Hi, int Wi, int C, int Ho, int Wo, int mode,
                                                                   int accumulate, int c8_shift) {
    typedef __attribute__((ext_vector_type(8))) T TV;
    const unsigned n = blockIdx.x / (unsigned)Hi;
    const int yi = (int)(blockIdx.x - n * (unsigned)Hi);
    const unsigned C8 = (unsigned)C >> 3, pieces = (unsigned)Wi * C8;
    int ylo, yhi;
    resize_candidates(yi, Hi, Ho, mode, ylo, yhi);
    const T* p = dy + (size_t)n * Ho * Wo * C;
    T* dr = dx + ((size_t)n * Hi + yi) * (size_t)Wi * C;
    for (unsigned j = blockIdx.y * 256u + threadIdx.x; j < pieces; j += gridDim.y * 256u) {
        const unsigned xw = c8_shift >= 0 ? (j >> c8_shift) : (j / C8), c = (j - xw * C8) * 8u;
        const int xi = (int)xw;
        int xlo, xhi;
        resize_candidates(xi, Wi, Wo, mode, xlo, xhi);
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        for (int yo = ylo; yo <= yhi; ++yo) {          // wave-uniform
            int a0, a1; float ly;
            resize_src(yo, Hi, Ho, mode, a0, a1, ly);
            float wy = 0.f;
            if (mode == 0) wy = (a0 == yi) ? 1.f : 0.f;
            else { if (a0 == yi) wy += 1.f - ly; if (a1 == yi) wy += ly; }
            if (wy == 0.f) continue;
            float rowacc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) rowacc[e] = 0.f;
            const T* prow = p + (size_t)yo * Wo * C + c;
            for (int xo = xlo; xo <= xhi; ++xo) {
                int b0, b1; float lx;
                resize_src(xo, Wi, Wo, mode, b0, b1, lx);
                float wx = 0.f;
                if (mode == 0) wx = (b0 == xi) ? 1.f : 0.f;
                else { if (b0 == xi) wx += 1.f - lx; if (b1 == xi) wx += lx; }
                if (wx != 0.f) {
                    const TV d = *(const TV*)(prow + (size_t)xo * C);
#pragma unroll
                    for (int e = 0; e < 8; ++e) rowacc[e] = fmaf(wx, (float)d[e], rowacc[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = fmaf(wy, rowacc[e], g[e]);
        }
        T* o = dr + (size_t)xi * C + c;
        TV out;
        if (accumulate) {
            const TV old = *(const TV*)o;
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (T)((float)old[e] + g[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (T)g[e];
        }
        *(TV*)o = out;
    }
}

static inline int c8_shift_of(int C8) { int s = 0; while ((1 << s) < C8) ++s; return (1 << s) == C8 ? s : -1; }
static inline dim3 row_grid(size_t rows, size_t pieces) {
    size_t gy = (pieces + 1023) / 1024;
    return dim3((unsigned)rows, (unsigned)(gy < 1 ? 1 : gy > 64 ? 64 : gy));
}

template <typename T>
static int maxpool_bwd_impl(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int accumulate, void* stream) {
    if (!x || !dy || !dx || N <= 0 || H < 2 || W < 2 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_maxpool2x2_backward_nhwc: bad args");
    const size_t total = (size_t)N * H * W * C;
    static const int rows_form = getenv("IE_TRAIN_ROWS") ? atoi(getenv("IE_TRAIN_ROWS")) : 1;      // A/B: 0 = the flat 16-bit kernels
    if (sizeof(T) == 2 && (C & 7) == 0 && rows_form && (size_t)N * ((H + 1) / 2) <= 0x7fffffffull)
        hipLaunchKernelGGL((maxpool2_backward_rows_kernel<T>), row_grid((size_t)N * ((H + 1) / 2), (size_t)((W + 1) / 2) * (C / 8)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (const T*)dy, (T*)dx, H, W, C, H / 2, W / 2, accumulate, c8_shift_of(C / 8));
    else if (sizeof(T) == 2 && (C & 7) == 0)
        hipLaunchKernelGGL((maxpool2_backward_kernel<T, 8>), dim3(nblk(total / 8)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx,
                           H, W, C, H / 2, W / 2, accumulate, total / 8);
    else
        hipLaunchKernelGGL((maxpool2_backward_kernel<T, 1>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx,
                           H, W, C, H / 2, W / 2, accumulate, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_maxpool2x2_backward_nhwc(const void* x, const void* dy, void* dx, int dtype, int N, int H, int W, int C, int accumulate, void* stream) {
    switch (dtype) {
    case IE_F32: return maxpool_bwd_impl<float>(x, dy, dx, N, H, W, C, accumulate, stream);
    case IE_F16: return maxpool_bwd_impl<_Float16>(x, dy, dx, N, H, W, C, accumulate, stream);
    case IE_BF16: return maxpool_bwd_impl<__bf16>(x, dy, dx, N, H, W, C, accumulate, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_maxpool2x2_backward_nhwc: bad dtype");
    }
}
IE_EXPORT int ie_maxpool2x2_backward_nhwc_f32(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int accumulate,
                                              void* stream) {
    return maxpool_bwd_impl<float>(x, dy, dx, N, H, W, C, accumulate, stream);
}

template <typename T>
static int resize_bwd_impl(const void* dy, void* dx, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, int accumulate, void* stream) {
    if (!dy || !dx || N <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || mode < 0 || mode > 2)
        IE_FAIL(IE_ERR_INVALID, "ie_resize_backward_nhwc: bad args");
    const size_t total = (size_t)N * Hi * Wi * C;
    static const int rows_form = getenv("IE_TRAIN_ROWS") ? atoi(getenv("IE_TRAIN_ROWS")) : 1;      // A/B: 0 = the flat 16-bit kernels
    if (sizeof(T) == 2 && (C & 7) == 0 && rows_form && (size_t)N * Hi <= 0x7fffffffull)
        hipLaunchKernelGGL((resize_backward_rows_kernel<T>), row_grid((size_t)N * Hi, (size_t)Wi * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, Hi, Wi,
                           C, Ho, Wo, mode, accumulate, c8_shift_of(C / 8));
    else if (sizeof(T) == 2 && (C & 7) == 0)
        hipLaunchKernelGGL((resize_backward_kernel<T, 8>), dim3(nblk(total / 8)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, Hi, Wi, C, Ho,
                           Wo, mode, accumulate, total / 8);
    else
        hipLaunchKernelGGL((resize_backward_kernel<T, 1>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, Hi, Wi, C, Ho,
                           Wo, mode, accumulate, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_resize_backward_nhwc(const void* dy, void* dx, int dtype, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, int accumulate,
                                      void* stream) {
    switch (dtype) {
    case IE_F32: return resize_bwd_impl<float>(dy, dx, N, Hi, Wi, C, Ho, Wo, mode, accumulate, stream);
    case IE_F16: return resize_bwd_impl<_Float16>(dy, dx, N, Hi, Wi, C, Ho, Wo, mode, accumulate, stream);
    case IE_BF16: return resize_bwd_impl<__bf16>(dy, dx, N, Hi, Wi, C, Ho, Wo, mode, accumulate, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_resize_backward_nhwc: bad dtype");
    }
}
IE_EXPORT int ie_resize_backward_nhwc_f32(const float* dy, float* dx, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode,
                                          int accumulate, void* stream) {
    return resize_bwd_impl<float>(dy, dx, N, Hi, Wi, C, Ho, Wo, mode, accumulate, stream);
}

template <typename T>
static int crop_impl(const void* src, void* dst, int N, int Hs, int Ws, int C, int top, int left, int H, int W, int accumulate, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || top < 0 || left < 0 || H <= 0 || W <= 0 || top + H > Hs || left + W > Ws)
        IE_FAIL(IE_ERR_INVALID, "ie_crop_nhwc: bad args");
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(crop_nhwc_kernel<T>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)src, (T*)dst, Hs, Ws, C, top, left, H, W,
                       accumulate, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_crop_nhwc(const void* src, void* dst, int dtype, int N, int Hs, int Ws, int C, int top, int left, int H, int W, int accumulate,
                           void* stream) {
    switch (dtype) {
    case IE_F32: return crop_impl<float>(src, dst, N, Hs, Ws, C, top, left, H, W, accumulate, stream);
    case IE_F16: return crop_impl<_Float16>(src, dst, N, Hs, Ws, C, top, left, H, W, accumulate, stream);
    case IE_BF16: return crop_impl<__bf16>(src, dst, N, Hs, Ws, C, top, left, H, W, accumulate, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_crop_nhwc: bad dtype");
    }
}
IE_EXPORT int ie_crop_nhwc_f32(const float* src, float* dst, int N, int Hs, int Ws, int C, int top, int left, int H, int W,
                               int accumulate, void* stream) {
    return crop_impl<float>(src, dst, N, Hs, Ws, C, top, left, H, W, accumulate, stream);
}

template <typename T>
static int paste_impl(const void* src, void* dst, int N, int H, int W, int C, int Hd, int Wd, int top, int left, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || top < 0 || left < 0 || H <= 0 || W <= 0 || top + H > Hd || left + W > Wd)
        IE_FAIL(IE_ERR_INVALID, "ie_paste_nhwc: bad args");
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(paste_nhwc_kernel<T>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)src, (T*)dst, Hd, Wd, C, top, left, H, W, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_paste_nhwc(const void* src, void* dst, int dtype, int N, int H, int W, int C, int Hd, int Wd, int top, int left, void* stream) {
    switch (dtype) {
    case IE_F32: return paste_impl<float>(src, dst, N, H, W, C, Hd, Wd, top, left, stream);
    case IE_F16: return paste_impl<_Float16>(src, dst, N, H, W, C, Hd, Wd, top, left, stream);
    case IE_BF16: return paste_impl<__bf16>(src, dst, N, H, W, C, Hd, Wd, top, left, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_paste_nhwc: bad dtype");
    }
}

template <typename T>
static int unshuffle_impl(const void* dy, void* dx, int N, int H, int W, int Cout, int r, int accumulate, void* stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || r <= 0) IE_FAIL(IE_ERR_INVALID, "ie_pixel_unshuffle_nhwc: bad args");
    const size_t total = (size_t)N * H * W * Cout * r * r;
    hipLaunchKernelGGL(pixel_unshuffle_kernel<T>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, H, W, Cout, r, accumulate,
                       total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_pixel_unshuffle_nhwc(const void* dy, void* dx, int dtype, int N, int H, int W, int Cout, int r, int accumulate, void* stream) {
    switch (dtype) {
    case IE_F32: return unshuffle_impl<float>(dy, dx, N, H, W, Cout, r, accumulate, stream);
    case IE_F16: return unshuffle_impl<_Float16>(dy, dx, N, H, W, Cout, r, accumulate, stream);
    case IE_BF16: return unshuffle_impl<__bf16>(dy, dx, N, H, W, Cout, r, accumulate, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_pixel_unshuffle_nhwc: bad dtype");
    }
}
IE_EXPORT int ie_pixel_unshuffle_nhwc_f32(const float* dy, float* dx, int N, int H, int W, int Cout, int r, int accumulate, void* stream) {
    return unshuffle_impl<float>(dy, dx, N, H, W, Cout, r, accumulate, stream);
}

template <typename T>
static int act_bwd_impl(const void* dy, const void* ref, void* dx, size_t n, int act, float slope, float lo, float hi, int accumulate, void* stream) {
    if (!dy || !ref || !dx || n == 0 || act < 0 || act > 5) IE_FAIL(IE_ERR_INVALID, "ie_act_backward: bad args");
    if (sizeof(T) == 2 && (n & 7) == 0)
        hipLaunchKernelGGL((act_backward_kernel<T, 8>), dim3(nblk(n / 8)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)ref, (T*)dx, act,
                           slope, lo, hi, accumulate, n / 8);
    else
        hipLaunchKernelGGL((act_backward_kernel<T, 1>), dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)ref, (T*)dx, act,
                           slope, lo, hi, accumulate, n);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_act_backward(const void* dy, const void* ref, void* dx, int dtype, size_t n, int act, float slope, float lo, float hi,
                              int accumulate, void* stream) {
    switch (dtype) {
    case IE_F32: return act_bwd_impl<float>(dy, ref, dx, n, act, slope, lo, hi, accumulate, stream);
    case IE_F16: return act_bwd_impl<_Float16>(dy, ref, dx, n, act, slope, lo, hi, accumulate, stream);
    case IE_BF16: return act_bwd_impl<__bf16>(dy, ref, dx, n, act, slope, lo, hi, accumulate, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_act_backward: bad dtype");
    }
}
IE_EXPORT int ie_act_backward_f32(const float* dy, const float* ref, float* dx, size_t n, int act, float slope, float lo, float hi,
                                  int accumulate, void* stream) {
    return act_bwd_impl<float>(dy, ref, dx, n, act, slope, lo, hi, accumulate, stream);
}

IE_EXPORT int ie_l1_loss_f32(const float* pred, const float* target, size_t n, double* sum_dev, float* grad, float grad_scale,
                             int accumulate, void* stream) {
    if (!pred || !target || !sum_dev || n == 0) IE_FAIL(IE_ERR_INVALID, "ie_l1_loss_f32: bad args");
    hipStream_t st = (hipStream_t)stream;
    IE_HIP_CHECK(hipMemsetAsync(sum_dev, 0, sizeof(double), st));
    size_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(l1_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pred, target, sum_dev, grad, grad_scale, accumulate, n);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_ssim_forward_f32(const float* x, const float* y, int N, int C, int H, int W, int64_t sn, int64_t sc, int64_t sh,
                                  int64_t sw, float data_range, float k1, float k2, float sigma, double* sum_dev, float* d_ws,
                                  void* stream) {
    if (!x || !y || !sum_dev || N <= 0 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_ssim_forward_f32: bad args");
    if (H < SS_K || W < SS_K) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_ssim_forward_f32: H and W must be >= 11 (got %dx%d)", H, W);
    hipStream_t st = (hipStream_t)stream;
    SsimParams p;
    p.x = x; p.y = y; p.sn = sn; p.sc = sc; p.sh = sh; p.sw = sw; p.N = N; p.C = C; p.H = H; p.W = W;
    p.c1 = (k1 * data_range) * (k1 * data_range); p.c2 = (k2 * data_range) * (k2 * data_range);
    p.acc = sum_dev;
    const size_t plane = (size_t)(H - SS_R) * (W - SS_R) * N * C;
    p.d1 = d_ws; p.d2 = d_ws ? d_ws + plane : nullptr; p.d3 = d_ws ? d_ws + 2 * plane : nullptr;
    gauss_window(p.g, sigma);
    IE_HIP_CHECK(hipMemsetAsync(sum_dev, 0, sizeof(double) * N * C, st));
    dim3 grid(ie_cdiv(W - SS_R, SS_TW), ie_cdiv(H - SS_R, SS_TH), N * C);
    hipLaunchKernelGGL(ssim_forward_kernel, grid, dim3(256), 0, st, p);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_ssim_backward_f32(const float* x, const float* y, const float* d_ws, const double* sum_dev, float* grad, int N, int C,
                                   int H, int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float sigma, float scale,
                                   int non_negative, int accumulate, void* stream) {
    if (!x || !y || !d_ws || !grad || !sum_dev || N <= 0 || C <= 0 || H < SS_K || W < SS_K)
        IE_FAIL(IE_ERR_INVALID, "ie_ssim_backward_f32: bad args");
    SsimBwdParams p;
    const size_t plane = (size_t)(H - SS_R) * (W - SS_R) * N * C;
    p.x = x; p.y = y; p.d1 = d_ws; p.d2 = d_ws + plane; p.d3 = d_ws + 2 * plane; p.grad = grad; p.acc = sum_dev;
    p.sn = sn; p.sc = sc; p.sh = sh; p.sw = sw; p.N = N; p.C = C; p.H = H; p.W = W; p.accumulate = accumulate; p.non_negative = non_negative;
    p.scale = scale;
    gauss_window(p.g, sigma);
    dim3 grid(ie_cdiv(W, SS_TW), ie_cdiv(H, SS_TH), N * C);
    hipLaunchKernelGGL(ssim_backward_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_sumsq_f32(const float* a, size_t n, float scale, double* acc_dev, double* ws_dev, int zero_first, void* stream) {
    if (!a || !acc_dev || !ws_dev || n == 0) IE_FAIL(IE_ERR_INVALID, "ie_sumsq_f32: bad args");
    hipStream_t st = (hipStream_t)stream;
    size_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > IE_SUMSQ_WS_DOUBLES) blocks = IE_SUMSQ_WS_DOUBLES;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, ws_dev, scale, n);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, ws_dev, (int)blocks, acc_dev, zero_first);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int step, float lr,
                               float beta1, float beta2, float eps, float weight_decay, float grad_scale, const double* sumsq_dev,
                               float max_norm, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n == 0 || step < 1) IE_FAIL(IE_ERR_INVALID, "ie_adam_step_f32: bad args");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, sumsq_dev,
                       grad_scale, max_norm, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), n);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_adam_step_counted_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int* state_dev, float lr,
                                       float beta1, float beta2, float eps, float weight_decay, float grad_scale, const double* sumsq_dev,
                                       float max_norm, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state_dev || n == 0) IE_FAIL(IE_ERR_INVALID, "ie_adam_step_counted_f32: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, st, state_dev, sumsq_dev, beta1, beta2);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(adam_counted_kernel, dim3(nblk(n)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, sumsq_dev, state_dev,
                       grad_scale, max_norm, lr, beta1, beta2, eps, weight_decay, n);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
