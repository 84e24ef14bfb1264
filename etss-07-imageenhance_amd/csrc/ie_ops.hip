// fp32 NHWC building blocks of the U-Net / LLUnet++ / RetinexNet / Zero-DCE++ generators (parity path):
// multi-source conv (virtual concat of up to 8 tensors) with fused bias + activation + residual +
// clamp, 2x2 max-pool, resize (nearest / bilinear), pixel-shuffle, 2x2 stride-2
// transposed conv, zero pad, broadcast multiply, RetinexNet's max-channel concat.
// All streaming kernels are HBM-bound (one read + one write of the tensor); the conv runs on the VALU.
// Reference call sites are cited at each entry point in include/ie_hip.h.
#include "ie_common.h"
#include "ie_conv_internal.h"
#include <math.h>
#include <stdlib.h>

namespace {

__device__ __forceinline__ float act_f(float v, int act, float slope) { return ie_act_apply(v, act, slope); }

constexpr int MAXS = IE_CONV_MAX_SRC;
constexpr int CT = 8;   // output channels per thread


// lane = output pixel, wave = 8 output channels (weights wave-uniform -> scalar loads)
__global__ __launch_bounds__(256) void convn_direct_kernel(ConvNParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.z;
    const int opix = blockIdx.x * 64 + lane;
    const int co0 = (blockIdx.y * 4 + wave) * CT;
    if (co0 >= p.Cout) return;
    const bool live = opix < p.Ho * p.Wo;
    const int ho = live ? opix / p.Wo : 0, wo = live ? opix % p.Wo : 0;
    const int nco = min(CT, p.Cout - co0);
    float acc[CT];
#pragma unroll
    for (int o = 0; o < CT; ++o) acc[o] = (p.bias && o < nco) ? p.bias[co0 + o] : 0.f;
    for (int i = 0; i < p.kh; ++i)
        for (int j = 0; j < p.kw; ++j) {
            int hi = ho * p.stride + i - p.pad, wi = wo * p.stride + j - p.pad;
            bool inb = true;
            if (p.pad_mode == IE_PAD_REPLICATE) {
                hi = min(max(hi, 0), p.H - 1);
                wi = min(max(wi, 0), p.W - 1);
            } else {
                inb = hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
            }
            const size_t pix = ((size_t)n * p.H + (inb ? hi : 0)) * p.W + (inb ? wi : 0);
            const float* wtap = p.w + (size_t)(i * p.kw + j) * p.Cin * p.w_ld + co0;
            int cbase = 0;
            for (int s = 0; s < p.nsrc; ++s) {
                const int Cs = p.cs[s];
                const float* sp = p.src[s] + pix * Cs;
                const float* ws = wtap + (size_t)cbase * p.w_ld;
                if ((Cs & 3) == 0) {
                    for (int c = 0; c < Cs; c += 4) {
                        ie_f32x4 v = inb ? *(const ie_f32x4*)(sp + c) : ie_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float* wr = ws + (size_t)(c + k) * p.w_ld;
                            if (nco == CT) {
#pragma unroll
                                for (int o = 0; o < CT; ++o) acc[o] = fmaf(v[k], wr[o], acc[o]);
                            } else {
#pragma unroll
                                for (int o = 0; o < CT; ++o) if (o < nco) acc[o] = fmaf(v[k], wr[o], acc[o]);
                            }
                        }
                    }
                } else {
                    for (int c = 0; c < Cs; ++c) {
                        const float v = inb ? sp[c] : 0.f;
                        const float* wr = ws + (size_t)c * p.w_ld;
#pragma unroll
                        for (int o = 0; o < CT; ++o) if (o < nco) acc[o] = fmaf(v, wr[o], acc[o]);
                    }
                }
                cbase += Cs;
            }
        }
    if (!live) return;
    const size_t ob = (((size_t)n * p.Ho + ho) * p.Wo + wo) * p.Cout + co0;
#pragma unroll
    for (int o = 0; o < CT; ++o)
        if (o < nco) {
            float v = act_f(acc[o], p.act, p.slope);
            if (p.res) v += p.res[ob + o];
            if (p.clamp) v = fminf(fmaxf(v, p.lo), p.hi);
            p.dst[ob + o] = p.accumulate ? p.dst[ob + o] + v : v;
        }
}

// ---- LDS-tiled fp32 convolution (implicit GEMM on the VALU) -----------------------------------------------------------
// Workgroup = 64 consecutive output pixels of one frame x 64 output channels; thread = 4 pixels x 4 channels; K = (tap, 16
// input channels of one source) staged through LDS: X tile [64 px][16 ci], W tile [16 ci][64 co].  16 FMAs per pair of
// ds_read_b128 -- about 3x the direct kernel on the 32..1024-channel layers of the U-Nets (same arithmetic, fp32 FMA).
constexpr int TK = 16;
// PT x CT outputs per thread (pixels x couts): the workgroup tile is 16*PT pixels x 16*CT couts.  4x4 reads 2 ds_read_b128 per
// 16 FMAs; 8x8 reads 4 per 64, halving the LDS traffic per FLOP (the kernel is LDS-bound, not FMA-bound).
template <int PT, int CT>
__global__ __launch_bounds__(256) void convn_tiled_kernel(ConvNParams p) {
    constexpr int TP = 16 * PT, TC = 16 * CT, NLP = PT / 4, NLW = CT / 4;
    __shared__ float xs[TP][TK + 4];
    __shared__ float wsh[TK][TC + 4];
    const int t = threadIdx.x;
    const int n = blockIdx.z;
    const int pix0 = blockIdx.x * TP, co_t = blockIdx.y * TC;
    const int npix = p.Ho * p.Wo;
    const int lc = (t & 3) * 4;                         // X loader: first of 4 channels of the chunk; pixels (t >> 2) + 64 i
    const int wk = t >> 4, wc = (t & 15) * 4;           // W loader: chunk row (input channel); couts wc + 64 i
    int lho[NLP], lwo[NLP];
    bool llive[NLP];
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
        const int lpix = pix0 + (t >> 2) + 64 * i;
        llive[i] = lpix < npix;
        lho[i] = llive[i] ? lpix / p.Wo : 0;
        lwo[i] = llive[i] ? lpix % p.Wo : 0;
    }
    const int tpx = (t >> 4) * PT, tco = (t & 15) * 4;   // thread's pixels tpx..tpx+PT-1; couts tco + 64 j + {0..3}
    ie_f32x2 acc[PT][NLW][2];
#pragma unroll
    for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int j = 0; j < NLW; ++j) { acc[a][j][0] = ie_f32x2{0.f, 0.f}; acc[a][j][1] = ie_f32x2{0.f, 0.f}; }
    const bool vec_w = (p.w_ld & 3) == 0;
    // K steps = (tap, source, 16-channel chunk), flattened so that the global loads of step k+1 are issued before the FMAs of
    // step k (register double buffering): without it every step exposes a full global-load latency between two barriers.
    int ti = 0, tj = 0, ts = 0, tc0 = 0, cbase = 0;     // iterator state of the step being LOADED
    size_t ipix[NLP];
    bool inb[NLP];
    auto set_tap = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NLP; ++u) {
            int hi = lho[u] * p.stride + ti - p.pad, wi = lwo[u] * p.stride + tj - p.pad;
            inb[u] = llive[u];
            if (p.pad_mode == IE_PAD_REPLICATE) { hi = min(max(hi, 0), p.H - 1); wi = min(max(wi, 0), p.W - 1); }
            else inb[u] = inb[u] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
            ipix[u] = ((size_t)n * p.H + (inb[u] ? hi : 0)) * p.W + (inb[u] ? wi : 0);
        }
    };
    ie_f32x4 vx[NLP], vw[NLW];
    auto load_step = [&]() __attribute__((always_inline)) {
        const int Cs = p.cs[ts];
        const float* wtap = p.w + (size_t)(ti * p.kw + tj) * p.Cin * p.w_ld;
#pragma unroll
        for (int u = 0; u < NLP; ++u) {
            vx[u] = ie_f32x4{0.f, 0.f, 0.f, 0.f};
            if (inb[u]) {
                const float* q = p.src[ts] + ipix[u] * Cs + tc0 + lc;
                if ((Cs & 3) == 0 && tc0 + lc + 3 < Cs) vx[u] = *(const ie_f32x4*)q;
                else
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (tc0 + lc + k < Cs) vx[u][k] = q[k];
            }
        }
#pragma unroll
        for (int u = 0; u < NLW; ++u) {
            vw[u] = ie_f32x4{0.f, 0.f, 0.f, 0.f};
            const int co = co_t + wc + 64 * u;
            if (tc0 + wk < Cs) {
                const float* q = wtap + (size_t)(cbase + tc0 + wk) * p.w_ld + co;
                if (vec_w && co + 3 < p.Cout) vw[u] = *(const ie_f32x4*)q;
                else
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (co + k < p.Cout) vw[u][k] = q[k];
            }
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {   // returns false after the last step
        tc0 += TK;
        if (tc0 < p.cs[ts]) return true;
        tc0 = 0; cbase += p.cs[ts]; ++ts;
        if (ts < p.nsrc) return true;
        ts = 0; cbase = 0; ++tj;
        if (tj == p.kw) { tj = 0; ++ti; }
        if (ti == p.kh) return false;
        set_tap();
        return true;
    };
    set_tap();
    load_step();
    bool more = true;
    while (more) {
        __syncthreads();                                  // the previous step's tiles have been read
#pragma unroll
        for (int u = 0; u < NLP; ++u) *(ie_f32x4*)&xs[(t >> 2) + 64 * u][lc] = vx[u];
#pragma unroll
        for (int u = 0; u < NLW; ++u) *(ie_f32x4*)&wsh[wk][wc + 64 * u] = vw[u];
        __syncthreads();
        more = advance();
        if (more) load_step();                            // in flight during the FMAs below
#pragma unroll
        for (int kk = 0; kk < TK; kk += 4) {
            ie_f32x4 a[PT];
#pragma unroll
            for (int q = 0; q < PT; ++q) a[q] = *(const ie_f32x4*)&xs[tpx + q][kk];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int u = 0; u < NLW; ++u) {
                    const ie_f32x4 bq = *(const ie_f32x4*)&wsh[kk + r][tco + 64 * u];
                    const ie_f32x2 b01 = {bq[0], bq[1]}, b23 = {bq[2], bq[3]};
#pragma unroll
                    for (int q = 0; q < PT; ++q) {
                        const ie_f32x2 av = {a[q][r], a[q][r]};
                        acc[q][u][0] = __builtin_elementwise_fma(av, b01, acc[q][u][0]);
                        acc[q][u][1] = __builtin_elementwise_fma(av, b23, acc[q][u][1]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NLW; ++u) {
        const int co = co_t + tco + 64 * u;
        if (co >= p.Cout) continue;
        const bool full = ((p.Cout & 3) == 0) && co + 3 < p.Cout;
        float bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = (p.bias && co + c < p.Cout) ? p.bias[co + c] : 0.f;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const int pix = pix0 + tpx + q;
            if (pix >= npix) continue;
            const size_t ob = ((size_t)n * npix + pix) * p.Cout + co;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = act_f(acc[q][u][c >> 1][c & 1] + bv[c], p.act, p.slope);
                if (p.res && co + c < p.Cout) v[c] += p.res[ob + c];
                if (p.clamp) v[c] = fminf(fmaxf(v[c], p.lo), p.hi);
                if (p.accumulate && co + c < p.Cout) v[c] += p.dst[ob + c];
            }
            if (full) *(ie_f32x4*)(p.dst + ob) = ie_f32x4{v[0], v[1], v[2], v[3]};
            else
#pragma unroll
                for (int c = 0; c < 4; ++c) if (co + c < p.Cout) p.dst[ob + c] = v[c];
        }
    }
}

__global__ void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int Ho, int Wo,
                                size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t r = i / C;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const size_t n = r / Ho;
    const float* p = x + ((n * H + 2 * ho) * W + 2 * wo) * C + c;
    y[i] = fmaxf(fmaxf(p[0], p[C]), fmaxf(p[(size_t)W * C], p[(size_t)W * C + C]));
}

// mode 0: legacy nearest (src = floor(dst*in/out)); 1: bilinear align_corners=False; 2: bilinear align_corners=True
__global__ void resize_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int Hi, int Wi, int C, int Ho, int Wo,
                                   int mode, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t r = i / C;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const size_t n = r / Ho;
    const float* p = x + n * (size_t)Hi * Wi * C + c;
    if (mode == 0) {
        const int ys = min((int)floorf(yo * ((float)Hi / (float)Ho)), Hi - 1);
        const int xs = min((int)floorf(xo * ((float)Wi / (float)Wo)), Wi - 1);
        y[i] = p[((size_t)ys * Wi + xs) * C];
        return;
    }
    float sy, sx;
    if (mode == 2) {
        const float ry = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
        const float rx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
        sy = ry * yo; sx = rx * xo;
    } else {
        sy = fmaxf((float)Hi / (float)Ho * (yo + 0.5f) - 0.5f, 0.f);
        sx = fmaxf((float)Wi / (float)Wo * (xo + 0.5f) - 0.5f, 0.f);
    }
    const int y0 = min((int)sy, Hi - 1), x0 = min((int)sx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float v00 = p[((size_t)y0 * Wi + x0) * C], v01 = p[((size_t)y0 * Wi + x1) * C];
    const float v10 = p[((size_t)y1 * Wi + x0) * C], v11 = p[((size_t)y1 * Wi + x1) * C];
    y[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

// y[n, h*r+i, w*r+j, c] = x[n, h, w, c*r*r + i*r + j]
__global__ void pixel_shuffle_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Co, int r,
                                     size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Co);
    size_t q = i / Co;
    const int wo = (int)(q % (W * r)); q /= (size_t)W * r;
    const int ho = (int)(q % (H * r));
    const size_t n = q / ((size_t)H * r);
    const int h = ho / r, ii = ho % r, w = wo / r, jj = wo % r;
    y[i] = x[((n * H + h) * W + w) * (size_t)(Co * r * r) + c * r * r + ii * r + jj];
}

// ConvTranspose2d(k=2, s=2): y[n,2h+i,2w+j,o] = b[o] + sum_c x[n,h,w,c] W[c,o,i,j]; weights repacked [i][j][c][o]
__global__ void convT2x2_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                float* __restrict__ y, int H, int W, int Ci, int Co, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int o = (int)(i % Co);
    size_t q = i / Co;
    const int wo = (int)(q % (2 * W)); q /= (size_t)2 * W;
    const int ho = (int)(q % (2 * H));
    const size_t n = q / ((size_t)2 * H);
    const float* xp = x + ((n * H + ho / 2) * W + wo / 2) * Ci;
    const float* wp = w + ((size_t)((ho & 1) * 2 + (wo & 1)) * Ci) * Co + o;
    float acc = b ? b[o] : 0.f;
    for (int c = 0; c < Ci; ++c) acc = fmaf(xp[c], wp[(size_t)c * Co], acc);
    y[i] = acc;
}

__global__ void pad_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int top, int left,
                                int Ho, int Wo, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int wo = (int)(q % Wo); q /= Wo;
    const int ho = (int)(q % Ho);
    const size_t n = q / Ho;
    const int h = ho - top, w = wo - left;
    y[i] = (h >= 0 && h < H && w >= 0 && w < W) ? x[((n * H + h) * W + w) * C + c] : 0.f;
}

// op 0: a + b, 1: a * b, 2: clamp(a, lo, hi), 3: a[:, :C] * b (a has C+1 channels), 4: a - b, 5: clamp(a / b, lo, hi),
// 6: a * lo + hi, 7: pow(a, lo), 8: a / max_all where b points at ONE float (a device-side reduction result),
// 9: a / b[0] > lo ? 1 : 0 (b ONE device float; BoundaryAwarePrior's normalised threshold);
// b may have 1 channel (broadcast) or C channels
__global__ void eltwise_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int C, int Cb,
                               int op, float lo, float hi, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (op == 3) { y[i] = a[(i / C) * (C + 1) + (i % C)] * b[i / C]; return; }
    const float va = a[i];
    float vb = 0.f;
    if (b && op < 8) vb = (Cb == C) ? b[i] : b[i / C];
    float r;
    switch (op) {
    case 0: r = va + vb; break;
    case 1: r = va * vb; break;
    case 2: r = fminf(fmaxf(va, lo), hi); break;
    case 4: r = va - vb; break;
    case 5: r = fminf(fmaxf(va / vb, lo), hi); break;
    case 6: r = va * lo + hi; break;
    case 7: r = powf(va, lo); break;
    case 8: r = va / b[0]; break;
    default: r = (va / b[0]) > lo ? 1.f : 0.f; break;       // 9
    }
    y[i] = r;
}

// RetinexNet DecomNet input (mon_extra/.../retinexnet/model.py:45-46): [max_c(x), x] -> 4 channels
__global__ void max_concat_kernel(const float* __restrict__ x, float* __restrict__ y, int C, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float* p = x + i * C;
    float m = p[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, p[c]);
    float* q = y + i * (C + 1);
    q[0] = m;
    for (int c = 0; c < C; ++c) q[c + 1] = p[c];
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

IE_EXPORT int ie_convn_nhwc_f32(ie_convn_desc* d, const float* const* srcs, const int* cs, const float* w, const float* bias,
                                const float* residual, float* dst, void* stream) {
    if (!d || !srcs || !cs || !w || !dst) IE_FAIL(IE_ERR_INVALID, "ie_convn_nhwc_f32: null pointer");
    if (d->nsrc < 1 || d->nsrc > MAXS) IE_FAIL(IE_ERR_INVALID, "ie_convn_nhwc_f32: nsrc=%d not in 1..%d", d->nsrc, MAXS);
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0)
        IE_FAIL(IE_ERR_INVALID, "ie_convn_nhwc_f32: bad shape");
    ConvNParams p;
    p.Cin = 0;
    for (int s = 0; s < MAXS; ++s) { p.src[s] = nullptr; p.cs[s] = 0; }
    for (int s = 0; s < d->nsrc; ++s) {
        if (!srcs[s] || cs[s] <= 0) IE_FAIL(IE_ERR_INVALID, "ie_convn_nhwc_f32: source %d invalid", s);
        p.src[s] = srcs[s]; p.cs[s] = cs[s]; p.Cin += cs[s];
    }
    d->Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    d->Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    if (d->Ho <= 0 || d->Wo <= 0) IE_FAIL(IE_ERR_INVALID, "ie_convn_nhwc_f32: empty output");
    p.nsrc = d->nsrc; p.w = w; p.bias = bias; p.res = residual; p.dst = dst;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.pad_mode = d->pad_mode; p.act = d->act; p.Ho = d->Ho; p.Wo = d->Wo; p.slope = d->slope;
    p.clamp = d->clamp; p.lo = d->clamp_lo; p.hi = d->clamp_hi;
    p.w_ld = d->Cout; p.accumulate = 0;
    return ie_convn_launch(p, (hipStream_t)stream);
}

int ie_convn_launch(const ConvNParams& p, hipStream_t stream) {
    static const int force_direct = getenv("IE_CONV_DIRECT") ? atoi(getenv("IE_CONV_DIRECT")) : 0;     // A/B switch
    static const int tile_sel = getenv("IE_CONV_TILE") ? atoi(getenv("IE_CONV_TILE")) : 0;             // 44 / 84 / 88 force a tile
    if (!force_direct && p.Cin >= 8 && p.Cout >= 16) {
        const long npix = (long)p.Ho * p.Wo;
        const bool big_px = npix * p.N >= 128L * 1024;                 // enough pixels to fill the chip with 128-pixel tiles
        int sel = tile_sel ? tile_sel : (big_px ? (p.Cout > 64 ? 88 : 84) : 44);
        if (sel == 88) {
            hipLaunchKernelGGL((convn_tiled_kernel<8, 8>), dim3(ie_cdiv((int)npix, 128), ie_cdiv(p.Cout, 128), p.N), dim3(256), 0, stream, p);
        } else if (sel == 84) {
            hipLaunchKernelGGL((convn_tiled_kernel<8, 4>), dim3(ie_cdiv((int)npix, 128), ie_cdiv(p.Cout, 64), p.N), dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL((convn_tiled_kernel<4, 4>), dim3(ie_cdiv((int)npix, 64), ie_cdiv(p.Cout, 64), p.N), dim3(256), 0, stream, p);
        }
    } else {
        dim3 grid(ie_cdiv(p.Ho * p.Wo, 64), ie_cdiv(p.Cout, 4 * CT), p.N);
        hipLaunchKernelGGL(convn_direct_kernel, grid, dim3(256), 0, stream, p);
    }
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_maxpool2x2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (!x || !y || N <= 0 || H < 2 || W < 2 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_maxpool2x2_nhwc_f32: bad args");
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * C;
    hipLaunchKernelGGL(maxpool2_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, Ho, Wo, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_resize_nhwc_f32(const float* x, float* y, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, void* stream) {
    if (!x || !y || N <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || mode < 0 || mode > 2)
        IE_FAIL(IE_ERR_INVALID, "ie_resize_nhwc_f32: bad args");
    const size_t total = (size_t)N * Ho * Wo * C;
    hipLaunchKernelGGL(resize_nhwc_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, Hi, Wi, C, Ho, Wo, mode, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_pixel_shuffle_nhwc_f32(const float* x, float* y, int N, int H, int W, int Cout, int r, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || r <= 0) IE_FAIL(IE_ERR_INVALID, "ie_pixel_shuffle_nhwc_f32: bad args");
    const size_t total = (size_t)N * H * r * W * r * Cout;
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, Cout, r, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_conv_transpose2x2_nhwc_f32(const float* x, const float* w_ijco, const float* bias, float* y, int N, int H, int W,
                                            int Cin, int Cout, void* stream) {
    if (!x || !w_ijco || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) IE_FAIL(IE_ERR_INVALID, "ie_conv_transpose2x2_nhwc_f32: bad args");
    const size_t total = (size_t)N * 2 * H * 2 * W * Cout;
    hipLaunchKernelGGL(convT2x2_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, w_ijco, bias, y, H, W, Cin, Cout, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_pad_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int top, int bottom, int left, int right, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || top < 0 || bottom < 0 || left < 0 || right < 0)
        IE_FAIL(IE_ERR_INVALID, "ie_pad_nhwc_f32: bad args");
    const int Ho = H + top + bottom, Wo = W + left + right;
    const size_t total = (size_t)N * Ho * Wo * C;
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, top, left, Ho, Wo, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_eltwise_nhwc_f32(int op, const float* a, const float* b, float* y, size_t npix, int C, int Cb, float lo, float hi, void* stream) {
    const bool unary = op == 2 || op == 6 || op == 7;
    if (!a || !y || npix == 0 || C <= 0 || op < 0 || op > 9 || (!unary && (!b || (Cb != C && Cb != 1))) || ((op == 3 || op == 8 || op == 9) && Cb != 1))
        IE_FAIL(IE_ERR_INVALID, "ie_eltwise_nhwc_f32: bad args");
    if (unary) b = nullptr;
    const size_t total = npix * C;
    hipLaunchKernelGGL(eltwise_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, a, b, y, C, Cb, op, lo, hi, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_max_concat_nhwc_f32(const float* x, float* y, size_t npix, int C, void* stream) {
    if (!x || !y || npix == 0 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_max_concat_nhwc_f32: bad args");
    hipLaunchKernelGGL(max_concat_kernel, dim3(nblk(npix)), dim3(256), 0, (hipStream_t)stream, x, y, C, npix);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
