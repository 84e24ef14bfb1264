// 16-bit (fp16 / bf16) NHWC streaming ops that sit between the matrix-core convolutions of the generators: 2x2 max-pool,
// resize (nearest / bilinear), zero pad, pixel shuffle, element-wise add / multiply, instance norm, dtype / channel-count
// casts.  All are HBM-bound single passes; a thread moves 8 channels (16 bytes) so every access is a full dwordx4 and
// a wave covers 1 KB of contiguous NHWC memory.  Arithmetic is fp32 on the loaded values, one rounding on store.
// Semantics are those of the fp32 kernels in ie_ops.hip / ie_norm.hip (reference call sites cited in include/ie_hip.h).
#include "ie_common.h"
#include <math.h>

namespace {

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

template <typename T> __device__ __forceinline__ void unpack8(const ie_u32x4& r, float (&f)[8]) {
    typedef __attribute__((ext_vector_type(8))) T t8;
    const t8 v = __builtin_bit_cast(t8, r);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (float)v[k];
}
template <typename T> __device__ __forceinline__ ie_u32x4 pack8(const float (&f)[8]) {
    return ie_u32x4{ie_pack2<T>(f[0], f[1]), ie_pack2<T>(f[2], f[3]), ie_pack2<T>(f[4], f[5]), ie_pack2<T>(f[6], f[7])};
}
template <typename T> __device__ __forceinline__ ie_u32x4 ld8(const T* p) { return *(const ie_u32x4*)p; }
template <typename T> __device__ __forceinline__ void st8(T* p, const ie_u32x4& v) { *(ie_u32x4*)p = v; }

// index decomposition of thread i over (n, h, w, c8)
struct Idx { size_t n; int h, w, c; };
__device__ __forceinline__ Idx split(size_t i, int H, int W, int C8) {
    Idx r;
    r.c = (int)(i % C8) * 8; i /= C8;
    r.w = (int)(i % W); i /= W;
    r.h = (int)(i % H);
    r.n = i / H;
    return r;
}

// Row form of the three ops a U-Net runs between its convolutions (max-pool, resize, zero pad): blockIdx.x = output row (n, h), the row's scalars computed
// once per workgroup in SGPRs, lanes over the row's 16-byte pieces with 32-bit index arithmetic.  (The flat form split a 64-bit index with six 64-bit
// divisions per 16 bytes moved: ALU-bound at 2.2 TB/s.)
struct RowIdx { unsigned n, h; };
__device__ __forceinline__ RowIdx row_of_block(int Ho) {
    const unsigned r = blockIdx.x, n = r / (unsigned)Ho;
    return RowIdx{n, r - n * (unsigned)Ho};
}
__device__ __forceinline__ void piece_of(unsigned j, unsigned C8, int c8_shift, unsigned& w, unsigned& c) {
    w = c8_shift >= 0 ? (j >> c8_shift) : (j / C8);
    c = (j - w * C8) * 8u;
}
inline int c8_shift_of(int C8) { int s = 0; while ((1 << s) < C8) ++s; return (1 << s) == C8 ? s : -1; }
inline dim3 row_grid(size_t rows, size_t pieces) {          // y: four pieces per lane, at most 64 column blocks
    size_t gy = (pieces + 1023) / 1024;
    return dim3((unsigned)rows, (unsigned)(gy < 1 ? 1 : gy > 64 ? 64 : gy));
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_v8(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int Ho, int Wo, int c8_shift) {
    const RowIdx o = row_of_block(Ho);
    const unsigned C8 = (unsigned)C / 8u, pieces = (unsigned)Wo * C8;
    const T* xr = x + ((size_t)o.n * H + 2 * o.h) * (size_t)W * C;
    T* yr = y + ((size_t)o.n * Ho + o.h) * (size_t)Wo * C;
    for (unsigned j = blockIdx.y * 256u + threadIdx.x; j < pieces; j += gridDim.y * 256u) {
        unsigned w, c;
        piece_of(j, C8, c8_shift, w, c);
        const T* p = xr + (size_t)(2 * w) * C + c;
        float a[8], b[8], cc[8], d[8], r[8];
        unpack8<T>(ld8(p), a); unpack8<T>(ld8(p + C), b); unpack8<T>(ld8(p + (size_t)W * C), cc); unpack8<T>(ld8(p + (size_t)W * C + C), d);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(cc[k], d[k]));
        st8(yr + (size_t)w * C + c, pack8<T>(r));
    }
}

// (top, left, Hp, Wp): the resized image sits at (top, left) of a zero (Hp, Wp) output -- F.pad of the up-sampled tensor in the same pass (unet.py:97-103)
template <typename T>
__global__ __launch_bounds__(256) void resize_v8(const T* __restrict__ x, T* __restrict__ y, int Hi, int Wi, int C, int Ho, int Wo, int mode, int c8_shift,
                                                 int top, int left, int Hp, int Wp) {
    const RowIdx o = row_of_block(Hp);
    const unsigned C8 = (unsigned)C / 8u, pieces = (unsigned)Wp * C8;
    const T* p = x + (size_t)o.n * Hi * Wi * C;
    T* yr = y + ((size_t)o.n * Hp + o.h) * (size_t)Wp * C;
    const int h = (int)o.h - top;
    if (h < 0 || h >= Ho) {
        for (unsigned j = blockIdx.y * 256u + threadIdx.x; j < pieces; j += gridDim.y * 256u) st8(yr + (size_t)j * 8, ie_u32x4{0u, 0u, 0u, 0u});
        return;
    }
    if (mode == 0) {
        const int ys = min((int)floorf(h * ((float)Hi / (float)Ho)), Hi - 1);
        for (unsigned j = blockIdx.y * 256u + threadIdx.x; j < pieces; j += gridDim.y * 256u) {
            unsigned wp, c;
            piece_of(j, C8, c8_shift, wp, c);
            const int w = (int)wp - left;
            const int xs = min((int)floorf(w * ((float)Wi / (float)Wo)), Wi - 1);
            st8(yr + (size_t)wp * C + c, (w >= 0 && w < Wo) ? ld8(p + ((size_t)ys * Wi + xs) * C + c) : ie_u32x4{0u, 0u, 0u, 0u});
        }
        return;
    }
    const float sy = mode == 2 ? (Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f) * h : fmaxf((float)Hi / (float)Ho * (h + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)sy, Hi - 1), y1 = min(y0 + 1, Hi - 1);
    const float ly = sy - (float)y0, hy = 1.f - ly;
    const T* r0 = p + (size_t)y0 * Wi * C;
    const T* r1 = p + (size_t)y1 * Wi * C;
    for (unsigned j = blockIdx.y * 256u + threadIdx.x; j < pieces; j += gridDim.y * 256u) {
        unsigned wp, c;
        piece_of(j, C8, c8_shift, wp, c);
        const int w = (int)wp - left;
        if (w < 0 || w >= Wo) { st8(yr + (size_t)wp * C + c, ie_u32x4{0u, 0u, 0u, 0u}); continue; }
        const float sx = mode == 2 ? (Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f) * w : fmaxf((float)Wi / (float)Wo * (w + 0.5f) - 0.5f, 0.f);
        const int x0 = min((int)sx, Wi - 1), x1 = min(x0 + 1, Wi - 1);
        const float lx = sx - (float)x0, hx = 1.f - lx;
        float a[8], b[8], cc[8], d[8], r[8];
        unpack8<T>(ld8(r0 + (size_t)x0 * C + c), a); unpack8<T>(ld8(r0 + (size_t)x1 * C + c), b);
        unpack8<T>(ld8(r1 + (size_t)x0 * C + c), cc); unpack8<T>(ld8(r1 + (size_t)x1 * C + c), d);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = hy * (hx * a[k] + lx * b[k]) + ly * (hx * cc[k] + lx * d[k]);
        st8(yr + (size_t)wp * C + c, pack8<T>(r));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pad_v8(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int top, int left, int Ho, int Wo, int c8_shift) {
    const RowIdx o = row_of_block(Ho);
    const unsigned C8 = (unsigned)C / 8u, pieces = (unsigned)Wo * C8;
    const int h = (int)o.h - top;
    const bool row_in = h >= 0 && h < H;
    const T* xr = x + ((size_t)o.n * H + (row_in ? h : 0)) * (size_t)W * C;
    T* yr = y + ((size_t)o.n * Ho + o.h) * (size_t)Wo * C;
    for (unsigned j = blockIdx.y * 256u + threadIdx.x; j < pieces; j += gridDim.y * 256u) {
        unsigned wo, c;
        piece_of(j, C8, c8_shift, wo, c);
        const int w = (int)wo - left;
        ie_u32x4 v = {0u, 0u, 0u, 0u};
        if (row_in && w >= 0 && w < W) v = ld8(xr + (size_t)w * C + c);
        st8(yr + (size_t)wo * C + c, v);
    }
}

// F.pad(x, (pad,) * 4, mode="replicate") on a 16-bit tensor, its adjoint (every border pixel of x collects the padding pixels that copied it: fp32 sum, one
// rounding), and the zero insertion in front of a stride-2 convolution's data gradient (dst[n, 2h, 2w] = src[n, h, w], zeros elsewhere)
template <typename T>
__global__ void pad_replicate_v8(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int pad, int Ho, int Wo, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const Idx o = split(i, Ho, Wo, C / 8);
    const int h = min(max(o.h - pad, 0), H - 1), w = min(max(o.w - pad, 0), W - 1);
    st8(y + ((o.n * Ho + o.h) * Wo + o.w) * C + o.c, ld8(x + ((o.n * H + h) * W + w) * C + o.c));
}
template <typename T>
__global__ void pad_replicate_bwd_v8(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C, int pad, int accumulate, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const Idx o = split(i, H, W, C / 8);
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int h0 = o.h == 0 ? 0 : o.h + pad, h1 = o.h == H - 1 ? Hp - 1 : o.h + pad;
    const int w0 = o.w == 0 ? 0 : o.w + pad, w1 = o.w == W - 1 ? Wp - 1 : o.w + pad;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    T* q = dx + ((o.n * H + o.h) * W + o.w) * C + o.c;
    if (accumulate) unpack8<T>(ld8(q), acc);
    for (int a = h0; a <= h1; ++a)
        for (int b = w0; b <= w1; ++b) {
            float v[8];
            unpack8<T>(ld8(dy + ((o.n * Hp + a) * Wp + b) * C + o.c), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
    st8(q, pack8<T>(acc));
}
template <typename T>
__global__ void zero_insert2_v8(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int Ho, int Wo, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const Idx o = split(i, Ho, Wo, C / 8);
    ie_u32x4 v = {0u, 0u, 0u, 0u};
    if (!(o.h & 1) && !(o.w & 1) && (o.h >> 1) < H && (o.w >> 1) < W) v = ld8(x + ((o.n * H + (o.h >> 1)) * W + (o.w >> 1)) * C + o.c);
    st8(y + ((o.n * Ho + o.h) * Wo + o.w) * C + o.c, v);
}

// y[n, h*r+i, w*r+j, c] = x[n, h, w, c*r*r + i*r + j]; thread = 8 output channels of one output pixel
template <typename T>
__global__ void pixel_shuffle_v8(const T* __restrict__ x, T* __restrict__ y, int H, int W, int Co, int r, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const Idx o = split(i, H * r, W * r, Co / 8);
    const int h = o.h / r, ii = o.h % r, w = o.w / r, jj = o.w % r;
    const T* p = x + ((o.n * H + h) * W + w) * (size_t)(Co * r * r) + ii * r + jj;
    typedef __attribute__((ext_vector_type(8))) T t8;
    t8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(o.c + k) * r * r];
    *(t8*)(y + ((o.n * H * r + o.h) * (size_t)(W * r) + o.w) * Co + o.c) = v;
}

// op 0: a + b; 1: a * b (b with C channels or 1 channel)
template <typename T>
__global__ void eltwise_v8(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int C, int Cb, int op, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t e = i * 8;
    float fa[8], fb[8], r[8];
    unpack8<T>(ld8(a + e), fa);
    if (Cb == C) unpack8<T>(ld8(b + e), fb);
    else {
        const float s = (float)b[e / C];
#pragma unroll
        for (int k = 0; k < 8; ++k) fb[k] = s;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = op == 0 ? fa[k] + fb[k] : fa[k] * fb[k];
    st8(y + e, pack8<T>(r));
}

// generic dtype / channel-count cast: dst[p][c] = c < Cs ? src[p][c] : 0 for c < Cd
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int Cs, int Cd, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cd);
    const size_t px = i / Cd;
    dst[i] = c < Cs ? (D)(float)src[px * Cs + c] : (D)0.f;
}

// the same into a 16-bit tensor whose channel count is a multiple of 8 (the RGB image carried as 8 / 16 channels): a thread writes one 16-byte group of 8 channels
// (the element-wise form above: one 2-byte store and a 64-bit division per element -- 549 us for the 8 x 1080p input of a U-Net-64 training step)
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_pad8_kernel(const S* __restrict__ src, D* __restrict__ dst, int Cs, int g_shift, size_t groups) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups) return;
    const size_t px = i >> g_shift;
    const int c0 = (int)(i - (px << g_shift)) * 8;
    const S* sp = src + px * Cs + c0;
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = c0 + k < Cs ? (float)sp[k] : 0.f;
    st8(dst + i * 8, pack8<D>(f));
}

// torch.cat(srcs, dim=1) of up to four fp32 NHWC tensors, converted to D and zero-padded to Cd channels in one pass (a thread = one output element)
struct CatSrcs { const float* p[4]; int c[4]; int n; };
template <typename D>
__global__ void concat_cast_kernel(CatSrcs s, D* __restrict__ dst, int Cd, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int c = (int)(i % Cd);
    const size_t px = i / Cd;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < s.n) {
            if (c >= 0 && c < s.c[k]) v = s.p[k][px * s.c[k] + c];
            c -= s.c[k];
        }
    }
    dst[i] = (D)v;
}

// pure dtype conversion (same channel count): 8 elements per thread, vector loads and stores, no index arithmetic
template <typename S, typename D>
__global__ void cast8_kernel(const S* __restrict__ src, D* __restrict__ dst, size_t total8) {
    typedef __attribute__((ext_vector_type(8))) S SV;
    typedef __attribute__((ext_vector_type(8))) D DV;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const SV v = *(const SV*)(src + i * 8);
    DV o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (D)(float)v[k];
    *(DV*)(dst + i * 8) = o;
}

// instance-norm statistics of a 16-bit tensor: thread = (pixel slot, 8 channels); fp64 partials + atomics
template <typename T>
__global__ void norm16_stats(const T* __restrict__ x, double* __restrict__ stats, int HW, int C, int c0, int cn) {
    const int n = blockIdx.y;
    const int g8 = cn / 8;                          // channel groups
    const int tpc = max(1, 256 / g8);
    const int cg = threadIdx.x % g8, ps = threadIdx.x / g8;
    double s[8], ss[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.0; ss[k] = 0.0; }
    if (ps < tpc) {
        const T* xp = x + (size_t)n * HW * C + c0 + cg * 8;
        for (int pix = blockIdx.x * tpc + ps; pix < HW; pix += gridDim.x * tpc) {
            float v[8];
            unpack8<T>(ld8(xp + (size_t)pix * C), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += v[k]; ss[k] += (double)v[k] * v[k]; }
        }
    }
    __shared__ double sh[16][256];                  // block partials first: one atomic per (block, channel, statistic)
#pragma unroll
    for (int k = 0; k < 8; ++k) { sh[k][threadIdx.x] = s[k]; sh[8 + k][threadIdx.x] = ss[k]; }
    __syncthreads();
    if (ps == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double a = s[k], b = ss[k];
            for (int j = 1; j < tpc; ++j) { a += sh[k][j * g8 + cg]; b += sh[8 + k][j * g8 + cg]; }
            atomicAdd(&stats[((size_t)n * cn + cg * 8 + k) * 2 + 0], a);
            atomicAdd(&stats[((size_t)n * cn + cg * 8 + k) * 2 + 1], b);
        }
    }
}
__global__ void norm16_finalize(const double* __restrict__ stats, float* __restrict__ mean, float* __restrict__ inv, int total, double count,
                                float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double m = stats[2 * (size_t)i] / count;
    const double var = fmax(stats[2 * (size_t)i + 1] / count - m * m, 0.0);
    mean[i] = (float)m;
    inv[i] = (float)(1.0 / sqrt(var + (double)eps));
}
// apply pass: thread = (pixel slot, 8-channel chunk) walking down the frame with its per-channel coefficients in registers -- one
// 16-byte load, a few FMAs and one 16-byte store per step (the one-thread-per-chunk form spent its time on 64-bit index divisions and
// 32 scalar coefficient loads per 16 bytes: 4.7x the statistics pass instead of 2x).  Grid (pixel chunks, N, channel blocks).
template <typename T>
__global__ __launch_bounds__(256) void norm16_apply(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ inv,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, int HW,
                                                    int C, int c0, int cn, int act, float slope, int nchb) {
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
            const size_t idx = (size_t)n * cn + (c - c0) + k;
            m[k] = mean[idx]; iv[k] = inv[idx];
            if (gamma) { ga[k] = gamma[c - c0 + k]; be[k] = beta[c - c0 + k]; }
        }
    }
    const size_t base = (size_t)n * HW * C + c;
    for (int pix = blockIdx.x * tpc + ps; pix < HW; pix += gridDim.x * tpc) {
        const size_t e = base + (size_t)pix * C;
        float v[8];
        unpack8<T>(ld8(x + e), v);
        if (in) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t = (v[k] - m[k]) * iv[k];
                if (gamma) t = t * ga[k] + be[k];
                v[k] = t;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ie_act_apply(v[k], act, slope);
        st8(y + e, pack8<T>(v));
    }
}


// depthwise 3x3 (groups = C, zero padding, stride 1) + bias + activation: thread = 8 channels of one output pixel,
// weights [9][C] fp32 staged in LDS once per block (nn/modules/conv/dsconv.py:122-134)
template <typename T>
__global__ void dwconv3x3_v8(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, int H,
                             int W, int C, int act, float slope, size_t total) {
    extern __shared__ float wl[];                  // [9][C] + [C]
    for (int i = threadIdx.x; i < 10 * C; i += blockDim.x) wl[i] = i < 9 * C ? w[i] : (bias ? bias[i - 9 * C] : 0.f);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const Idx o = split(i, H, W, C / 8);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = wl[9 * C + o.c + k];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = o.h + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = o.w + dx;
            if (xx < 0 || xx >= W) continue;
            float v[8];
            unpack8<T>(ld8(x + ((o.n * H + yy) * W + xx) * C + o.c), v);
            const float* wt = wl + ((dy + 1) * 3 + (dx + 1)) * C + o.c;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = fmaf(v[k], wt[k], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = ie_act_apply(acc[k], act, slope);
    st8(y + i * 8, pack8<T>(acc));
}

bool is16(int dt) { return dt == IE_F16 || dt == IE_BF16; }

}  // namespace

IE_EXPORT int ie_maxpool2x2_nhwc(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream) {
    if (dtype == IE_F32) return ie_maxpool2x2_nhwc_f32((const float*)x, (float*)y, N, H, W, C, stream);
    if (!x || !y || !is16(dtype) || N <= 0 || H < 2 || W < 2 || C <= 0 || (C & 7)) IE_FAIL(IE_ERR_INVALID, "ie_maxpool2x2_nhwc: bad args (C %% 8 == 0 on the 16-bit path)");
    const int Ho = H / 2, Wo = W / 2;
    if ((size_t)N * Ho > 0x7fffffffull) IE_FAIL(IE_ERR_INVALID, "ie_maxpool2x2_nhwc: too many rows");
    if (dtype == IE_F16) hipLaunchKernelGGL(maxpool2_v8<_Float16>, row_grid((size_t)N * Ho, (size_t)Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, H, W, C, Ho, Wo, c8_shift_of(C / 8));
    else hipLaunchKernelGGL(maxpool2_v8<__bf16>, row_grid((size_t)N * Ho, (size_t)Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, H, W, C, Ho, Wo, c8_shift_of(C / 8));
    IE_LAUNCH_CHECK();
    return IE_OK;
}

static int resize_pad16(const void* x, void* y, int dtype, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, int top, int bottom, int left, int right, void* stream,
                        const char* who) {
    if (!x || !y || !is16(dtype) || N <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || (C & 7) || Ho <= 0 || Wo <= 0 || mode < 0 || mode > 2 || top < 0 || bottom < 0 || left < 0 ||
        right < 0)
        IE_FAIL(IE_ERR_INVALID, "%s: bad args (16-bit tensors, C %% 8 == 0, padding >= 0)", who);
    const int Hp = Ho + top + bottom, Wp = Wo + left + right;
    if ((size_t)N * Hp > 0x7fffffffull) IE_FAIL(IE_ERR_INVALID, "%s: too many rows", who);
    if (dtype == IE_F16) hipLaunchKernelGGL(resize_v8<_Float16>, row_grid((size_t)N * Hp, (size_t)Wp * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, Hi, Wi, C, Ho, Wo, mode, c8_shift_of(C / 8), top, left, Hp, Wp);
    else hipLaunchKernelGGL(resize_v8<__bf16>, row_grid((size_t)N * Hp, (size_t)Wp * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, Hi, Wi, C, Ho, Wo, mode, c8_shift_of(C / 8), top, left, Hp, Wp);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_resize_nhwc(const void* x, void* y, int dtype, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, void* stream) {
    if (dtype == IE_F32) return ie_resize_nhwc_f32((const float*)x, (float*)y, N, Hi, Wi, C, Ho, Wo, mode, stream);
    return resize_pad16(x, y, dtype, N, Hi, Wi, C, Ho, Wo, mode, 0, 0, 0, 0, stream, "ie_resize_nhwc");
}

// F.pad(resize(x, (Ho, Wo)), [left, right, top, bottom]) in one pass: y is (N, Ho + top + bottom, Wo + left + right, C), zeros in the border
IE_EXPORT int ie_resize_pad_nhwc(const void* x, void* y, int dtype, int N, int Hi, int Wi, int C, int Ho, int Wo, int mode, int top, int bottom, int left, int right,
                                 void* stream) {
    return resize_pad16(x, y, dtype, N, Hi, Wi, C, Ho, Wo, mode, top, bottom, left, right, stream, "ie_resize_pad_nhwc");
}

IE_EXPORT int ie_pad_nhwc(const void* x, void* y, int dtype, int N, int H, int W, int C, int top, int bottom, int left, int right, void* stream) {
    if (dtype == IE_F32) return ie_pad_nhwc_f32((const float*)x, (float*)y, N, H, W, C, top, bottom, left, right, stream);
    if (!x || !y || !is16(dtype) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || top < 0 || bottom < 0 || left < 0 || right < 0)
        IE_FAIL(IE_ERR_INVALID, "ie_pad_nhwc: bad args (C %% 8 == 0 on the 16-bit path)");
    const int Ho = H + top + bottom, Wo = W + left + right;
    if ((size_t)N * Ho > 0x7fffffffull) IE_FAIL(IE_ERR_INVALID, "ie_pad_nhwc: too many rows");
    if (dtype == IE_F16) hipLaunchKernelGGL(pad_v8<_Float16>, row_grid((size_t)N * Ho, (size_t)Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, H, W, C, top, left, Ho, Wo, c8_shift_of(C / 8));
    else hipLaunchKernelGGL(pad_v8<__bf16>, row_grid((size_t)N * Ho, (size_t)Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, H, W, C, top, left, Ho, Wo, c8_shift_of(C / 8));
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_pad_replicate_nhwc(const void* x, void* y, int dtype, int N, int H, int W, int C, int pad, void* stream) {
    if (dtype == IE_F32) return ie_pad_replicate_nhwc_f32((const float*)x, (float*)y, N, H, W, C, pad, stream);
    if (!x || !y || !is16(dtype) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || pad < 0) IE_FAIL(IE_ERR_INVALID, "ie_pad_replicate_nhwc: bad args (C %% 8 == 0 on the 16-bit path)");
    const int Ho = H + 2 * pad, Wo = W + 2 * pad;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    if (dtype == IE_F16) hipLaunchKernelGGL(pad_replicate_v8<_Float16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, H, W, C, pad, Ho, Wo, total);
    else hipLaunchKernelGGL(pad_replicate_v8<__bf16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, H, W, C, pad, Ho, Wo, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_pad_replicate_backward_nhwc(const void* dy, void* dx, int dtype, int N, int H, int W, int C, int pad, int accumulate, void* stream) {
    if (dtype == IE_F32) return ie_pad_replicate_backward_nhwc_f32((const float*)dy, (float*)dx, N, H, W, C, pad, accumulate, stream);
    if (!dy || !dx || !is16(dtype) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || pad < 0)
        IE_FAIL(IE_ERR_INVALID, "ie_pad_replicate_backward_nhwc: bad args (C %% 8 == 0 on the 16-bit path)");
    const size_t total = (size_t)N * H * W * (C / 8);
    if (dtype == IE_F16) hipLaunchKernelGGL(pad_replicate_bwd_v8<_Float16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dy, (_Float16*)dx, H, W, C, pad, accumulate, total);
    else hipLaunchKernelGGL(pad_replicate_bwd_v8<__bf16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, (__bf16*)dx, H, W, C, pad, accumulate, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_zero_insert2_nhwc(const void* x, void* y, int dtype, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
    if (!x || !y || !is16(dtype) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || Ho < 2 * H - 1 || Wo < 2 * W - 1)
        IE_FAIL(IE_ERR_INVALID, "ie_zero_insert2_nhwc: 16-bit tensors, C %% 8 == 0, Ho >= 2 H - 1, Wo >= 2 W - 1");
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    if (dtype == IE_F16) hipLaunchKernelGGL(zero_insert2_v8<_Float16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, H, W, C, Ho, Wo, total);
    else hipLaunchKernelGGL(zero_insert2_v8<__bf16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, H, W, C, Ho, Wo, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_pixel_shuffle_nhwc(const void* x, void* y, int dtype, int N, int H, int W, int Cout, int r, void* stream) {
    if (dtype == IE_F32) return ie_pixel_shuffle_nhwc_f32((const float*)x, (float*)y, N, H, W, Cout, r, stream);
    if (!x || !y || !is16(dtype) || N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout & 7) || r <= 0)
        IE_FAIL(IE_ERR_INVALID, "ie_pixel_shuffle_nhwc: bad args (Cout %% 8 == 0 on the 16-bit path)");
    const size_t total = (size_t)N * H * r * W * r * (Cout / 8);
    if (dtype == IE_F16) hipLaunchKernelGGL(pixel_shuffle_v8<_Float16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, H, W, Cout, r, total);
    else hipLaunchKernelGGL(pixel_shuffle_v8<__bf16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, H, W, Cout, r, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_eltwise_nhwc(int op, const void* a, const void* b, void* y, int dtype, size_t npix, int C, int Cb, void* stream) {
    if (!a || !b || !y || !is16(dtype) || npix == 0 || C <= 0 || (C & 7) || (op != 0 && op != 1) || (Cb != C && Cb != 1))
        IE_FAIL(IE_ERR_INVALID, "ie_eltwise_nhwc: bad args (16-bit add / multiply, C %% 8 == 0)");
    const size_t total = npix * (C / 8);
    if (dtype == IE_F16) hipLaunchKernelGGL(eltwise_v8<_Float16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)a, (const _Float16*)b, (_Float16*)y, C, Cb, op, total);
    else hipLaunchKernelGGL(eltwise_v8<__bf16>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, (const __bf16*)b, (__bf16*)y, C, Cb, op, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

template <typename S>
static int cast_from(const S* src, void* dst, int ddt, int Cs, int Cd, size_t total, hipStream_t st) {
    if (Cs == Cd && (total & 7) == 0) {
        const size_t t8 = total / 8;
        switch (ddt) {
        case IE_F32: hipLaunchKernelGGL((cast8_kernel<S, float>), dim3(nblk(t8)), dim3(256), 0, st, src, (float*)dst, t8); break;
        case IE_F16: hipLaunchKernelGGL((cast8_kernel<S, _Float16>), dim3(nblk(t8)), dim3(256), 0, st, src, (_Float16*)dst, t8); break;
        case IE_BF16: hipLaunchKernelGGL((cast8_kernel<S, __bf16>), dim3(nblk(t8)), dim3(256), 0, st, src, (__bf16*)dst, t8); break;
        default: IE_FAIL(IE_ERR_INVALID, "ie_cast_nhwc: bad destination dtype");
        }
        IE_LAUNCH_CHECK();
        return IE_OK;
    }
    if ((ddt == IE_F16 || ddt == IE_BF16) && (Cd & 7) == 0 && c8_shift_of(Cd / 8) >= 0) {
        const size_t groups = total / 8;
        if (ddt == IE_F16) hipLaunchKernelGGL((cast_pad8_kernel<S, _Float16>), dim3(nblk(groups)), dim3(256), 0, st, src, (_Float16*)dst, Cs, c8_shift_of(Cd / 8), groups);
        else hipLaunchKernelGGL((cast_pad8_kernel<S, __bf16>), dim3(nblk(groups)), dim3(256), 0, st, src, (__bf16*)dst, Cs, c8_shift_of(Cd / 8), groups);
        IE_LAUNCH_CHECK();
        return IE_OK;
    }
    switch (ddt) {
    case IE_F32: hipLaunchKernelGGL((cast_kernel<S, float>), dim3(nblk(total)), dim3(256), 0, st, src, (float*)dst, Cs, Cd, total); break;
    case IE_F16: hipLaunchKernelGGL((cast_kernel<S, _Float16>), dim3(nblk(total)), dim3(256), 0, st, src, (_Float16*)dst, Cs, Cd, total); break;
    case IE_BF16: hipLaunchKernelGGL((cast_kernel<S, __bf16>), dim3(nblk(total)), dim3(256), 0, st, src, (__bf16*)dst, Cs, Cd, total); break;
    default: IE_FAIL(IE_ERR_INVALID, "ie_cast_nhwc: bad destination dtype");
    }
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_cast_nhwc(const void* src, int src_dtype, int Cs, void* dst, int dst_dtype, int Cd, size_t npix, void* stream) {
    if (!src || !dst || npix == 0 || Cs <= 0 || Cd <= 0) IE_FAIL(IE_ERR_INVALID, "ie_cast_nhwc: bad args");
    const size_t total = npix * Cd;
    hipStream_t st = (hipStream_t)stream;
    switch (src_dtype) {
    case IE_F32: return cast_from((const float*)src, dst, dst_dtype, Cs, Cd, total, st);
    case IE_F16: return cast_from((const _Float16*)src, dst, dst_dtype, Cs, Cd, total, st);
    case IE_BF16: return cast_from((const __bf16*)src, dst, dst_dtype, Cs, Cd, total, st);
    default: IE_FAIL(IE_ERR_INVALID, "ie_cast_nhwc: bad source dtype");
    }
}

IE_EXPORT int ie_concat_cast_nhwc(const float* const* srcs, const int* channels, int nsrc, void* dst, int dst_dtype, int Cd, size_t npix, void* stream) {
    if (!srcs || !channels || !dst || nsrc < 1 || nsrc > 4 || npix == 0 || Cd <= 0) IE_FAIL(IE_ERR_INVALID, "ie_concat_cast_nhwc: bad args (1 .. 4 sources)");
    CatSrcs s;
    int sum = 0;
    for (int k = 0; k < 4; ++k) {
        s.p[k] = k < nsrc ? srcs[k] : nullptr; s.c[k] = k < nsrc ? channels[k] : 0;
        if (k < nsrc && (!srcs[k] || channels[k] <= 0)) IE_FAIL(IE_ERR_INVALID, "ie_concat_cast_nhwc: source %d", k);
        sum += s.c[k];
    }
    s.n = nsrc;
    if (sum > Cd) IE_FAIL(IE_ERR_INVALID, "ie_concat_cast_nhwc: %d source channels do not fit %d", sum, Cd);
    const size_t total = npix * Cd;
    hipStream_t st = (hipStream_t)stream;
    switch (dst_dtype) {
    case IE_F32: hipLaunchKernelGGL(concat_cast_kernel<float>, dim3(nblk(total)), dim3(256), 0, st, s, (float*)dst, Cd, total); break;
    case IE_F16: hipLaunchKernelGGL(concat_cast_kernel<_Float16>, dim3(nblk(total)), dim3(256), 0, st, s, (_Float16*)dst, Cd, total); break;
    case IE_BF16: hipLaunchKernelGGL(concat_cast_kernel<__bf16>, dim3(nblk(total)), dim3(256), 0, st, s, (__bf16*)dst, Cd, total); break;
    default: IE_FAIL(IE_ERR_INVALID, "ie_concat_cast_nhwc: bad destination dtype");
    }
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_instnorm_nhwc(const void* x, const float* gamma, const float* beta, void* y, int dtype, double* stats_ws, int N, int H, int W,
                               int C, int c_begin, int c_count, float eps, int act, float slope, void* stream) {
    if (dtype == IE_F32) return ie_instnorm_nhwc_f32((const float*)x, gamma, beta, (float*)y, stats_ws, N, H, W, C, c_begin, c_count, eps, act, slope, stream);
    if (!x || !y || !stats_ws || !is16(dtype) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || c_begin < 0 || (c_begin & 7) || c_count <= 0 ||
        (c_count & 7) || c_count > 2048 || c_begin + c_count > C || (gamma == nullptr) != (beta == nullptr))
        IE_FAIL(IE_ERR_INVALID, "ie_instnorm_nhwc: bad args (channel counts multiples of 8 on the 16-bit path)");
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    IE_HIP_CHECK(hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * (size_t)N * c_count, st));
    float* mean = (float*)(stats_ws + 2 * (size_t)N * c_count);
    float* inv = mean + (size_t)N * c_count;
    const int g8 = c_count / 8;
    const int tpc = 256 / g8 > 0 ? 256 / g8 : 1;
    int chunks = ie_cdiv(HW, tpc * 32);
    if (chunks > 512) chunks = 512;
    if (chunks < 1) chunks = 1;
    if (dtype == IE_F16) hipLaunchKernelGGL(norm16_stats<_Float16>, dim3(chunks, N), dim3(256), 0, st, (const _Float16*)x, stats_ws, HW, C, c_begin, c_count);
    else hipLaunchKernelGGL(norm16_stats<__bf16>, dim3(chunks, N), dim3(256), 0, st, (const __bf16*)x, stats_ws, HW, C, c_begin, c_count);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm16_finalize, dim3(nblk((size_t)N * c_count)), dim3(256), 0, st, stats_ws, mean, inv, N * c_count, (double)HW, eps);
    IE_LAUNCH_CHECK();
    const int C8 = C / 8, nchb = C8 < 256 ? C8 : 256, atpc = 256 / nchb;
    int achunks = ie_cdiv(HW, atpc * 8);                     // ~8 pixels per thread, at most ~8192 workgroups
    const int acap = ie_cdiv(8192, N);
    if (achunks > acap) achunks = acap;
    if (achunks < 1) achunks = 1;
    const dim3 agrid(achunks, N, ie_cdiv(C8, nchb));
    if (dtype == IE_F16) hipLaunchKernelGGL(norm16_apply<_Float16>, agrid, dim3(256), 0, st, (const _Float16*)x, mean, inv, gamma, beta, (_Float16*)y, HW, C, c_begin, c_count, act, slope, nchb);
    else hipLaunchKernelGGL(norm16_apply<__bf16>, agrid, dim3(256), 0, st, (const __bf16*)x, mean, inv, gamma, beta, (__bf16*)y, HW, C, c_begin, c_count, act, slope, nchb);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_dwconv3x3_nhwc(const void* x, const float* w_tap_c, const float* bias, void* y, int dtype, int N, int H, int W, int C, int act,
                                float slope, void* stream) {
    if (!x || !w_tap_c || !y || !is16(dtype) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || C > 1024)
        IE_FAIL(IE_ERR_INVALID, "ie_dwconv3x3_nhwc: bad args (16-bit, C %% 8 == 0, C <= 1024)");
    const size_t total = (size_t)N * H * W * (C / 8);
    const size_t lds = sizeof(float) * 10 * C;
    if (dtype == IE_F16) hipLaunchKernelGGL(dwconv3x3_v8<_Float16>, dim3(nblk(total)), dim3(256), lds, (hipStream_t)stream, (const _Float16*)x, w_tap_c, bias, (_Float16*)y, H, W, C, act, slope, total);
    else hipLaunchKernelGGL(dwconv3x3_v8<__bf16>, dim3(nblk(total)), dim3(256), lds, (hipStream_t)stream, (const __bf16*)x, w_tap_c, bias, (__bf16*)y, H, W, C, act, slope, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
