// Generic implicit-GEMM convolution on the matrix cores for the U-Net / RetinexNet / LLUnet++ / HINet generators:
// 16-bit (fp16 or bf16) NHWC activations, fp32 accumulation, any odd/even kernel size with stride 1 (stride 2 =
// the stride-1 result sub-sampled in the epilogue), zero or replicate padding, virtual channel concat of up to
// IE_CONV_MAX_SRC sources, fused bias + activation + residual + clamp, 16-bit or fp32 output.
//
// GEMM view: D[cout][pixel] += W[cout][k] * X[k][pixel], k = (tap, cin), on v_mfma_f32_16x16x32_{f16,bf16}.
// Workgroup (4 waves) = 16 x 32 output pixels x (16*NF) output channels; wave = 4 rows x 32 columns x 16*NF couts
// (8 pixel fragments x NF cout fragments = 32 accumulator tiles at NF=4).  K is walked in chunks of 32 input channels
// (chunks never straddle two sources; a short chunk is zero-filled): per chunk the input halo tile
// [(16+kh-1) x (32+kw-1) px][32 ch] and the weights [taps][16*NF cout][32 ch] are staged in LDS (64-byte rows whose 16-byte
// pieces are XOR-swizzled by (row>>1)&3: every 16-lane x 4-piece ds_read_b128 fragment is bank-conflict free, the same
// rule as the Zero-DCE strip kernels), then every tap issues NF + 8 fragment reads for 8*NF MFMAs.  Two workgroups per
// CU (<= 78 KB LDS each for 3x3) overlap one's loads with the other's MFMAs.
// Roofline: 3x3, Cin = Cout = 64: 37.7 MFLOP per 142 KB of HBM traffic per tile = 265 FLOP/B, just under the 312 FLOP/B
// machine balance -> the kernel is MFMA-bound only if the input tile is shared through L2 by the cout tiles of the same
// pixels; the block index is therefore XCD-swizzled so that those workgroups run on the same XCD.
#include "ie_common.h"
#include <string.h>
#include <vector>

namespace {

constexpr int TH = 16, TW = 32, KC = 32;       // output tile, channels per K chunk
constexpr int MAX_CHUNKS = 96;

struct MConvParams {
    const void* src[IE_CONV_MAX_SRC];
    int cs[IE_CONV_MAX_SRC];
    unsigned char chunk_src[MAX_CHUNKS];
    unsigned short chunk_c0[MAX_CHUNKS];
    int nchunks;
    const void* wpk; const float* bias; const void* res; void* dst;
    int N, H, W, Cout, kh, kw, pad, pad_mode, sub, act, clamp, out_f32;
    int Ho, Wo;                  // stride-1 output extent
    int Hs, Ws;                  // stored extent (= Ho, Wo, or the sub-sampled size)
    int tiles_x, tiles_y, cout_tiles, total_blocks;
    int taps_stage, nstages;     // weight staging: taps per LDS stage
    float slope, lo, hi;
};

template <typename T, int NF>
__global__ __launch_bounds__(256, 2) void mconv_kernel(MConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using frag = typename ie_elem<T>::frag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block order: physical id -> logical id so that consecutive logical ids (the cout tiles of one pixel tile,
    // then the neighbouring pixel tile) share an XCD and therefore an L2
    const int pb = blockIdx.x;
    const int per_xcd = p.total_blocks >> 3;
    int lb = (pb & 7) * per_xcd + (pb >> 3);
    const int ct = lb % p.cout_tiles; lb /= p.cout_tiles;
    const int tx = lb % p.tiles_x; lb /= p.tiles_x;
    const int ty = lb % p.tiles_y;
    const int n = lb / p.tiles_y;
    if (n >= p.N) return;                      // padding blocks (uniform over the workgroup)
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int TWH = TW + p.kw - 1, THH = TH + p.kh - 1, TPX = TWH * THH;
    unsigned char* in_lds = smem;
    unsigned char* w_lds = smem + (size_t)TPX * 64;
    const int kc = lane >> 4, l16 = lane & 15;

    ie_f32x4 acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = ie_f32x4{0.f, 0.f, 0.f, 0.f};

    int q0[8];                                  // LDS pixel index of each of the wave's 8 pixel fragments at tap (0,0)
#pragma unroll
    for (int i = 0; i < 8; ++i) q0[i] = (wave * 4 + (i >> 1)) * TWH + (i & 1) * 16 + l16;

    const size_t w_stage_bytes = (size_t)p.taps_stage * 16 * NF * 64;
    const unsigned char* wpk = (const unsigned char*)p.wpk + (size_t)ct * p.nchunks * p.nstages * w_stage_bytes;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        const int s = p.chunk_src[ch], c0 = p.chunk_c0[ch], Cs = p.cs[s];
        const T* sp = (const T*)p.src[s] + (size_t)n * p.H * p.W * Cs + c0;
        __syncthreads();                        // the previous chunk's fragments have been read
        for (int i = tid; i < TPX * 4; i += 256) {
            const int q = i >> 2, j = i & 3;
            const int r = q / TWH, c = q - r * TWH;
            int iy = oy0 + r - p.pad, ix = ox0 + c - p.pad;
            bool ok = c0 + j * 8 < Cs;
            if (p.pad_mode == IE_PAD_REPLICATE) { iy = min(max(iy, 0), p.H - 1); ix = min(max(ix, 0), p.W - 1); }
            else ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            ie_u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *(const ie_u32x4*)(sp + ((size_t)iy * p.W + ix) * Cs + j * 8);
            *(ie_u32x4*)(in_lds + (size_t)q * 64 + ((j ^ ((q >> 1) & 3)) << 4)) = v;
        }
        for (int st = 0; st < p.nstages; ++st) {
            if (st > 0) __syncthreads();        // previous stage's weights have been read
            const ie_u32x4* wsrc = (const ie_u32x4*)(wpk + ((size_t)ch * p.nstages + st) * w_stage_bytes);
            for (int i = tid; i < (int)(w_stage_bytes >> 4); i += 256) ((ie_u32x4*)w_lds)[i] = wsrc[i];
            __syncthreads();
            const int tap0 = st * p.taps_stage;
            for (int t = 0; t < p.taps_stage; ++t) {
                const int tap = tap0 + t;
                if (tap >= p.kh * p.kw) break;
                const int ky = tap / p.kw, kx = tap - ky * p.kw;
                frag a[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int row = (t * NF + f) * 16 + l16;
                    a[f] = *(const frag*)(w_lds + (size_t)row * 64 + ((kc ^ ((l16 >> 1) & 3)) << 4));
                }
                const int dq = ky * TWH + kx;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = q0[i] + dq;
                    const frag b = *(const frag*)(in_lds + (size_t)q * 64 + ((kc ^ ((q >> 1) & 3)) << 4));
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[i][f] = ie_elem<T>::mfma16(a[f], b, acc[i][f]);
                }
            }
        }
    }

    // epilogue: lane holds couts co0 + 4*kc + {0..3} of pixel (row, col + l16) for each (i, f)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int oy = oy0 + wave * 4 + (i >> 1), ox = ox0 + (i & 1) * 16 + l16;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        int sy = oy, sx = ox;
        if (p.sub > 1) {
            if ((oy % p.sub) || (ox % p.sub)) continue;
            sy = oy / p.sub; sx = ox / p.sub;
        }
        const size_t opix = ((size_t)n * p.Hs + sy) * p.Ws + sx;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int co = (ct * NF + f) * 16 + kc * 4;
            if (co >= p.Cout) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = acc[i][f][j];
                if (p.bias && co + j < p.Cout) x += p.bias[co + j];
                x = ie_act_apply(x, p.act, p.slope);
                v[j] = x;
            }
            const size_t o = opix * p.Cout + co;
            const bool full = (co + 3 < p.Cout) && ((p.Cout & 3) == 0);
            if (p.res) {
                const T* rp = (const T*)p.res + o;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (co + j < p.Cout) v[j] += (float)rp[j];
            }
            if (p.clamp) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j], p.lo), p.hi);
            }
            if (p.out_f32) {
                float* dp = (float*)p.dst + o;
                if (full) *(ie_f32x4*)dp = ie_f32x4{v[0], v[1], v[2], v[3]};
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < p.Cout) dp[j] = v[j];
            } else {
                T* dp = (T*)p.dst + o;
                if (full) *(ie_u32x2*)dp = ie_u32x2{ie_pack2<T>(v[0], v[1]), ie_pack2<T>(v[2], v[3])};
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < p.Cout) dp[j] = (T)v[j];
            }
        }
    }
}

int pick_nf(int Cout) { return Cout > 32 ? 4 : (Cout > 16 ? 2 : 1); }

struct Plan {
    int nf, cout_tiles, nchunks, taps, taps_stage, nstages;
    unsigned char chunk_src[MAX_CHUNKS];
    unsigned short chunk_c0[MAX_CHUNKS];
    size_t stage_bytes, total_bytes;
};

int make_plan(const int* cs, int nsrc, int Cout, int kh, int kw, Plan* pl, const char* who) {
    if (!cs || nsrc < 1 || nsrc > IE_CONV_MAX_SRC || Cout <= 0 || kh <= 0 || kw <= 0 || kh > 15 || kw > 15)
        IE_FAIL(IE_ERR_INVALID, "%s: bad shape", who);
    pl->nf = pick_nf(Cout);
    pl->cout_tiles = ie_cdiv(Cout, 16 * pl->nf);
    pl->nchunks = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (cs[s] <= 0 || (cs[s] & 7)) IE_FAIL(IE_ERR_UNSUPPORTED, "%s: source %d has %d channels; the 16-bit path needs multiples of 8", who, s, cs[s]);
        for (int c0 = 0; c0 < cs[s]; c0 += KC) {
            if (pl->nchunks >= MAX_CHUNKS) IE_FAIL(IE_ERR_UNSUPPORTED, "%s: more than %d K chunks", who, MAX_CHUNKS);
            pl->chunk_src[pl->nchunks] = (unsigned char)s;
            pl->chunk_c0[pl->nchunks] = (unsigned short)c0;
            ++pl->nchunks;
        }
    }
    pl->taps = kh * kw;
    pl->taps_stage = pl->taps <= 9 ? pl->taps : kw;          // all taps at once for <= 3x3, else one kernel row per stage
    pl->nstages = ie_cdiv(pl->taps, pl->taps_stage);
    pl->stage_bytes = (size_t)pl->taps_stage * 16 * pl->nf * 64;
    pl->total_bytes = (size_t)pl->cout_tiles * pl->nchunks * pl->nstages * pl->stage_bytes;
    return IE_OK;
}

unsigned short f32_to_16(float v, int dtype) {
    if (dtype == IE_F16) { _Float16 h = (_Float16)v; unsigned short u; memcpy(&u, &h, 2); return u; }
    unsigned int x; memcpy(&x, &v, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((x >> 16) | 0x40);     // NaN
    x += 0x7fffu + ((x >> 16) & 1u);                                                    // round to nearest even
    return (unsigned short)(x >> 16);
}

}  // namespace

IE_EXPORT size_t ie_mconv_packed_bytes(const int* channels, int nsrc, int Cout, int kh, int kw) {
    Plan pl;
    if (make_plan(channels, nsrc, Cout, kh, kw, &pl, "ie_mconv_packed_bytes") != IE_OK) return 0;
    return pl.total_bytes;
}

IE_EXPORT int ie_mconv_pack_weights(const float* w_oihw_host, const int* channels, int nsrc, int Cout, int kh, int kw, int dtype,
                                    void* dst_dev, void* stream) {
    Plan pl;
    int rc = make_plan(channels, nsrc, Cout, kh, kw, &pl, "ie_mconv_pack_weights");
    if (rc != IE_OK) return rc;
    if (!w_oihw_host || !dst_dev || (dtype != IE_F16 && dtype != IE_BF16)) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pack_weights: bad args");
    int Cin = 0, cbase[IE_CONV_MAX_SRC];
    for (int s = 0; s < nsrc; ++s) { cbase[s] = Cin; Cin += channels[s]; }
    std::vector<unsigned short> host(pl.total_bytes / 2, 0);
    const int rows = 16 * pl.nf;
    for (int ct = 0; ct < pl.cout_tiles; ++ct)
        for (int ch = 0; ch < pl.nchunks; ++ch) {
            const int s = pl.chunk_src[ch], c0 = pl.chunk_c0[ch];
            const int valid = channels[s] - c0 < KC ? channels[s] - c0 : KC;
            for (int st = 0; st < pl.nstages; ++st) {
                unsigned short* blk = host.data() + (((size_t)ct * pl.nchunks + ch) * pl.nstages + st) * (pl.stage_bytes / 2);
                for (int t = 0; t < pl.taps_stage; ++t) {
                    const int tap = st * pl.taps_stage + t;
                    if (tap >= pl.taps) break;
                    const int ky = tap / kw, kx = tap % kw;
                    for (int r = 0; r < rows; ++r) {
                        const int co = ct * rows + r;
                        if (co >= Cout) continue;
                        unsigned short* row = blk + (size_t)(t * rows + r) * 32;
                        const int sw = ((r & 15) >> 1) & 3;
                        for (int k = 0; k < valid; ++k) {
                            const float v = w_oihw_host[(((size_t)co * Cin + cbase[s] + c0 + k) * kh + ky) * kw + kx];
                            row[(((k >> 3) ^ sw) << 3) + (k & 7)] = f32_to_16(v, dtype);
                        }
                    }
                }
            }
        }
    IE_HIP_CHECK(hipMemcpyAsync(dst_dev, host.data(), pl.total_bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    IE_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));     // `host` is freed on return
    return IE_OK;
}

IE_EXPORT int ie_mconv_nhwc(ie_mconv_desc* d, const void* const* srcs, const int* channels, const void* packed_w, const float* bias,
                            const void* residual, void* dst, void* stream) {
    if (!d || !srcs || !channels || !packed_w || !dst) IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: null pointer");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->pad < 0 || (d->stride != 1 && d->stride != 2))
        IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: bad shape (stride must be 1 or 2)");
    if ((d->in_dtype != IE_F16 && d->in_dtype != IE_BF16) || (d->out_dtype != d->in_dtype && d->out_dtype != IE_F32))
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: in_dtype must be f16/bf16 and out_dtype the same or f32");
    Plan pl;
    int rc = make_plan(channels, d->nsrc, d->Cout, d->kh, d->kw, &pl, "ie_mconv_nhwc");
    if (rc != IE_OK) return rc;
    MConvParams p;
    memset(&p, 0, sizeof(p));
    for (int s = 0; s < d->nsrc; ++s) {
        if (!srcs[s]) IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: source %d is null", s);
        p.src[s] = srcs[s]; p.cs[s] = channels[s];
    }
    memcpy(p.chunk_src, pl.chunk_src, sizeof(p.chunk_src));
    memcpy(p.chunk_c0, pl.chunk_c0, sizeof(p.chunk_c0));
    p.nchunks = pl.nchunks; p.wpk = packed_w; p.bias = bias; p.res = residual; p.dst = dst;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.kh = d->kh; p.kw = d->kw; p.pad = d->pad; p.pad_mode = d->pad_mode;
    p.sub = d->stride; p.act = d->act; p.clamp = d->clamp; p.out_f32 = d->out_dtype == IE_F32; p.slope = d->slope; p.lo = d->clamp_lo;
    p.hi = d->clamp_hi;
    p.Ho = d->H + 2 * d->pad - d->kh + 1; p.Wo = d->W + 2 * d->pad - d->kw + 1;
    if (p.Ho <= 0 || p.Wo <= 0) IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: empty output");
    p.Hs = (d->H + 2 * d->pad - d->kh) / d->stride + 1; p.Ws = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    d->Ho = p.Hs; d->Wo = p.Ws;
    p.tiles_x = ie_cdiv(p.Wo, TW); p.tiles_y = ie_cdiv(p.Ho, TH); p.cout_tiles = pl.cout_tiles;
    const long logical = (long)p.tiles_x * p.tiles_y * p.cout_tiles * d->N;
    p.total_blocks = (int)((logical + 7) / 8 * 8);
    p.taps_stage = pl.taps_stage; p.nstages = pl.nstages;
    const size_t lds = (size_t)(TH + d->kh - 1) * (TW + d->kw - 1) * 64 + pl.stage_bytes;
    if (lds > 160 * 1024) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: %dx%d kernel needs %zu bytes of LDS", d->kh, d->kw, lds);
    void (*fn)(MConvParams) = nullptr;
    const bool h = d->in_dtype == IE_F16;
    switch (pl.nf) {
    case 1: fn = h ? mconv_kernel<_Float16, 1> : mconv_kernel<__bf16, 1>; break;
    case 2: fn = h ? mconv_kernel<_Float16, 2> : mconv_kernel<__bf16, 2>; break;
    default: fn = h ? mconv_kernel<_Float16, 4> : mconv_kernel<__bf16, 4>; break;
    }
    if (lds > 64 * 1024) {       // opt in to > 64 KB of dynamic LDS once per kernel (raised monotonically)
        static size_t granted[6] = {0, 0, 0, 0, 0, 0};
        const int slot = (h ? 0 : 3) + (pl.nf == 1 ? 0 : (pl.nf == 2 ? 1 : 2));
        if (granted[slot] < lds) {
            IE_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            granted[slot] = 160 * 1024;
        }
    }
    hipLaunchKernelGGL(fn, dim3(p.total_blocks), dim3(256), lds, (hipStream_t)stream, p);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
