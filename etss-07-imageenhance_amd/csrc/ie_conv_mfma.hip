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

// activation with the hardware exp2/rcp forms (|err| <= 3e-7): one code path per template instance keeps the epilogue small
template <int ACT> __device__ __forceinline__ float act_fast(float v, float slope) {
    if constexpr (ACT == IE_ACT_RELU) return fmaxf(v, 0.f);
    else if constexpr (ACT == IE_ACT_LRELU) return v > 0.f ? v : v * slope;
    else if constexpr (ACT == IE_ACT_TANH) return ie_tanh_fast(v);
    else if constexpr (ACT == IE_ACT_SIGMOID) return ie_sigmoid_fast(v);
    else return v;
}

template <typename T, int NF, int ACT>
__device__ __forceinline__ void mconv_epilogue(const MConvParams& p, ie_f32x4 (&acc)[8][NF], int n, int ct, int oy0, int ox0, int wave, int kc,
                                            int l16) {
    float bv[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = (ct * NF + f) * 16 + kc * 4 + j;
            bv[f][j] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        }
    const bool quad = (p.Cout & 3) == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int oy = oy0 + wave * 4 + (i >> 1), ox = ox0 + (i & 1) * 16 + l16;
        bool live = oy < p.Ho && ox < p.Wo;
        int sy = oy, sx = ox;
        if (p.sub > 1) {
            live = live && !(oy % p.sub) && !(ox % p.sub);
            sy = oy / p.sub; sx = ox / p.sub;
        }
        if (!live) continue;
        const size_t opix = ((size_t)n * p.Hs + sy) * p.Ws + sx;
        float v[NF][4];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[f][j] = act_fast<ACT>(acc[i][f][j] + bv[f][j], p.slope);
        if (p.res) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int co = (ct * NF + f) * 16 + kc * 4;
                const T* rp = (const T*)p.res + opix * p.Cout + co;
                if (quad && co + 3 < p.Cout) {
                    typedef __attribute__((ext_vector_type(4))) T t4;
                    const t4 r4 = *(const t4*)rp;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[f][j] += (float)r4[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < p.Cout) v[f][j] += (float)rp[j];
                }
            }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int co = (ct * NF + f) * 16 + kc * 4;
            if (co >= p.Cout) continue;
            if (p.clamp) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[f][j] = fminf(fmaxf(v[f][j], p.lo), p.hi);
            }
            const size_t o = opix * p.Cout + co;
            const bool full = quad && (co + 3 < p.Cout);
            if (p.out_f32) {
                float* dp = (float*)p.dst + o;
                if (full) *(ie_f32x4*)dp = ie_f32x4{v[f][0], v[f][1], v[f][2], v[f][3]};
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < p.Cout) dp[j] = v[f][j];
            } else {
                T* dp = (T*)p.dst + o;
                if (full) *(ie_u32x2*)dp = ie_u32x2{ie_pack2<T>(v[f][0], v[f][1]), ie_pack2<T>(v[f][2], v[f][3])};
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < p.Cout) dp[j] = (T)v[f][j];
            }
        }
    }
}

template <typename T, int NF, bool K3>
__global__ __launch_bounds__(256, 2) void mconv_kernel(MConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using frag = typename ie_elem<T>::frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int KH = K3 ? 3 : p.kh, KW = K3 ? 3 : p.kw;
    const int TWH = TW + KW - 1, THH = TH + KH - 1, TPX = TWH * THH;
    const int ND = (TPX + 15) >> 4;            // 1-KB (16-pixel) LDS-DMA instructions that fill the input tile
    unsigned char* in_lds = smem;
    unsigned char* w_lds = smem + (size_t)ND * 1024;
    const int kc = lane >> 4, l16 = lane & 15;

    ie_f32x4 acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = ie_f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned w_stage_bytes = (unsigned)p.taps_stage * 16 * NF * 64;
    const unsigned char* wpk = (const unsigned char*)p.wpk + (size_t)ct * p.nchunks * p.nstages * w_stage_bytes;
    constexpr unsigned OOB = 0xFFFFFFF0u;      // beyond every buffer range (host-checked): the load returns zeros

    for (int ch = 0; ch < p.nchunks; ++ch) {
        const int s = p.chunk_src[ch], c0 = p.chunk_c0[ch], Cs = p.cs[s];
        const size_t frame_b = (size_t)p.H * p.W * Cs * sizeof(T);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src[s] + (size_t)n * frame_b), 0, (int)frame_b, 0x00020000);
        __syncthreads();                        // the previous chunk's fragments have been read
        // input halo tile by LDS-DMA: lane -> (pixel q, 16-byte slot); the slot holds channel piece slot ^ ((q>>1)&3).
        // Zero padding, the tile's overhang and the zero fill of a short chunk are all "out-of-range offset -> zeros".
        for (int d = wave; d < ND; d += 4) {
            const int q = d * 16 + (lane >> 2);
            const int j = (lane & 3) ^ ((q >> 1) & 3);
            const int r = q / TWH, c = q - r * TWH;
            int iy = oy0 + r - p.pad, ix = ox0 + c - p.pad;
            bool ok = (c0 + j * 8 < Cs) && (q < TPX);
            if (p.pad_mode == IE_PAD_REPLICATE) { iy = min(max(iy, 0), p.H - 1); ix = min(max(ix, 0), p.W - 1); }
            else ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const unsigned off = ok ? (unsigned)(((unsigned)(iy * p.W + ix) * (unsigned)Cs + (unsigned)(c0 + j * 8)) * sizeof(T)) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(in_lds + d * 1024), 16, (int)off, 0, 0, 0);
        }
        for (int st = 0; st < p.nstages; ++st) {
            if (st > 0) __syncthreads();        // previous stage's weights have been read
            const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(wpk + ((size_t)ch * p.nstages + st) * w_stage_bytes), 0,
                                                              (int)w_stage_bytes, 0x00020000);
            for (int d = wave; d < p.taps_stage * NF; d += 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(w_lds + d * 1024), 16, d * 1024 + lane * 16, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if constexpr (K3) {
                // 3x3: every pixel fragment (input row, column half, kx) feeds the MFMAs of up to three output rows (ky = 0,1,2)
#pragma unroll 1
                for (int kx = 0; kx < 3; ++kx) {
                    frag a[3][NF];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const int row = ((ky * 3 + kx) * NF + f) * 16 + l16;
                            a[ky][f] = *(const frag*)(w_lds + (size_t)row * 64 + ((kc ^ ((l16 >> 1) & 3)) << 4));
                        }
#pragma unroll
                    for (int ir = 0; ir < 6; ++ir) {
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const int q = (wave * 4 + ir) * TWH + hf * 16 + l16 + kx;
                            const frag b = *(const frag*)(in_lds + (size_t)q * 64 + ((kc ^ ((q >> 1) & 3)) << 4));
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky) {
                                const int orow = ir - ky;
                                if (orow < 0 || orow > 3) continue;
#pragma unroll
                                for (int f = 0; f < NF; ++f) acc[orow * 2 + hf][f] = ie_elem<T>::mfma16(a[ky][f], b, acc[orow * 2 + hf][f]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);      // keep the fragment reads of later rows from being hoisted (VGPR budget)
                    }
                }
            } else {
                const int tap0 = st * p.taps_stage;
                for (int t = 0; t < p.taps_stage; ++t) {
                    const int tap = tap0 + t;
                    if (tap >= KH * KW) break;
                    const int ky = tap / KW, kx = tap - ky * KW;
                    frag a[NF];
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const int row = (t * NF + f) * 16 + l16;
                        a[f] = *(const frag*)(w_lds + (size_t)row * 64 + ((kc ^ ((l16 >> 1) & 3)) << 4));
                    }
                    const int dq = ky * TWH + kx;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int q = (wave * 4 + (i >> 1)) * TWH + (i & 1) * 16 + l16 + dq;
                        const frag b = *(const frag*)(in_lds + (size_t)q * 64 + ((kc ^ ((q >> 1) & 3)) << 4));
#pragma unroll
                        for (int f = 0; f < NF; ++f) acc[i][f] = ie_elem<T>::mfma16(a[f], b, acc[i][f]);
                    }
                }
            }
        }
    }

    // epilogue: lane holds couts co0 + 4*kc + {0..3} of pixel (row, col + l16) for each (i, f)
    switch (p.act) {
    case IE_ACT_RELU: mconv_epilogue<T, NF, IE_ACT_RELU>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_LRELU: mconv_epilogue<T, NF, IE_ACT_LRELU>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_TANH: mconv_epilogue<T, NF, IE_ACT_TANH>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_SIGMOID: mconv_epilogue<T, NF, IE_ACT_SIGMOID>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    default: mconv_epilogue<T, NF, IE_ACT_NONE>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
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
    const size_t lds = (size_t)(((TH + d->kh - 1) * (TW + d->kw - 1) + 15) / 16) * 1024 + pl.stage_bytes;
    for (int s = 0; s < d->nsrc; ++s)
        if ((double)d->H * d->W * channels[s] * 2.0 >= 4294967040.0) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: a frame of source %d exceeds 4 GiB", s);
    if (lds > 160 * 1024) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: %dx%d kernel needs %zu bytes of LDS", d->kh, d->kw, lds);
    void (*fn)(MConvParams) = nullptr;
    const bool h = d->in_dtype == IE_F16;
    const bool k3 = d->kh == 3 && d->kw == 3;
    switch (pl.nf * 2 + (k3 ? 1 : 0)) {
    case 2: fn = h ? mconv_kernel<_Float16, 1, false> : mconv_kernel<__bf16, 1, false>; break;
    case 3: fn = h ? mconv_kernel<_Float16, 1, true> : mconv_kernel<__bf16, 1, true>; break;
    case 4: fn = h ? mconv_kernel<_Float16, 2, false> : mconv_kernel<__bf16, 2, false>; break;
    case 5: fn = h ? mconv_kernel<_Float16, 2, true> : mconv_kernel<__bf16, 2, true>; break;
    case 8: fn = h ? mconv_kernel<_Float16, 4, false> : mconv_kernel<__bf16, 4, false>; break;
    default: fn = h ? mconv_kernel<_Float16, 4, true> : mconv_kernel<__bf16, 4, true>; break;
    }
    if (lds > 64 * 1024) {       // opt in to > 64 KB of dynamic LDS once per kernel (raised monotonically)
        static size_t granted[12] = {0};
        const int slot = (h ? 0 : 6) + (pl.nf == 1 ? 0 : (pl.nf == 2 ? 1 : 2)) * 2 + (k3 ? 1 : 0);
        if (granted[slot] < lds) {
            IE_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            granted[slot] = 160 * 1024;
        }
    }
    hipLaunchKernelGGL(fn, dim3(p.total_blocks), dim3(256), lds, (hipStream_t)stream, p);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
