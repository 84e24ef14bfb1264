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
#include "ie_conv_internal.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

constexpr int TH = 16, TW = 32, KC = 32;       // output tile, channels per K chunk
constexpr int MAX_CHUNKS = 96;

struct MConvParams {
    const void* src[IE_CONV_MAX_SRC];
    int cs[IE_CONV_MAX_SRC];
    int chunk_tab[MAX_CHUNKS];   // (source << 16) | first channel; dwords so that the wave-uniform lookups are scalar loads
    const void* cd_base[MAX_CHUNKS];     // per chunk, flattened (one scalar load, no dependent lookups): source tensor ...
    unsigned cd_info[MAX_CHUNKS];        // ... and (bytes per pixel of that source) | (byte offset of the chunk's first channel << 16)
    int nchunks;
    const void* wpk; const float* bias; const void* res; void* dst;
    int N, H, W, Cout, kh, kw, pad, pad_mode, sub, act, clamp, out_f32;
    int Ho, Wo;                  // stride-1 output extent
    int Hs, Ws;                  // stored extent (= Ho, Wo, or the sub-sampled size)
    int sh[IE_CONV_MAX_SRC], sw[IE_CONV_MAX_SRC];   // stored size of source s (== H, W unless the conv reads it through a nearest-neighbour up-sampling)
    void* pool;                  // mconv3_kernel, ReLU layers: also write nn.MaxPool2d(2) of the output (N x Ho/2 x Wo/2 x Cout), or null
    int shuf_c;                  // fused PixelShuffle(2): channels of the shuffled output (Cout = 4 shuf_c, rows packed as [i][j][c]); 0 = off
    int tiles_x, tiles_y, cout_tiles, total_blocks;
    int taps_stage, nstages;     // weight staging: taps per LDS stage
    float slope, lo, hi;
#ifdef IE_MSTAMP
    unsigned long long* stamps;  // diagnostic build only: 64 s_memtime stamps per workgroup
#endif
};

#ifdef IE_MSTAMP
#define MST(i) do { if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MST(i) do { } while (0)
#endif

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
        const int s = p.chunk_tab[ch] >> 16, c0 = p.chunk_tab[ch] & 0xffff, Cs = p.cs[s];
        const int SH = p.sh[s], SW = p.sw[s];
        const bool up = SH != p.H || SW != p.W;        // F.interpolate(size=(H, W)) in front of the conv (retinexnet/model.py:79-86), folded into the addressing
        const float fy = (float)SH / (float)p.H, fx = (float)SW / (float)p.W;
        const size_t frame_b = (size_t)SH * SW * Cs * sizeof(T);
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
            if (up) { iy = min((int)floorf(iy * fy), SH - 1); ix = min((int)floorf(ix * fx), SW - 1); }     // legacy nearest: src = floor(dst * in / out)
            const unsigned off = ok ? (unsigned)(((unsigned)(iy * SW + ix) * (unsigned)Cs + (unsigned)(c0 + j * 8)) * sizeof(T)) : OOB;
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

// ---- fused depthwise 3x3 + pointwise 1x1 (DepthwiseSeparableConv2d, nn/modules/conv/dsconv.py:106-149) --------------------
// Per 32-channel chunk of a source: the halo tile arrives by LDS-DMA as in mconv_kernel; the 256 threads run the depthwise
// 3x3 on it (thread = 8 channels of a pixel: 9 ds_read_b128 + 72 fp32 FMAs, weights [9][32] + bias staged in LDS) and write
// the 16-bit result as a second LDS tile in MFMA-fragment layout; the pointwise conv is then one MFMA step per pixel fragment.
// The depthwise result never goes to HBM: a Zero-DCE++ layer moves 128 B/px instead of 256.  VALU-bound (the MFMAs are ~5 %).
struct DsParams {
    MConvParams m;
    const float* dw_w;       // [Cin_total][9] fp32 (the checkpoint's (Cin,1,3,3) flattened)
    const float* dw_b;       // [Cin_total] or NULL
};

template <typename T, int NF>
__global__ __launch_bounds__(256, 2) void dsconv_kernel(DsParams dp) {
    const MConvParams& p = dp.m;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using frag = typename ie_elem<T>::frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = blockIdx.x;
    const int per_xcd = p.total_blocks >> 3;
    int lb = (pb & 7) * per_xcd + (pb >> 3);
    const int ct = lb % p.cout_tiles; lb /= p.cout_tiles;
    const int tx = lb % p.tiles_x; lb /= p.tiles_x;
    const int ty = lb % p.tiles_y;
    const int n = lb / p.tiles_y;
    if (n >= p.N) return;
    const int oy0 = ty * TH, ox0 = tx * TW;
    constexpr int TWH = TW + 2, THH = TH + 2, TPX = TWH * THH, ND = (TPX + 15) >> 4;
    unsigned char* in_lds = smem;                                   // [18 x 34 px][64 B], pieces swizzled by (q>>1)&3
    unsigned char* dw_lds = smem + ND * 1024;                       // [16 x 32 px][64 B], same rule on the tile-linear pixel index
    unsigned char* w_lds = dw_lds + TH * TW * 64;                   // pointwise weights of the chunk [16*NF][64 B]
    float* dww = (float*)(w_lds + 16 * NF * 64);                    // depthwise weights [9][32] + bias [32]
    const int kc = lane >> 4, l16 = lane & 15;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    ie_f32x4 acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = ie_f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned w_chunk_bytes = 16u * NF * 64;
    const unsigned char* wpk = (const unsigned char*)p.wpk + (size_t)ct * p.nchunks * w_chunk_bytes;

    int cin_base = 0, prev_src = 0;
    for (int ch = 0; ch < p.nchunks; ++ch) {
        const int s = p.chunk_tab[ch] >> 16, c0 = p.chunk_tab[ch] & 0xffff, Cs = p.cs[s];
        if (s != prev_src) { cin_base += p.cs[prev_src]; prev_src = s; }
        const size_t frame_b = (size_t)p.H * p.W * Cs * sizeof(T);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src[s] + (size_t)n * frame_b), 0, (int)frame_b, 0x00020000);
        __syncthreads();                                            // previous chunk: fragments and weights have been read
        for (int d = wave; d < ND; d += 4) {
            const int q = d * 16 + (lane >> 2);
            const int j = (lane & 3) ^ ((q >> 1) & 3);
            const int r = q / TWH, c = q - r * TWH;
            const int iy = oy0 + r - 1, ix = ox0 + c - 1;
            const bool ok = (c0 + j * 8 < Cs) && (q < TPX) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const unsigned off = ok ? (unsigned)(((unsigned)(iy * p.W + ix) * (unsigned)Cs + (unsigned)(c0 + j * 8)) * sizeof(T)) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(in_lds + d * 1024), 16, (int)off, 0, 0, 0);
        }
        {
            const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(wpk + (size_t)ch * w_chunk_bytes), 0, (int)w_chunk_bytes, 0x00020000);
            for (int d = wave; d < NF; d += 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(w_lds + d * 1024), 16, d * 1024 + lane * 16, 0, 0, 0);
            for (int i = tid; i < 10 * 32; i += 256) {              // depthwise weights of the chunk's channels (zeros beyond the source)
                const int t = i >> 5, k = i & 31;
                const bool live = c0 + k < Cs;
                dww[i] = !live ? 0.f : (t < 9 ? dp.dw_w[(size_t)(cin_base + c0 + k) * 9 + t] : (dp.dw_b ? dp.dw_b[cin_base + c0 + k] : 0.f));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // depthwise: item = (tile pixel, 8-channel piece)
        for (int i = tid; i < TH * TW * 4; i += 256) {
            const int px = i >> 2, j = i & 3;
            const int r = px >> 5, c = px & 31;
            float a8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a8[k] = dww[9 * 32 + j * 8 + k];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int q = (r + ky) * TWH + c + kx;
                    typedef __attribute__((ext_vector_type(8))) T t8;
                    const t8 v = *(const t8*)(in_lds + (size_t)q * 64 + ((j ^ ((q >> 1) & 3)) << 4));
                    const float* wt = dww + (ky * 3 + kx) * 32 + j * 8;
#pragma unroll
                    for (int k = 0; k < 8; ++k) a8[k] = fmaf((float)v[k], wt[k], a8[k]);
                }
            *(ie_u32x4*)(dw_lds + (size_t)px * 64 + ((j ^ ((px >> 1) & 3)) << 4)) =
                ie_u32x4{ie_pack2<T>(a8[0], a8[1]), ie_pack2<T>(a8[2], a8[3]), ie_pack2<T>(a8[4], a8[5]), ie_pack2<T>(a8[6], a8[7])};
        }
        __syncthreads();
        frag a[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) a[f] = *(const frag*)(w_lds + (size_t)(f * 16 + l16) * 64 + ((kc ^ ((l16 >> 1) & 3)) << 4));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = (wave * 4 + (i >> 1)) * TW + (i & 1) * 16 + l16;
            const frag b = *(const frag*)(dw_lds + (size_t)q * 64 + ((kc ^ ((q >> 1) & 3)) << 4));
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[i][f] = ie_elem<T>::mfma16(a[f], b, acc[i][f]);
        }
    }
    switch (p.act) {
    case IE_ACT_RELU: mconv_epilogue<T, NF, IE_ACT_RELU>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_LRELU: mconv_epilogue<T, NF, IE_ACT_LRELU>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_TANH: mconv_epilogue<T, NF, IE_ACT_TANH>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_SIGMOID: mconv_epilogue<T, NF, IE_ACT_SIGMOID>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    default: mconv_epilogue<T, NF, IE_ACT_NONE>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    }
}

// ---- thin inputs (one source of 8 channels: an RGB / RGB+max image carried as 8) ---------------------------------------
// A 32-channel K chunk would be three quarters zeros.  Here K runs over (kx, channel): a K step of 32 = 4 neighbouring pixels
// of an input row x 8 channels = 64 contiguous bytes of the LDS tile [row][col][8 ch] (16 bytes per pixel, no swizzle: a
// fragment's 16 lanes read 256 contiguous bytes).  A kh x kw kernel costs kh * ceil(kw/4) MFMA steps instead of kh * kw
// (9x9: 27 instead of 81; 3x3: 3 instead of 9).  Weights are staged one kernel row (ky) at a time.
template <typename T, int NF>
__global__ __launch_bounds__(256, 2) void mconv_thin_kernel(MConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using frag = typename ie_elem<T>::frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = blockIdx.x;
    const int per_xcd = p.total_blocks >> 3;
    int lb = (pb & 7) * per_xcd + (pb >> 3);
    const int ct = lb % p.cout_tiles; lb /= p.cout_tiles;
    const int tx = lb % p.tiles_x; lb /= p.tiles_x;
    const int ty = lb % p.tiles_y;
    const int n = lb / p.tiles_y;
    if (n >= p.N) return;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int KXG = (p.kw + 3) >> 2;                          // groups of 4 taps along kx
    const int TWH = TW + 4 * KXG - 1, THH = TH + p.kh - 1, TPX = TWH * THH;
    const int ND = (TPX + 63) >> 6;                           // 1-KB DMA instructions (64 pixels of 16 bytes)
    unsigned char* in_lds = smem;
    unsigned char* w_lds = smem + (size_t)ND * 1024;
    const int kc = lane >> 4, l16 = lane & 15;

    ie_f32x4 acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = ie_f32x4{0.f, 0.f, 0.f, 0.f};

    const size_t frame_b = (size_t)p.H * p.W * 8 * sizeof(T);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src[0] + (size_t)n * frame_b), 0, (int)frame_b, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    for (int d = wave; d < ND; d += 4) {                      // whole halo tile: lane = one pixel (16 bytes)
        const int q = d * 64 + lane;
        const int r = q / TWH, c = q - r * TWH;
        int iy = oy0 + r - p.pad, ix = ox0 + c - p.pad;
        bool ok = q < TPX;
        if (p.pad_mode == IE_PAD_REPLICATE) { iy = min(max(iy, 0), p.H - 1); ix = min(max(ix, 0), p.W - 1); }
        else ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const unsigned off = ok ? (unsigned)((iy * p.W + ix) * 8 * (int)sizeof(T)) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(in_lds + d * 1024), 16, (int)off, 0, 0, 0);
    }
    const unsigned w_stage_bytes = (unsigned)KXG * 16 * NF * 64;
    const unsigned char* wpk = (const unsigned char*)p.wpk + (size_t)ct * p.kh * w_stage_bytes;
    for (int ky = 0; ky < p.kh; ++ky) {
        if (ky > 0) __syncthreads();                          // previous row's weights have been read
        const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(wpk + (size_t)ky * w_stage_bytes), 0, (int)w_stage_bytes, 0x00020000);
        for (int d = wave; d < KXG * NF; d += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(w_lds + d * 1024), 16, d * 1024 + lane * 16, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int g = 0; g < KXG; ++g) {
            frag a[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int row = (g * NF + f) * 16 + l16;
                a[f] = *(const frag*)(w_lds + (size_t)row * 64 + ((kc ^ ((l16 >> 1) & 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = (wave * 4 + (i >> 1) + ky) * TWH + (i & 1) * 16 + l16 + g * 4 + kc;   // K quarter kc = pixel +kc
                const frag b = *(const frag*)(in_lds + (size_t)q * 16);
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[i][f] = ie_elem<T>::mfma16(a[f], b, acc[i][f]);
            }
        }
    }
    switch (p.act) {
    case IE_ACT_RELU: mconv_epilogue<T, NF, IE_ACT_RELU>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_LRELU: mconv_epilogue<T, NF, IE_ACT_LRELU>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_TANH: mconv_epilogue<T, NF, IE_ACT_TANH>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    case IE_ACT_SIGMOID: mconv_epilogue<T, NF, IE_ACT_SIGMOID>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    default: mconv_epilogue<T, NF, IE_ACT_NONE>(p, acc, n, ct, oy0, ox0, wave, kc, l16); break;
    }
}

// ---- 3x3, weights in registers, double-buffered input tile ---------------------------------------------------------
// The workgroup's 4 waves are CW cout fragments x RW row groups (CW*RW = 4, CW = 16-cout fragments of the cout tile = NF of
// the packing).  A wave keeps its 9 weight fragments of the current K chunk in registers (loaded straight from the packed
// image in L2, the next chunk's prefetched during the MFMAs), so LDS holds only the input tile -- twice: the LDS-DMA of chunk
// ch+1 runs under the MFMAs of chunk ch, one barrier per chunk, two workgroups per CU.  Every pixel fragment read from LDS
// (input row, column half, kx) feeds up to three MFMAs (ky = 0,1,2).
template <typename T, int ACT, int NT, int TH3>
__device__ __forceinline__ void mconv3_epilogue(const MConvParams& p, ie_f32x4 (&acc)[NT], const float (&bv)[4], int n, int co0, int oy0, int ox0,
                                                int r0, int kc, int l16) {
    const int co = co0 + kc * 4;
    if (co >= p.Cout) return;
    const bool full = ((p.Cout & 3) == 0) && (co + 3 < p.Cout);
    if constexpr (ACT == IE_ACT_RELU || ACT == IE_ACT_NONE || ACT == IE_ACT_LRELU) {
        // interior tile of a plain Conv(+BN)+ReLU / LeakyReLU layer (+ residual add: hinet.py:77, llunetpp.py) -- nearly all of the work: no
        // masks; per accumulator tile 2 packed adds, 2 cvt_pk (+ 2 packed max), one store through a wave-uniform base pointer and a 32-bit lane offset
        if (p.sub == 1 && full && !(p.res && p.shuf_c) && !p.clamp && !p.out_f32 && oy0 + TH3 <= p.Ho && ox0 + TW <= p.Wo) {
            // fused PixelShuffle(2) (edsr.py:88-123): the weight rows were packed as [i][j][c], so this wave's 16-cout fragment is 16
            // consecutive channels c of ONE sub-pixel (i, j): out[2 oy + i][2 ox + j][c]
            const int SC = p.shuf_c ? p.shuf_c : p.Cout, SS = p.shuf_c ? 2 : 1;
            const int ij = p.shuf_c ? co0 / SC : 0, c0 = p.shuf_c ? co0 - ij * SC : co0;
            const char* tile = (const char*)((T*)p.dst + (((size_t)n * SS * p.Hs + SS * (oy0 + r0) + (ij >> 1)) * (SS * p.Ws) + SS * ox0 + (ij & 1)) * SC);
            const unsigned lane_off = (unsigned)((l16 * p.Cout + co) * (int)sizeof(T));
            const unsigned row_b = (unsigned)(SS * SS) * (unsigned)p.Ws * (unsigned)SC * (unsigned)sizeof(T), half_b = 16u * (unsigned)p.Cout * (unsigned)sizeof(T);
            const ie_f32x2 b01 = {bv[0], bv[1]}, b23 = {bv[2], bv[3]};
            // the two column halves of a row are stored by ONE instruction: lanes kc / kc+1 exchange halves (v_permlane16_swap), so
            // even-kc lanes hold couts 4kc..4kc+7 of the left-half pixel and odd-kc lanes couts 4(kc-1)..+7 of the right-half pixel:
            // 16 bytes per lane, half the store instructions (the epilogue is store-ISSUE-bound: 32 -> 16 per wave and tile)
#ifdef IE_MEXP_FULL_LINES   // timing experiment (wrong results, Cout = 64): the same bytes, but every store instruction writes 1 KiB of whole lines
            const unsigned lane_off16 = (unsigned)(((co0 >> 4) & 3) * 1024 + (kc * 16 + l16) * 16);
#else
            const unsigned lane_off16 = (unsigned)(((l16 + 16 * (kc & 1)) * SS * SC + c0 + 4 * (kc & 2)) * (int)sizeof(T));
#endif
            (void)half_b; (void)lane_off;
            // residual: read in the layout of the store (16 bytes = 8 channels of one pixel per lane), swapped back to the accumulator layout
            // (the swap is its own inverse) and added in fp32 behind the activation, before the one rounding
            const char* rtile = p.res ? (const char*)p.res + (tile - (const char*)p.dst) : nullptr;
            ie_u32x4 rnext = {0u, 0u, 0u, 0u};          // one row pair ahead: the accumulators leave little room (128 of 256 VGPRs)
            if (rtile) rnext = *(const ie_u32x4*)(rtile + lane_off16);
            const float slope = p.slope;
            // fused MaxPool2d(2) (ReLU layers: 16-bit values >= +0 order as signed integers): rows 2q, 2q + 1 are this lane's, the column pair is lanes l16, l16 ^ 1
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const bool pool = ACT == IE_ACT_RELU && p.pool != nullptr;
            const unsigned prow_b = (unsigned)(p.Wo >> 1) * (unsigned)p.Cout * (unsigned)sizeof(T);
            const char* ptile = pool ? (const char*)((T*)p.pool + (((size_t)n * (p.Ho >> 1) + ((oy0 + r0) >> 1)) * (p.Wo >> 1) + (ox0 >> 1)) * p.Cout) : nullptr;
            const unsigned plane_off = (unsigned)((((l16 >> 1) + 8 * (kc & 1)) * p.Cout + c0 + 4 * (kc & 2)) * (int)sizeof(T));
            s16x8 prev_row = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < NT; t += 2) {
                ie_u32x2 o[2];
                ie_u32x2 rp[2] = {ie_u32x2{0u, 0u}, ie_u32x2{0u, 0u}};
                if (rtile) {
                    const ie_u32x4 rc = rnext;
                    if (t + 2 < NT) rnext = *(const ie_u32x4*)(rtile + (unsigned)((t >> 1) + 1) * row_b + lane_off16);
                    const auto rx = __builtin_amdgcn_permlane16_swap(rc[0], rc[2], false, false);
                    const auto ry = __builtin_amdgcn_permlane16_swap(rc[1], rc[3], false, false);
                    rp[0] = ie_u32x2{rx[0], ry[0]}; rp[1] = ie_u32x2{rx[1], ry[1]};
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ie_f32x2 v01 = ie_f32x2{acc[t + h][0], acc[t + h][1]} + b01, v23 = ie_f32x2{acc[t + h][2], acc[t + h][3]} + b23;
                    if constexpr (ACT == IE_ACT_LRELU) {
                        v01[0] = v01[0] > 0.f ? v01[0] : v01[0] * slope; v01[1] = v01[1] > 0.f ? v01[1] : v01[1] * slope;
                        v23[0] = v23[0] > 0.f ? v23[0] : v23[0] * slope; v23[1] = v23[1] > 0.f ? v23[1] : v23[1] * slope;
                    }
                    if (rtile) {
                        if constexpr (ACT == IE_ACT_RELU) { v01[0] = fmaxf(v01[0], 0.f); v01[1] = fmaxf(v01[1], 0.f); v23[0] = fmaxf(v23[0], 0.f); v23[1] = fmaxf(v23[1], 0.f); }
                        v01[0] += ie_unpack_lo<T>(rp[h][0]); v01[1] += ie_unpack_hi<T>(rp[h][0]);
                        v23[0] += ie_unpack_lo<T>(rp[h][1]); v23[1] += ie_unpack_hi<T>(rp[h][1]);
                        o[h] = ie_u32x2{ie_pack2<T>(v01[0], v01[1]), ie_pack2<T>(v23[0], v23[1])};
                    } else if constexpr (ACT == IE_ACT_RELU) o[h] = ie_u32x2{ie_pack2_relu<T>(v01[0], v01[1]), ie_pack2_relu<T>(v23[0], v23[1])};
                    else o[h] = ie_u32x2{ie_pack2<T>(v01[0], v01[1]), ie_pack2<T>(v23[0], v23[1])};
                }
                const auto sx = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);
                const ie_u32x4 line = {sx[0], sy[0], sx[1], sy[1]};
                *(ie_u32x4*)(const_cast<char*>(tile) + (unsigned)(t >> 1) * row_b + lane_off16) = line;
                if constexpr (ACT == IE_ACT_RELU) {
                    if (pool) {          // wave-uniform
                        const s16x8 cur_row = __builtin_bit_cast(s16x8, line);
                        if ((t >> 1) & 1) {
                            const ie_u32x4 m = __builtin_bit_cast(ie_u32x4, __builtin_elementwise_max(prev_row, cur_row));
                            ie_u32x4 nb;
#pragma unroll
                            for (int e = 0; e < 4; ++e) nb[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m[e], 0xb1, 0xf, 0xf, true);      // quad_perm [1, 0, 3, 2]
                            const s16x8 mm = __builtin_elementwise_max(__builtin_bit_cast(s16x8, m), __builtin_bit_cast(s16x8, nb));
                            if (!(l16 & 1)) *(ie_u32x4*)(const_cast<char*>(ptile) + (unsigned)(t >> 2) * prow_b + plane_off) = __builtin_bit_cast(ie_u32x4, mm);
                        } else prev_row = cur_row;
                    }
                }
            }
            return;
        }
    }
    if (p.sub == 1 && full && !p.res && !p.clamp && !p.out_f32) {
        // fast path (every hot layer): one 64-bit base per lane, wave-uniform row / half offsets, 8 VALU + 1 store per tile
        const int ox_a = ox0 + l16;
        const bool live_x[2] = {ox_a < p.Wo, ox_a + 16 < p.Wo};
        const int SC = p.shuf_c ? p.shuf_c : p.Cout, SS = p.shuf_c ? 2 : 1;
        const int ij = p.shuf_c ? co / SC : 0, cc = p.shuf_c ? co - ij * SC : co;
        T* dst = (T*)p.dst + ((((size_t)n * SS * p.Hs + SS * (oy0 + r0) + (ij >> 1)) * (SS * p.Ws) + SS * ox_a + (ij & 1)) * SC + cc);
        const float slope = p.slope;
        const unsigned row_stride = (unsigned)(SS * SS) * (unsigned)p.Ws * (unsigned)SC;       // elements
        const unsigned half_stride = 16u * (unsigned)SS * (unsigned)SC;
        const int rows_live = p.Ho - (oy0 + r0);                             // wave-uniform
        const bool pool = ACT == IE_ACT_RELU && p.pool != nullptr && !p.shuf_c;
        if (pool) {          // edge tiles of a layer with the fused MaxPool2d(2): every lane runs the row pair (the column pair is lanes l16, l16 ^ 1), stores are masked
            const int Hp = p.Ho >> 1, Wp = p.Wo >> 1;
            T* pdst = (T*)p.pool + ((((size_t)n * Hp + ((oy0 + r0) >> 1)) * Wp + (ox_a >> 1)) * p.Cout + co);
            const unsigned prow_stride = (unsigned)Wp * (unsigned)p.Cout, phalf_stride = 8u * (unsigned)p.Cout;
            const int prow_live = Hp - ((oy0 + r0) >> 1);
            const bool plive_x[2] = {!(l16 & 1) && (ox_a >> 1) < Wp, !(l16 & 1) && ((ox_a + 16) >> 1) < Wp};
#pragma unroll
            for (int t = 0; t < NT; t += 4) {             // rows 2q, 2q + 1; halves 0, 1
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float m[4];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int tt = t + 2 * rr + h;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = act_fast<ACT>(acc[tt][j] + bv[j], slope);
                        if ((tt >> 1) < rows_live && live_x[h])
                            *(ie_u32x2*)(dst + (unsigned)(tt >> 1) * row_stride + (unsigned)h * half_stride) = ie_u32x2{ie_pack2<T>(v[0], v[1]), ie_pack2<T>(v[2], v[3])};
#pragma unroll
                        for (int j = 0; j < 4; ++j) m[j] = rr ? fmaxf(m[j], v[j]) : v[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        m[j] = fmaxf(m[j], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[j]), 0xb1, 0xf, 0xf, true)));
                    if ((t >> 2) < prow_live && plive_x[h])
                        *(ie_u32x2*)(pdst + (unsigned)(t >> 2) * prow_stride + (unsigned)h * phalf_stride) = ie_u32x2{ie_pack2<T>(m[0], m[1]), ie_pack2<T>(m[2], m[3])};
                }
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if ((t >> 1) >= rows_live || !live_x[t & 1]) continue;
            const unsigned o = (unsigned)(t >> 1) * row_stride + (unsigned)(t & 1) * half_stride;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_fast<ACT>(acc[t][j] + bv[j], slope);
            *(ie_u32x2*)(dst + o) = ie_u32x2{ie_pack2<T>(v[0], v[1]), ie_pack2<T>(v[2], v[3])};
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {             // general path (stride 2, residual, ragged Cout)
        const int oy = oy0 + r0 + (t >> 1), ox = ox0 + (t & 1) * 16 + l16;
        bool live = oy < p.Ho && ox < p.Wo;
        int sy = oy, sx = ox;
        if (p.sub > 1) {
            live = live && !(oy % p.sub) && !(ox % p.sub);
            sy = oy / p.sub; sx = ox / p.sub;
        }
        if (!live) continue;
        size_t o = (((size_t)n * p.Hs + sy) * p.Ws + sx) * p.Cout + co;
        if (p.shuf_c) {
            const int ij = co / p.shuf_c, cc = co - ij * p.shuf_c;
            o = (((size_t)n * 2 * p.Hs + 2 * sy + (ij >> 1)) * (2 * p.Ws) + 2 * sx + (ij & 1)) * p.shuf_c + cc;
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_fast<ACT>(acc[t][j] + bv[j], p.slope);
        if (p.res) {
            const T* rp = (const T*)p.res + o;
            if (full) {
                typedef __attribute__((ext_vector_type(4))) T t4;
                const t4 r4 = *(const t4*)rp;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)r4[j];
    } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (co + j < p.Cout) v[j] += (float)rp[j];
            }
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

template <typename T, int RW, int TH3>
__global__ __launch_bounds__(256, TH3 == 16 ? 2 : 3) void mconv3_kernel(MConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using frag = typename ie_elem<T>::frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int CW = 4 / RW;                 // cout fragments per workgroup (= NF of the weight packing)
    constexpr int R = TH3 / RW;                // output rows per wave
    constexpr int NT = 2 * R;                  // accumulator tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave % CW, rw = wave / CW;
    constexpr int TWH = TW + 2, THH = TH3 + 2, TPX = TWH * THH;
    constexpr int ND = ((TPX + 15) / 16 + 3) / 4 * 4;    // 1-KB DMA instructions per tile, padded so every wave issues ND/4
    constexpr unsigned BUF_B = ND * 1024;
    const int kc = lane >> 4, l16 = lane & 15;
    const int r0 = rw * R;

    // Persistent workgroup: tiles pos, pos + G, pos + 2G, ... of the logical order (cout tile fastest, then x, y, frame).
    // pos is XCD-swizzled: the G/8 workgroups of one XCD work on G/8 consecutive logical tiles at any time, so the cout
    // tiles of a pixel tile and its neighbours share that XCD's L2.
    const int G = gridDim.x;
    const int pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int ntiles = p.tiles_x * p.tiles_y * p.cout_tiles * p.N;
    if (pos >= ntiles) return;
    struct Tile { int ct, n, oy0, ox0; };
    auto decode = [&](int id) __attribute__((always_inline)) {
        Tile t;
        t.ct = id % p.cout_tiles; id /= p.cout_tiles;
        const int tx = id % p.tiles_x; id /= p.tiles_x;
        t.oy0 = (id % p.tiles_y) * TH3; t.ox0 = tx * TW;
        t.n = id / p.tiles_y;
        return t;
    };

    ie_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = ie_f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned w_chunk_bytes = 9u * 16 * CW * 64;
    // this lane's 16 bytes of fragment (tap, cw) inside a chunk's packed image (same swizzled image the LDS path uses)
    const unsigned w_lane_off = (unsigned)((cw * 16 + l16) * 64 + ((kc ^ ((l16 >> 1) & 3)) << 4));
    frag w[9];                                  // w[ky*3+kx]
    auto load_w_kx = [&](const Tile& t, int ch, int kx) __attribute__((always_inline)) {
        const unsigned char* base = (const unsigned char*)p.wpk + ((size_t)t.ct * p.nchunks + ch) * w_chunk_bytes + w_lane_off;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) w[ky * 3 + kx] = *(const frag*)(base + (ky * 3 + kx) * (16 * CW * 64));
    };
    // LDS-DMA addressing: structured buffer (stride = one pixel of the source, index = pixel of the frame), so the per-chunk
    // work of a DMA instruction is nil: vindex (pixel or "out of range" -> zeros) is computed once per tile, the channel
    // piece offset once per kernel, the chunk's first channel goes into the scalar offset.
    constexpr int NDW = ND / 4;                 // DMA instructions per wave per tile
#ifndef IE_MCONV_IL1
#define IE_MCONV_IL1 0
#endif
    // interleave the DMA issue with the MFMAs.  (The 128-accumulator variant RW = 1 had no VGPRs to spare for it until the DMA slot words moved
    // into quads: 232 VGPRs without, see slotq.)
    constexpr bool IL_DMA = RW > 1 || IE_MCONV_IL1;
    // per tile and DMA slot: bits 0-27 = pixel index in the frame (replicate: clamped; all ones = out of range -> zeros),
    // bits 28-29 = the column swizzle (c>>1)&3 of the halo pixel this lane fills.  The four lanes of a quad fill the four 16-byte pieces
    // of ONE pixel, so a slot's word is the same in all four: lane j of the quad keeps the words of slots j, j + 4, j + 8, ... and
    // hands them to its quad with a DPP broadcast when the slot's instruction is issued -- ceil(NDW / 4) registers instead of NDW
    // (ten at TH3 = 16: with 128 accumulators they were the ones spilled, and their reloads sat in front of every chunk's DMA issue).
    constexpr int NDQ = (NDW + 3) / 4;
    unsigned slotq[NDQ];
    auto set_tile_pixels = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NDQ; ++i) {
            const int k = 4 * i + (lane & 3);
            const int q = (wave + 4 * k) * 16 + (lane >> 2);
            const int r = q / TWH, c = q - r * TWH;
            int iy = t.oy0 + r - p.pad, ix = t.ox0 + c - p.pad;
            bool ok = q < TPX && k < NDW;
            if (p.pad_mode == IE_PAD_REPLICATE) { iy = min(max(iy, 0), p.H - 1); ix = min(max(ix, 0), p.W - 1); }
            else ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            slotq[i] = (ok ? (unsigned)(iy * p.W + ix) : 0x0fffffffu) | ((unsigned)((c >> 1) & 3) << 28);
        }
    };
    auto slot_of = [&](int k) __attribute__((always_inline)) {      // k is a compile-time constant at every call site
        const unsigned v = slotq[k >> 2];
        switch (k & 3) {      // quad_perm broadcast of lane (k & 3) of each quad
        case 0: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xf, 0xf, true);
        case 1: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xf, 0xf, true);
        case 2: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xaa, 0xf, 0xf, true);
        default: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xff, 0xf, 0xf, true);
        }
    };
    struct CDesc { const char* base; unsigned info; };
    auto fetch_desc = [&](int ch) __attribute__((always_inline)) { return CDesc{(const char*)p.cd_base[ch], p.cd_info[ch]}; };
    const unsigned frame_px = (unsigned)p.H * (unsigned)p.W;
    // one DMA instruction (slot k of this wave) of the tile described by (frame, d) into buffer bufsel
    auto dma_one = [&](int k, int frame, const CDesc& d, unsigned bufsel) __attribute__((always_inline)) {
        const unsigned cs_b = d.info & 0xffffu, c0_b = d.info >> 16;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(d.base + (size_t)frame * frame_px * cs_b), (short)cs_b, (int)frame_px, 0x00020000);
        unsigned char* buf = smem + bufsel * BUF_B + wave * 1024;
        const int rem = (int)(cs_b - c0_b);                     // bytes of this chunk that exist in the source (>= 64: whole chunk)
        const unsigned sk = slot_of(k);
        const int voff = (int)(((lane & 3) ^ (sk >> 28)) << 4);              // this lane's channel piece of the pixel (column swizzle)
        const int vi = voff < rem ? (int)(sk & 0x0fffffffu) : 0x0fffffff;        // short chunk: missing channel pieces read as zeros
        __builtin_amdgcn_struct_ptr_buffer_load_lds(rs, (lds_ptr)(buf + k * 4096), 16, vi, voff, (int)c0_b, 0, 0);
    };
    auto issue_dma = [&](int frame, const CDesc& d, unsigned bufsel) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NDW; ++k) dma_one(k, frame, d, bufsel);
    };
    // The 16-byte pieces of a pixel are swizzled by its COLUMN, (c>>1)&3 (not by the linear pixel index): a fragment's lane
    // address is then lane_base[kx] + (ir*TWH*64 + hf*1024), a compile-time immediate of the ds_read -- three address
    // registers instead of one per (row, half, kx).  Bank behaviour is that of the linear rule (a per-row phase only).
    unsigned lane_base[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) lane_base[kx] = (unsigned)((r0 * TWH + l16 + kx) * 64 + ((kc ^ (((l16 + kx) >> 1) & 3)) << 4));
    auto compute_kx = [&](unsigned bufsel, int kx, bool dma, int dframe, const CDesc& dd) __attribute__((always_inline)) {
        const unsigned char* pk = smem + bufsel * BUF_B + lane_base[kx];
        // fragment reads run one input row (6 MFMAs) ahead of their use; sched_group_barrier pins "the 2 reads first, then this
        // row's MFMAs" inside every step (a second row of look-ahead changed nothing measurable and costs 8 VGPRs)
        frag cur[2], nxt[2];
        cur[0] = *(const frag*)(pk);
        cur[1] = *(const frag*)(pk + 1024);
#pragma unroll
        for (int ir = 0; ir < R + 2; ++ir) {
            if (ir + 1 < R + 2) {
                nxt[0] = *(const frag*)(pk + (ir + 1) * (TWH * 64));
                nxt[1] = *(const frag*)(pk + (ir + 1) * (TWH * 64) + 1024);
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int orow = ir - ky;
                    if (orow < 0 || orow >= R) continue;
                    acc[orow * 2 + hf] = ie_elem<T>::mfma16(w[ky * 3 + kx], cur[hf], acc[orow * 2 + hf]);
                }
            if (ir + 1 < R + 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the next tile's LDS-DMA instructions ride in the shadow of the MFMAs, one per row step (issuing them in a block
            // costs ~270 cycles each with the matrix pipe idle)
            if (IL_DMA && kx == 0 && dma) {
#pragma unroll
                for (int k = ir; k < NDW; k += R + 2) dma_one(k, dframe, dd, bufsel ^ 1);
            }
            cur[0] = nxt[0]; cur[1] = nxt[1];
        }
    };

    // Flat software pipeline over (tile, chunk): the LDS-DMA of the NEXT (tile, chunk) -- which may be chunk 0 of the next
    // tile -- and its weights are issued under the MFMAs of the current one; one barrier per chunk.  VMEM order per chunk:
    // [DMA of the next input tile: ND/4 per wave] [3 weight loads after each kx].  Inside a tile the 6 youngest loads
    // (weights of kx = 1, 2) may still be in flight at the top (the compiler waits for those registers itself); after an
    // epilogue the stores are in the queue too (loads and stores retire out of order with each other): wait for all.
    int id = pos;
    Tile cur = decode(id);
    unsigned bufsel = 0;
    set_tile_pixels(cur);
    CDesc dn = fetch_desc(0);
    issue_dma(cur.n, dn, 0);
    dn = fetch_desc(p.nchunks > 1 ? 1 : 0);     // descriptor of the NEXT (tile, chunk): its scalar loads never sit on the critical path
    load_w_kx(cur, 0, 0); load_w_kx(cur, 0, 1); load_w_kx(cur, 0, 2);
    bool fresh = true;                          // first chunk of the kernel or after an epilogue
    while (true) {
        const int nid = id + G;
        const bool has_next_tile = nid < ntiles;
        Tile nxt_tile = cur;
        if (has_next_tile) nxt_tile = decode(nid);
        float bv[4];                            // this lane's 4 biases of the tile: loaded now, used in the epilogue
        {
            const int co = (cur.ct * CW + cw) * 16 + kc * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = (p.bias && co + j < p.Cout) ? p.bias[co + j] : 0.f;
        }
        for (int ch = 0; ch < p.nchunks; ++ch) {
            const int sb = (id == pos && ch < 6) ? ch * 8 : 56;      // stamp slot (diagnostic build)
            (void)sb;
            MST(sb + 0);
            if (fresh) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // this wave's part of the tile (and the kx = 0 weights) has landed
            fresh = false;
            MST(sb + 1);
            __syncthreads();                                    // ... everyone's; and everyone is done with the other buffer
            MST(sb + 2);
            const bool last = ch + 1 == p.nchunks;
            const bool more = !last || has_next_tile;
            const Tile& nt = last ? nxt_tile : cur;
            const int nch = last ? 0 : ch + 1;
            if (last && has_next_tile) set_tile_pixels(nxt_tile);
            if (!IL_DMA && more) issue_dma(nt.n, dn, bufsel ^ 1);
            MST(sb + 3);
            compute_kx(bufsel, 0, more, nt.n, dn);
            {
                int c2 = ch + 2;
                if (c2 >= p.nchunks) c2 -= p.nchunks;
                if (c2 >= p.nchunks) c2 = 0;
                dn = fetch_desc(c2);
            }
            MST(sb + 4);
            if (more) load_w_kx(nt, nch, 0);
            compute_kx(bufsel, 1, false, 0, dn);
            MST(sb + 5);
            if (more) load_w_kx(nt, nch, 1);
            compute_kx(bufsel, 2, false, 0, dn);
            MST(sb + 6);
            if (more) load_w_kx(nt, nch, 2);
            bufsel ^= 1;
        }
        const int co0 = (cur.ct * CW + cw) * 16;
        switch (p.act) {
        case IE_ACT_RELU: mconv3_epilogue<T, IE_ACT_RELU, NT, TH3>(p, acc, bv, cur.n, co0, cur.oy0, cur.ox0, r0, kc, l16); break;
        case IE_ACT_LRELU: mconv3_epilogue<T, IE_ACT_LRELU, NT, TH3>(p, acc, bv, cur.n, co0, cur.oy0, cur.ox0, r0, kc, l16); break;
        case IE_ACT_TANH: mconv3_epilogue<T, IE_ACT_TANH, NT, TH3>(p, acc, bv, cur.n, co0, cur.oy0, cur.ox0, r0, kc, l16); break;
        case IE_ACT_SIGMOID: mconv3_epilogue<T, IE_ACT_SIGMOID, NT, TH3>(p, acc, bv, cur.n, co0, cur.oy0, cur.ox0, r0, kc, l16); break;
        default: mconv3_epilogue<T, IE_ACT_NONE, NT, TH3>(p, acc, bv, cur.n, co0, cur.oy0, cur.ox0, r0, kc, l16); break;
        }
        if (id == pos) MST(48);
        if (!has_next_tile) break;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = ie_f32x4{0.f, 0.f, 0.f, 0.f};
        cur = nxt_tile; id = nid; fresh = true;
    }
}


// ---- device-side weight packing (training: the fp32 master weights change every step) ---------------------------------------
// One thread per 16-bit element of the packed image (layout of ie_mconv_pack_weights, non-thin).  The source is the trainer's
// HWIO fp32 matrix W[tap][ci][co]; mode 1 packs the weights of the DATA-GRADIENT conv of source slice [cb, cb+Cs): the conv of dY
// (K runs over the original couts) with W'[tap][co][ci] = W[taps-1-tap][cb+ci][co] (output channels = the slice's cins).
struct PackDevParams {
    const float* w; void* dst;
    int nf, cout_tiles, nchunks, nstages, taps_stage, taps, Cout_p, dtype;
    int s_tap, s_k, s_o, off, flip;          // element strides of W for (tap, K index, output index), base offset, tap flip
    int chunk_k0[MAX_CHUNKS];                 // first K index of the chunk
    int chunk_valid[MAX_CHUNKS];              // K indices that exist in the chunk (<= 32)
    size_t total;
};
__global__ void mconv_pack_dev_kernel(PackDevParams q) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.total) return;
    const int rows = 16 * q.nf;
    size_t r = e;
    const int kk = (int)(r & 31); r >>= 5;
    const int row = (int)(r % rows); r /= rows;
    const int t = (int)(r % q.taps_stage); r /= q.taps_stage;
    const int st = (int)(r % q.nstages); r /= q.nstages;
    const int ch = (int)(r % q.nchunks);
    const int ct = (int)(r / q.nchunks);
    const int sw = ((row & 15) >> 1) & 3;
    const int k = (((kk >> 3) ^ sw) << 3) + (kk & 7);
    const int tap = st * q.taps_stage + t, co = ct * rows + row;
    float v = 0.f;
    if (tap < q.taps && co < q.Cout_p && k < q.chunk_valid[ch]) {
        const int tp = q.flip ? q.taps - 1 - tap : tap;
        v = q.w[(size_t)tp * q.s_tap + (size_t)(q.chunk_k0[ch] + k) * q.s_k + (size_t)co * q.s_o + q.off];
    }
    if (q.dtype == IE_F16) ((_Float16*)q.dst)[e] = (_Float16)v;
    else ((__bf16*)q.dst)[e] = (__bf16)v;
}

int pick_nf(int Cout) { return Cout > 32 ? 4 : (Cout > 16 ? 2 : 1); }

struct Plan {
    int nf, cout_tiles, nchunks, taps, taps_stage, nstages;
    int thin, kxg;               // thin: one 8-channel source, K = (4 kx taps x 8 channels) per step, one stage per kernel row
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
    static const int no_thin = getenv("IE_MCONV_NO_THIN") ? atoi(getenv("IE_MCONV_NO_THIN")) : 0;     // A/B switch (packing + kernel)
    pl->thin = (nsrc == 1 && cs[0] == 8 && !no_thin) ? 1 : 0;
    pl->kxg = (kw + 3) / 4;
    if (pl->thin) {
        pl->taps_stage = pl->kxg; pl->nstages = kh;
        pl->stage_bytes = (size_t)pl->kxg * 16 * pl->nf * 64;
        pl->total_bytes = (size_t)pl->cout_tiles * kh * pl->stage_bytes;
        return IE_OK;
    }
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

#ifdef IE_MSTAMP
static unsigned long long* g_mconv_stamps = nullptr;
IE_EXPORT void ie_debug_mconv_stamps(void* dev) { g_mconv_stamps = (unsigned long long*)dev; }
#endif

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
    if (pl.thin) {                                           // [ct][ky][kx group][row][k = kxo*8 + ci], rows swizzled like the LDS image
        for (int ct = 0; ct < pl.cout_tiles; ++ct)
            for (int ky = 0; ky < kh; ++ky)
                for (int g = 0; g < pl.kxg; ++g)
                    for (int r = 0; r < rows; ++r) {
                        const int co = ct * rows + r;
                        if (co >= Cout) continue;
                        unsigned short* row = host.data() + ((((size_t)ct * kh + ky) * pl.kxg + g) * rows + r) * 32;
                        const int sw = ((r & 15) >> 1) & 3;
                        for (int k = 0; k < 32; ++k) {
                            const int kx = g * 4 + (k >> 3), ci = k & 7;
                            if (kx >= kw || ci >= Cin) continue;
                            row[(((k >> 3) ^ sw) << 3) + (k & 7)] = f32_to_16(w_oihw_host[(((size_t)co * Cin + ci) * kh + ky) * kw + kx], dtype);
                        }
                    }
        IE_HIP_CHECK(hipMemcpyAsync(dst_dev, host.data(), pl.total_bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        IE_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        return IE_OK;
    }
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

static int mconv_impl(ie_mconv_desc* d, const void* const* srcs, const int* channels, const void* packed_w, const float* bias, const void* residual,
                      void* dst, void* pool_dst, bool* pooled, void* stream);

IE_EXPORT int ie_mconv_nhwc(ie_mconv_desc* d, const void* const* srcs, const int* channels, const void* packed_w, const float* bias,
                            const void* residual, void* dst, void* stream) {
    return mconv_impl(d, srcs, channels, packed_w, bias, residual, dst, nullptr, nullptr, stream);
}

// Conv (+ bias + ReLU) and the nn.MaxPool2d(2) behind it (unet.py:70-72 `Down` = MaxPool2d(2) -> DoubleConv): the strip kernel writes the pooled
// copy from the rows it has parked in LDS; every other shape runs the conv and then the streaming pool kernel.
IE_EXPORT int ie_mconv_pool_nhwc(ie_mconv_desc* d, const void* const* srcs, const int* channels, const void* packed_w, const float* bias, void* dst,
                                 void* pool_dst, void* stream) {
    if (!pool_dst) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pool_nhwc: null pointer");
    if (d && (d->pixel_shuffle || d->out_dtype != d->in_dtype || d->stride != 1)) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_pool_nhwc: stride 1, same-type output, no pixel shuffle");
    bool pooled = false;
    int rc = mconv_impl(d, srcs, channels, packed_w, bias, nullptr, dst, pool_dst, &pooled, stream);
    if (rc != IE_OK || pooled) return rc;
    return ie_maxpool2x2_nhwc(dst, pool_dst, d->in_dtype, d->N, d->Ho, d->Wo, d->Cout, stream);
}

// Conv3x3 (+ bias + ReLU) -> 1x1 conv (+ bias, optional clamp) -> NCHW fp32, one launch: the U-Net head (unet.py:186-195 `outc` behind the last
// DoubleConv).  The 3x3 conv's rows never reach HBM; the 1x1 contracts them on the matrix cores while they are parked in LDS.
static bool tail_ok(const ie_mconv_desc* d, const int* channels, int tail_cout) {
    static const int on = getenv("IE_SCONV_TAIL") ? atoi(getenv("IE_SCONV_TAIL")) : 1;           // A/B: 0 = the two separate convs
    return on && d && channels && !d->pixel_shuffle && d->stride == 1 && d->act == IE_ACT_RELU && (d->W & 3) == 0 && tail_cout >= 1 && tail_cout <= 16 && !d->clamp &&
           d->out_dtype == d->in_dtype && ie_sconv3_fused_ok(d->nsrc, channels) &&
           ie_sconv3_applicable(d->nsrc, channels, d->Cout, d->kh, d->kw, d->stride, d->pad, d->pad_mode, d->act, false, false, false, d->W) &&
           (double)tail_cout * d->H * d->W * 4.0 < 2147483647.0;
}

IE_EXPORT int ie_mconv_tail_applicable(const ie_mconv_desc* d, const int* channels, int tail_cout) { return tail_ok(d, channels, tail_cout) ? 1 : 0; }

IE_EXPORT int ie_mconv_tail_nhwc(ie_mconv_desc* d, const void* const* srcs, const int* channels, const void* packed_w, const float* bias,
                                 const void* tail_w, const float* tail_bias, int tail_cout, int clamp, float clamp_lo, float clamp_hi, float* dst_nchw,
                                 void* stream) {
    if (!d || !srcs || !channels || !packed_w || !tail_w || !tail_bias || !dst_nchw) IE_FAIL(IE_ERR_INVALID, "ie_mconv_tail_nhwc: null pointer");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0) IE_FAIL(IE_ERR_INVALID, "ie_mconv_tail_nhwc: bad shape");
    if ((d->in_dtype != IE_F16 && d->in_dtype != IE_BF16)) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_tail_nhwc: in_dtype must be f16/bf16");
    if (!tail_ok(d, channels, tail_cout))
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_tail_nhwc: needs a strip-kernel layer (3x3, stride 1, pad 1, Cout 32/64, ReLU) with W %% 4 == 0 -- ask "
                                    "ie_mconv_tail_applicable first");
    for (int s = 0; s < d->nsrc; ++s)
        if (!srcs[s]) IE_FAIL(IE_ERR_INVALID, "ie_mconv_tail_nhwc: source %d is null", s);
    d->Ho = d->H; d->Wo = d->W;
    ie_sconv_tail t;
    t.w = tail_w; t.b = tail_bias; t.dst = dst_nchw; t.cout = tail_cout; t.clamp = clamp; t.lo = clamp_lo; t.hi = clamp_hi;
    return ie_sconv3_launch(srcs, channels, d->nsrc, packed_w, bias, nullptr, d->in_dtype, d->N, d->H, d->W, d->Cout, d->act, d->slope, d->pad_mode,
                            d->src_h, d->src_w, nullptr, &t, 1, (hipStream_t)stream);
}

// The first DoubleConv of a U-Net + the MaxPool2d(2) behind it in one launch (unet.py:43-48, 70-72, 152): conv1 = the network's first conv on the
// 8-channel image (its rows are computed inside the kernel, one row ahead of the second conv, and never reach HBM), conv2 = C -> C.
IE_EXPORT int ie_mconv_first_applicable(int C, int H, int W) {
    static const int on = getenv("IE_SCONV_FIRST") ? atoi(getenv("IE_SCONV_FIRST")) : 1;          // A/B: 0 = thin conv + strip conv
    return (on && (C == 32 || C == 64) && H >= 2 && W >= 64 && (double)H * W * 256.0 < 2147483647.0) ? 1 : 0;
}

IE_EXPORT int ie_mconv_first_pool_nhwc(int dtype, int N, int H, int W, int C, const void* img8, const void* packed_w1, const float* bias1,
                                       const void* packed_w2, const float* bias2, void* dst, void* pool_dst, void* stream) {
    if (!img8 || !packed_w1 || !packed_w2 || !dst || !pool_dst || N <= 0) IE_FAIL(IE_ERR_INVALID, "ie_mconv_first_pool_nhwc: null pointer / bad batch");
    if (dtype != IE_F16 && dtype != IE_BF16) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_first_pool_nhwc: f16 / bf16");
    if (!ie_mconv_first_applicable(C, H, W)) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_first_pool_nhwc: C 32 / 64, H >= 2, W >= 64 -- ask ie_mconv_first_applicable first");
    return ie_sconv3_first_launch(img8, packed_w1, bias1, packed_w2, bias2, dst, pool_dst, dtype, N, H, W, C, (hipStream_t)stream);
}

static int mconv_impl(ie_mconv_desc* d, const void* const* srcs, const int* channels, const void* packed_w, const float* bias, const void* residual,
                      void* dst, void* pool_dst, bool* pooled, void* stream) {
    if (!d || !srcs || !channels || !packed_w || !dst) IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: null pointer");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->pad < 0 || (d->stride != 1 && d->stride != 2))
        IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: bad shape (stride must be 1 or 2)");
    if ((d->in_dtype != IE_F16 && d->in_dtype != IE_BF16) || (d->out_dtype != d->in_dtype && d->out_dtype != IE_F32))
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: in_dtype must be f16/bf16 and out_dtype the same or f32");
    Plan pl;
    int rc = make_plan(channels, d->nsrc, d->Cout, d->kh, d->kw, &pl, "ie_mconv_nhwc");
    if (rc != IE_OK) return rc;
    if (!d->pixel_shuffle && (double)d->H * d->W * 256.0 < 2147483647.0 &&        // the strip kernel addresses a frame with 31-bit buffer offsets
        ie_sconv3_applicable(d->nsrc, channels, d->Cout, d->kh, d->kw, d->stride, d->pad, d->pad_mode, d->act, residual != nullptr, d->clamp != 0,
                             d->out_dtype == IE_F32, d->W)) {
        // narrow 3x3 layers: the strip kernel (weights in registers for the whole launch, an output row stored every step)
        for (int s = 0; s < d->nsrc; ++s)
            if (!srcs[s]) IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: source %d is null", s);
        d->Ho = (d->H + d->stride - 1) / d->stride; d->Wo = (d->W + d->stride - 1) / d->stride;       // k 3, pad 1: ceil(H / stride)
        static const int fuse = getenv("IE_SCONV_POOL") ? atoi(getenv("IE_SCONV_POOL")) : 1;      // A/B: 0 = the separate pool kernel
        void* pd = (pool_dst && fuse && d->act == IE_ACT_RELU && d->H >= 2 && d->W >= 2 && d->stride == 1 && ie_sconv3_fused_ok(d->nsrc, channels)) ? pool_dst : nullptr;
        if (pooled) *pooled = pd != nullptr;
        return ie_sconv3_launch(srcs, channels, d->nsrc, packed_w, bias, dst, d->in_dtype, d->N, d->H, d->W, d->Cout, d->act, d->slope, d->pad_mode,
                                d->src_h, d->src_w, pd, nullptr, d->stride, (hipStream_t)stream);
    }
    if (pooled) *pooled = false;
    if (!pool_dst && (double)d->H * d->W * 256.0 < 2147483647.0 &&
        ie_sconv3_thin3_applicable(d->nsrc, channels, d->Cout, d->kh, d->kw, d->stride, d->pad, d->pad_mode, d->act, residual != nullptr, d->clamp != 0,
                                   d->out_dtype == IE_F32, d->pixel_shuffle != 0, d->W)) {
        // thin output (1 .. 4 channels, fp32) over 64-channel sources: the HBM-bound strip form with one wave pair per source
        for (int s = 0; s < d->nsrc; ++s)
            if (!srcs[s]) IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: source %d is null", s);
        d->Ho = d->H; d->Wo = d->W;
        if (pooled) *pooled = false;
        return ie_sconv3_thin3_launch(srcs, d->nsrc, d->src_h, d->src_w, packed_w, bias, (float*)dst, d->in_dtype, d->N, d->H, d->W, d->Cout, d->act, d->pad_mode,
                                      (hipStream_t)stream);
    }
    {
        bool up = false;
        for (int s = 0; s < d->nsrc; ++s) up = up || (d->src_h[s] > 0 && d->src_h[s] != d->H) || (d->src_w[s] > 0 && d->src_w[s] != d->W);
        if (!pl.thin && ie_conv1x1_applicable(d->nsrc, channels, d->Cout, d->kh, d->kw, d->stride, d->pad, d->act, residual != nullptr, d->clamp != 0,
                                              d->out_dtype == IE_F32, d->pixel_shuffle != 0, up)) {
            // 1x1 layers: the streaming GEMM over 128-pixel rows (ie_conv1x1.hip)
            d->Ho = d->H; d->Wo = d->W;
            return ie_conv1x1_launch(srcs, channels, d->nsrc, packed_w, bias, dst, d->in_dtype, (long long)d->N * d->H * d->W, d->Cout, pl.nf, pl.cout_tiles, d->act,
                                     d->slope, (hipStream_t)stream);
        }
    }
    MConvParams p;
    memset(&p, 0, sizeof(p));
    bool any_up = false;
    for (int s = 0; s < d->nsrc; ++s) {
        if (!srcs[s]) IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: source %d is null", s);
        p.src[s] = srcs[s]; p.cs[s] = channels[s];
        p.sh[s] = d->src_h[s] > 0 ? d->src_h[s] : d->H; p.sw[s] = d->src_w[s] > 0 ? d->src_w[s] : d->W;
        any_up = any_up || p.sh[s] != d->H || p.sw[s] != d->W;
    }
    if (any_up && pl.thin) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: an up-sampled source cannot be the 8-channel image source");
#ifdef IE_MSTAMP
    p.stamps = g_mconv_stamps;
#endif
    for (int c = 0; c < pl.nchunks; ++c) {
        p.chunk_tab[c] = ((int)pl.chunk_src[c] << 16) | (int)pl.chunk_c0[c];
        p.cd_base[c] = srcs[pl.chunk_src[c]];
        p.cd_info[c] = (unsigned)(channels[pl.chunk_src[c]] * 2) | ((unsigned)(pl.chunk_c0[c] * 2) << 16);
    }
    p.nchunks = pl.nchunks; p.wpk = packed_w; p.bias = bias; p.res = residual; p.dst = dst;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.kh = d->kh; p.kw = d->kw; p.pad = d->pad; p.pad_mode = d->pad_mode;
    p.sub = d->stride; p.act = d->act; p.clamp = d->clamp; p.out_f32 = d->out_dtype == IE_F32; p.slope = d->slope; p.lo = d->clamp_lo;
    p.hi = d->clamp_hi;
    p.Ho = d->H + 2 * d->pad - d->kh + 1; p.Wo = d->W + 2 * d->pad - d->kw + 1;
    if (p.Ho <= 0 || p.Wo <= 0) IE_FAIL(IE_ERR_INVALID, "ie_mconv_nhwc: empty output");
    p.Hs = (d->H + 2 * d->pad - d->kh) / d->stride + 1; p.Ws = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    d->Ho = p.Hs; d->Wo = p.Ws;
    if (d->pixel_shuffle) {
        if (d->pixel_shuffle != 2 || d->kh != 3 || d->kw != 3 || d->stride != 1 || (d->Cout & 63) || pl.thin || d->out_dtype == IE_F32 || residual)
            IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: fused PixelShuffle needs r = 2, a 3x3 stride-1 conv, Cout a multiple of 64, 16-bit output, no residual");
        p.shuf_c = d->Cout / 4;
        d->Ho = 2 * p.Hs; d->Wo = 2 * p.Ws;
    }
    p.tiles_x = ie_cdiv(p.Wo, TW); p.tiles_y = ie_cdiv(p.Ho, TH); p.cout_tiles = pl.cout_tiles;
    const long logical = (long)p.tiles_x * p.tiles_y * p.cout_tiles * d->N;
    p.total_blocks = (int)((logical + 7) / 8 * 8);
    p.taps_stage = pl.taps_stage; p.nstages = pl.nstages;
    const size_t lds = (size_t)(((TH + d->kh - 1) * (TW + d->kw - 1) + 15) / 16) * 1024 + pl.stage_bytes;
    for (int s = 0; s < d->nsrc; ++s)
        if ((double)d->H * d->W * channels[s] * 2.0 >= 4294967040.0 || (double)d->H * d->W >= 268435455.0)
            IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: a frame of source %d exceeds 4 GiB / 2^28 pixels", s);
    if (lds > 160 * 1024) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_nhwc: %dx%d kernel needs %zu bytes of LDS", d->kh, d->kw, lds);
    void (*fn)(MConvParams) = nullptr;
    const bool h = d->in_dtype == IE_F16;
    const bool k3 = d->kh == 3 && d->kw == 3;
    static const int use_v1_env = getenv("IE_MCONV_V1") ? atoi(getenv("IE_MCONV_V1")) : 0;     // A/B switch: single-buffered LDS-weights kernel
    const int use_v1 = use_v1_env || any_up;      // nearest-up-sampled sources: the strip kernel (above) or this one fold them into their addressing
    size_t lds_bytes = lds;
    int th3 = TH, wgs_per_cu = 2;
    if (pl.thin) {
        switch (pl.nf) {
        case 1: fn = h ? mconv_thin_kernel<_Float16, 1> : mconv_thin_kernel<__bf16, 1>; break;
        case 2: fn = h ? mconv_thin_kernel<_Float16, 2> : mconv_thin_kernel<__bf16, 2>; break;
        default: fn = h ? mconv_thin_kernel<_Float16, 4> : mconv_thin_kernel<__bf16, 4>; break;
        }
        const int twh = TW + 4 * pl.kxg - 1, thh = TH + d->kh - 1;
        lds_bytes = (size_t)((twh * thh + 63) / 64) * 1024 + pl.stage_bytes;
    } else if (k3 && !use_v1) {
        // 64-cout tiles (one cout fragment per wave, 16-row tile = 128 accumulator VGPRs) leave room for only two workgroups per
        // CU; an 8-row tile (64 accumulator VGPRs, 2 x 22 KB of LDS) runs three and overlaps the per-tile serial work better
        // Which is faster depends on K and on the rows the last tile row wastes (profiles/r03/mconv3_tile_height_ab.log, U-Net-32's layers at batch 32): with <= 4 K chunks
        // the per-tile serial part (epilogue, drain, first fill) dominates and three small workgroups hide it better (8-row tiles 6 % faster); at 8 chunks 8-row tiles cost
        // ~3 %, at 16 chunks ~6.5 % per row -- but 135 or 67 rows pad to 144 / 80 with 16-row tiles and to 136 / 72 with 8-row ones.  IE_MCONV_TH = 8 | 16 forces one.
        static const int th_nf4 = getenv("IE_MCONV_TH") ? atoi(getenv("IE_MCONV_TH")) : 0;
        if (pl.nf == 4) {
            bool th8 = th_nf4 == 8;
            if (th_nf4 != 8 && th_nf4 != 16) {
                const double per_row8 = pl.nchunks <= 4 ? 0.95 : (pl.nchunks <= 8 ? 1.03 : 1.065);
                th8 = per_row8 * (double)(ie_cdiv(p.Ho, 8) * 8) < (double)(ie_cdiv(p.Ho, 16) * 16);
            }
            if (th8) { th3 = 8; wgs_per_cu = 3; }
        }
        switch (pl.nf) {
        case 1: fn = h ? mconv3_kernel<_Float16, 4, 16> : mconv3_kernel<__bf16, 4, 16>; break;
        case 2: fn = h ? mconv3_kernel<_Float16, 2, 16> : mconv3_kernel<__bf16, 2, 16>; break;
        default:
            if (th3 == 8) fn = h ? mconv3_kernel<_Float16, 1, 8> : mconv3_kernel<__bf16, 1, 8>;
            else fn = h ? mconv3_kernel<_Float16, 1, 16> : mconv3_kernel<__bf16, 1, 16>;
            break;
        }
        p.tiles_y = ie_cdiv(p.Ho, th3);
        lds_bytes = 2 * (size_t)((((th3 + 2) * (TW + 2) + 15) / 16 + 3) / 4 * 4) * 1024;
    } else {
        switch (pl.nf * 2 + (k3 ? 1 : 0)) {
        case 2: fn = h ? mconv_kernel<_Float16, 1, false> : mconv_kernel<__bf16, 1, false>; break;
        case 3: fn = h ? mconv_kernel<_Float16, 1, true> : mconv_kernel<__bf16, 1, true>; break;
        case 4: fn = h ? mconv_kernel<_Float16, 2, false> : mconv_kernel<__bf16, 2, false>; break;
        case 5: fn = h ? mconv_kernel<_Float16, 2, true> : mconv_kernel<__bf16, 2, true>; break;
        case 8: fn = h ? mconv_kernel<_Float16, 4, false> : mconv_kernel<__bf16, 4, false>; break;
        default: fn = h ? mconv_kernel<_Float16, 4, true> : mconv_kernel<__bf16, 4, true>; break;
        }
    }
    if (lds_bytes > 64 * 1024) {       // opt in to > 64 KB of dynamic LDS once per kernel
        IE_GRANT_LDS(fn, 160 * 1024);
    }
    if (k3 && !use_v1 && !pl.thin && pool_dst) {
        // the 3x3 register-weights kernel pools in its epilogue (ReLU layer, whole 16-cout fragments, 16-bit output, no residual / clamp / shuffle)
        static const int fuse3 = getenv("IE_MCONV_POOL") ? atoi(getenv("IE_MCONV_POOL")) : 1;      // A/B: 0 = the separate pool kernel
        if (fuse3 && d->act == IE_ACT_RELU && d->stride == 1 && !(d->Cout & 15) && !residual && !d->clamp && d->out_dtype != IE_F32 && !d->pixel_shuffle &&
            p.Ho >= 2 && p.Wo >= 2) {
            p.pool = pool_dst;
            if (pooled) *pooled = true;
        }
    }
    int grid = p.total_blocks;
    if (k3 && !use_v1 && !pl.thin) {     // persistent: two workgroups per CU (LDS 2 x 80 KB), each walks its share of the tiles
        static const int wg_cap_env = getenv("IE_MCONV_WGS") ? atoi(getenv("IE_MCONV_WGS")) : 0;
        const int wg_cap = wg_cap_env ? wg_cap_env : 256 * wgs_per_cu;
        const long logical3 = (long)p.tiles_x * p.tiles_y * p.cout_tiles * d->N;
        grid = (int)((logical3 + 7) / 8 * 8);
        if (grid > wg_cap) grid = wg_cap;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    IE_LAUNCH_CHECK();
    return IE_OK;
}


IE_EXPORT int ie_dsconv_nhwc(ie_mconv_desc* d, const void* const* srcs, const int* channels, const float* dw_w_dev, const float* dw_b_dev,
                             const void* packed_pw, const float* pw_bias, void* dst, void* stream) {
    if (!d || !srcs || !channels || !dw_w_dev || !packed_pw || !dst) IE_FAIL(IE_ERR_INVALID, "ie_dsconv_nhwc: null pointer");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != IE_PAD_ZEROS)
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_dsconv_nhwc: depthwise 3x3, stride 1, zero padding 1 only");
    if ((d->in_dtype != IE_F16 && d->in_dtype != IE_BF16) || (d->out_dtype != d->in_dtype && d->out_dtype != IE_F32))
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_dsconv_nhwc: in_dtype must be f16/bf16 and out_dtype the same or f32");
    Plan pl;
    int rc = make_plan(channels, d->nsrc, d->Cout, 1, 1, &pl, "ie_dsconv_nhwc");
    if (rc != IE_OK) return rc;
    if (pl.thin) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_dsconv_nhwc: a single 8-channel source takes the unfused path (ie_dwconv3x3_nhwc + ie_mconv_nhwc)");
    DsParams dp;
    memset(&dp, 0, sizeof(dp));
    MConvParams& p = dp.m;
    for (int s = 0; s < d->nsrc; ++s) {
        if (!srcs[s]) IE_FAIL(IE_ERR_INVALID, "ie_dsconv_nhwc: source %d is null", s);
        if ((double)d->H * d->W * channels[s] * 2.0 >= 4294967040.0) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_dsconv_nhwc: a frame of source %d exceeds 4 GiB", s);
        p.src[s] = srcs[s]; p.cs[s] = channels[s];
    }
    for (int c = 0; c < pl.nchunks; ++c) p.chunk_tab[c] = ((int)pl.chunk_src[c] << 16) | (int)pl.chunk_c0[c];
    p.nchunks = pl.nchunks; p.wpk = packed_pw; p.bias = pw_bias; p.res = nullptr; p.dst = dst;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.kh = 1; p.kw = 1; p.pad = 0; p.pad_mode = IE_PAD_ZEROS;
    p.sub = 1; p.act = d->act; p.clamp = d->clamp; p.out_f32 = d->out_dtype == IE_F32; p.slope = d->slope; p.lo = d->clamp_lo; p.hi = d->clamp_hi;
    p.Ho = p.Hs = d->H; p.Wo = p.Ws = d->W;
    d->Ho = d->H; d->Wo = d->W;
    p.tiles_x = ie_cdiv(p.Wo, TW); p.tiles_y = ie_cdiv(p.Ho, TH); p.cout_tiles = pl.cout_tiles;
    const long logical = (long)p.tiles_x * p.tiles_y * p.cout_tiles * d->N;
    p.total_blocks = (int)((logical + 7) / 8 * 8);
    p.taps_stage = 1; p.nstages = 1;
    dp.dw_w = dw_w_dev; dp.dw_b = dw_b_dev;
    const size_t lds = (size_t)((((TH + 2) * (TW + 2)) + 15) / 16) * 1024 + (size_t)TH * TW * 64 + 16 * pl.nf * 64 + 10 * 32 * sizeof(float);
    void (*fn)(DsParams) = nullptr;
    const bool h = d->in_dtype == IE_F16;
    switch (pl.nf) {
    case 1: fn = h ? dsconv_kernel<_Float16, 1> : dsconv_kernel<__bf16, 1>; break;
    case 2: fn = h ? dsconv_kernel<_Float16, 2> : dsconv_kernel<__bf16, 2>; break;
    default: fn = h ? dsconv_kernel<_Float16, 4> : dsconv_kernel<__bf16, 4>; break;
    }
    IE_GRANT_LDS(fn, 160 * 1024);
    hipLaunchKernelGGL(fn, dim3(p.total_blocks), dim3(256), lds, (hipStream_t)stream, dp);
    IE_LAUNCH_CHECK();
    return IE_OK;
}


IE_EXPORT int ie_mconv_pack_weights_dev(const float* w_hwio_dev, const int* channels, int nsrc, int Cout, int kh, int kw, int dtype, int mode,
                                        int src_index, void* dst_dev, void* stream) {
    if (!w_hwio_dev || !channels || !dst_dev || (dtype != IE_F16 && dtype != IE_BF16) || (mode != 0 && mode != 1) || nsrc < 1 || nsrc > IE_CONV_MAX_SRC)
        IE_FAIL(IE_ERR_INVALID, "ie_mconv_pack_weights_dev: bad args");
    int Cin = 0, cbase[IE_CONV_MAX_SRC];
    for (int s = 0; s < nsrc; ++s) { cbase[s] = Cin; Cin += channels[s]; }
    Plan pl;
    PackDevParams q;
    memset(&q, 0, sizeof(q));
    int rc;
    if (mode == 0) {
        rc = make_plan(channels, nsrc, Cout, kh, kw, &pl, "ie_mconv_pack_weights_dev");
        if (rc != IE_OK) return rc;
        q.Cout_p = Cout; q.s_tap = Cin * Cout; q.s_k = Cout; q.s_o = 1; q.off = 0; q.flip = 0;
        for (int c = 0; c < pl.nchunks; ++c) {
            const int s = pl.chunk_src[c], c0 = pl.chunk_c0[c];
            q.chunk_k0[c] = cbase[s] + c0;
            q.chunk_valid[c] = channels[s] - c0 < KC ? channels[s] - c0 : KC;
        }
    } else {
        if (src_index < 0 || src_index >= nsrc) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pack_weights_dev: bad source index");
        const int one[1] = {Cout};                       // the gradient conv reads dY: one source of Cout channels
        rc = make_plan(one, 1, channels[src_index], kh, kw, &pl, "ie_mconv_pack_weights_dev");
        if (rc != IE_OK) return rc;
        q.Cout_p = channels[src_index]; q.s_tap = Cin * Cout; q.s_k = 1; q.s_o = Cout; q.off = cbase[src_index] * Cout; q.flip = 1;
        for (int c = 0; c < pl.nchunks; ++c) { q.chunk_k0[c] = pl.chunk_c0[c]; q.chunk_valid[c] = Cout - pl.chunk_c0[c] < KC ? Cout - pl.chunk_c0[c] : KC; }
    }
    if (pl.thin) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_mconv_pack_weights_dev: 8-channel (thin) sources are packed on the host only");
    q.w = w_hwio_dev; q.dst = dst_dev; q.nf = pl.nf; q.cout_tiles = pl.cout_tiles; q.nchunks = pl.nchunks; q.nstages = pl.nstages;
    q.taps_stage = pl.taps_stage; q.taps = pl.taps; q.dtype = dtype; q.total = pl.total_bytes / 2;
    hipLaunchKernelGGL(mconv_pack_dev_kernel, dim3((unsigned)((q.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
