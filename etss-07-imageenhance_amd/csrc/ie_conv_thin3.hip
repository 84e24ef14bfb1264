// A 3x3 convolution over the concatenation of up to three 64-channel sources with a THIN output (1 .. 4 channels, fp32), each source at its own resolution
// (nearest up-sampling to the output size folded into the addressing): RetinexNet RelightNet's tail, retinexnet/model.py:66-67, 83-88 -- F.interpolate(deconv1,
// deconv2 to the input size), cat, net2_fusion (1x1, replicate padding 1), net2_output (3x3) -- which generators.py composes into ONE 3x3 conv 192 -> 1 on the host.
//
// The tiled implicit GEMM spends a 16-row MFMA tile per output channel on it and runs at 2 TB/s; the layer is 168 bytes per pixel for 3.5 kFLOP: HBM-bound by far.
// Here: a workgroup marches down a 64-column strip; 6 waves = 3 sources x 2 halves of the strip.  A wave holds ITS source's 18 weight fragments (2 K chunks x 9 taps,
// 16-row fragments of which rows 0 .. cout-1 are real) in registers, loads its own source's rows into its own LDS ring (LDS-DMA; an up-sampled source just repeats
// pixels / rows), keeps three rotating accumulator sets over its two pixel tiles, and leaves its finished partial row (fp32, the real channels only) in LDS.  After
// the step's barrier the partial rows of the three sources are summed, bias and activation applied, and the row leaves as fp32 -- one small store per step.
// LDS: 3 sources x 3 rows x 8.5 KiB + the partial rows = 78 KiB at one output channel: two workgroups per CU.
#include "ie_conv_internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int TW = 64, ROW_PX = TW + 4;        // 64 body pixels, the two halo pixels, two spare ones (one 4-byte-per-lane DMA covers the four)
constexpr int PXB = 64;                        // one 32-channel chunk of a pixel
constexpr int CH_ROW_B = ROW_PX * PXB;         // 4,352
constexpr int SRC_ROW_B = 2 * CH_ROW_B;        // one ring row of one source
constexpr int MAXS = 3, PF = 2, NR = PF + 1;
constexpr int OFF_PART = MAXS * NR * SRC_ROW_B;                 // 78,336
constexpr unsigned OOB = 0x80000000u;

struct Thin3Params {
    const void* src[MAXS]; int sh[MAXS], sw[MAXS];
    const void* w; const float* bias; float* dst;
    int nsrc, N, H, W, cout, act, replicate, rows_per_block, strips, chunks;
};

__device__ __forceinline__ void wait_vm_cold(int n) {
    switch (n) {
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <typename T>
__device__ __forceinline__ void thin3_body(const Thin3Params& p) {
    using frag = ie_u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int s = wave >> 1, hf = wave & 1;           // source; half of the strip (tiles 2 hf, 2 hf + 1).  Waves of sources >= nsrc only keep the barriers company.
    const bool live = s < p.nsrc;
    const int si = live ? s : 0;

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = b;
    const int x0 = strip * TW;
    const int H = p.H, W = p.W, cout = p.cout;
    const int ys = chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, H);
    const int n_steps = (ye - ys) + 3;                // input rows ys - 1 .. ye, one more step for the last partial row
    const bool rep = p.replicate != 0;
    const int PART_B = MAXS * TW * cout * 4;          // one buffer of partial rows: [source][pixel][cout floats]; + 256 bytes dump for the lanes without a real channel

    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    };
    const int sh = p.sh[si], sw = p.sw[si];
    const size_t in_b = (size_t)sh * sw * 128;
    const auto rsS = mk((const char*)p.src[si] + (size_t)n * in_b, in_b);
    const size_t out_b = (size_t)H * W * cout * 4;
    const auto rsD = mk((char*)p.dst + (size_t)n * out_b, out_b);

    // ---- this wave's 18 weight fragments: ie_mconv_pack_weights image for Cout <= 16 (one 16-row fragment per (chunk, tap)); chunk = 2 s + c ----
    frag wv[18];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            wv[c * 9 + t] = *(const frag*)((const char*)p.w + ((size_t)((2 * si + c) * 9 + t) * 16 + p16) * 64 + ((kq ^ ((p16 >> 1) & 3)) << 4));

    // ---- LDS-DMA of this wave's half of its source's rows.  Source pixel of output column x: min(floor(x * sw / W), sw - 1) (F.interpolate nearest) ----
    const float fx = (float)sw / (float)W, fy = (float)sh / (float)H;
    auto map_x = [&](int x) __attribute__((always_inline)) { return sw == W ? x : min((int)floorf((float)x * fx), sw - 1); };
    auto map_y = [&](int y) __attribute__((always_inline)) { return sh == H ? y : min((int)floorf((float)y * fy), sh - 1); };
    unsigned body_off[2], halo_off;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (2 * hf + j) * 16 + (lane >> 2);
        int x = x0 + q;
        if (rep) x = min(x, W - 1);
        body_off[j] = x < W ? (unsigned)(map_x(x) * 128 + (((lane & 3) ^ ((q >> 1) & 3)) << 4)) : OOB;
    }
    {
        const int hp = lane >> 4;                       // ring pixels 64 (column x0 - 1), 65 (column x0 + 64), 66, 67 (spare)
        int x = hp == 0 ? x0 - 1 : x0 + TW;
        if (rep) x = min(max(x, 0), W - 1);
        halo_off = (hp < 2 && x >= 0 && x < W) ? (unsigned)(map_x(x) * 128 + (lane & 15) * 4) : OOB;
    }
    char* ring = smem + si * NR * SRC_ROW_B;
    auto issue = [&](int t) __attribute__((always_inline)) {        // input row of step t: 2 chunks x 2 body pieces + the halo of chunk hf
        int yi = ys - 1 + t;
        if (rep) yi = min(max(yi, 0), H - 1);
        const unsigned rmask = (yi >= 0 && yi < H) ? 0u : OOB;
        const unsigned row_b = rmask ? 0u : (unsigned)map_y(yi) * (unsigned)sw * 128u;
        char* slot = ring + (t % NR) * SRC_ROW_B;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(slot + c * CH_ROW_B + (2 * hf + j) * 1024), 16,
                                                         (int)((row_b + body_off[j] + c * 64) | rmask), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(slot + hf * CH_ROW_B + TW * PXB), 4,
                                                 (int)((row_b + halo_off + hf * 64) | rmask), 0, 0, 0);
    };

    // ---- fragment read offsets: lane (p16, kq) of tile j at tap column kx reads chunk piece kq of ring pixel 16 (2 hf + j) + p16 + kx - 1 (-1 -> 64, 64 -> 65) ----
    auto lds_px = [&](int q) __attribute__((always_inline)) { return q * PXB + ((kq ^ ((q >> 1) & 3)) << 4); };
    int rd[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 16 * (2 * hf + j) + p16 + kx - 1;
            rd[kx][j] = lds_px(q < 0 ? TW : (q >= TW ? TW + 1 : q));
        }
    auto row_mfma = [&](const char* slot, ie_f32x4 (&accN)[2], ie_f32x4 (&accM)[2], ie_f32x4 (&accO)[2]) __attribute__((always_inline)) {
        frag bx[4];
        auto rdf = [&](int g) __attribute__((always_inline)) {         // g = (chunk, kx, tile)
            const int c = g / 6, kx = (g % 6) >> 1, j = g & 1;
            bx[g & 3] = *(const frag*)(slot + c * CH_ROW_B + rd[kx][j]);
        };
        rdf(0); rdf(1); rdf(2);
        __builtin_amdgcn_sched_barrier(0);
        const ie_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int c = g / 6, kx = (g % 6) >> 1, j = g & 1;
            if (g + 3 < 12) rdf(g + 3);
            accN[j] = ie_mfma16_raw<T>(wv[c * 9 + 0 * 3 + kx], bx[g & 3], (c == 0 && kx == 0) ? zero4 : accN[j]);
            accM[j] = ie_mfma16_raw<T>(wv[c * 9 + 1 * 3 + kx], bx[g & 3], accM[j]);
            accO[j] = ie_mfma16_raw<T>(wv[c * 9 + 2 * 3 + kx], bx[g & 3], accO[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- partial rows: lanes kq == 0 hold channels 0 .. 3 of pixel (tile, p16); the others write into the dump area (no divergent branch near the DMAs) ----
    int pt_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = 16 * (2 * hf + j) + p16;
        pt_off[j] = (s * TW + q) * cout * 4;
    }
    // reduction: item = wave * 64 + lane < 64 * cout: pixel item / cout, channel item % cout
    const int item = wave * 64 + lane;
    const bool reducer = wave * 64 < TW * cout;         // wave-uniform
    const int rpx = item / cout, rch = item - rpx * cout;
    const float rbias = (reducer && p.bias && item < TW * cout) ? p.bias[rch] : 0.f;
    const unsigned rcol = (item < TW * cout && x0 + rpx < W) ? (unsigned)(((x0 + rpx) * cout + rch) * 4) : OOB;
    const int nst = reducer ? 1 : 0;

    ie_f32x4 a0[2], a1[2], a2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { a0[j] = ie_f32x4{0.f, 0.f, 0.f, 0.f}; a1[j] = a0[j]; a2[j] = a0[j]; }

    if (live) {
#pragma unroll
        for (int t = 0; t < PF; ++t) issue(t);
    }
    // step t: input row ys - 1 + t; accO finishes output row ys - 2 + t (its partial row goes to buffer t & 1); the partial rows of step t - 1 are summed and leave
    auto step = [&](int t, ie_f32x4 (&accN)[2], ie_f32x4 (&accM)[2], ie_f32x4 (&accO)[2]) __attribute__((always_inline)) {
        if (live) ie_wait_vm_hot<5, 6>(t == 0 ? 5 : 5 + nst, [](int n_) __attribute__((always_inline)) { wait_vm_cold(n_); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (reducer) {
            const int yo = ys - 3 + t;
            const unsigned rmask = (yo >= ys && yo < ye) ? 0u : OOB;
            const char* pb = smem + OFF_PART + ((t + 1) & 1) * (PART_B + 256) + (item < TW * cout ? item * 4 : 0);
            float v = rbias + *(const float*)pb;
            if (p.nsrc > 1) v += *(const float*)(pb + TW * cout * 4);
            if (p.nsrc > 2) v += *(const float*)(pb + 2 * TW * cout * 4);
            const int act = p.act;
            v = act == IE_ACT_SIGMOID ? ie_sigmoid_fast(v) : act == IE_ACT_TANH ? ie_tanh_fast(v) : act == IE_ACT_RELU ? fmaxf(v, 0.f) : v;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsD, (int)(((rmask ? 0u : (unsigned)yo * (unsigned)W * (unsigned)(cout * 4)) + rcol) | rmask), 0, 0);
        }
        if (live) {
            row_mfma(ring + (t % NR) * SRC_ROW_B, accN, accM, accO);
            issue(t + PF);
            char* pbuf = smem + OFF_PART + (t & 1) * (PART_B + 256);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    if (ch < cout) {          // uniform
                        const float pv = accO[j][ch];
                        *(float*)(kq == 0 ? pbuf + pt_off[j] + ch * 4 : pbuf + PART_B + lane * 4) = pv;
                    }
                }
            }
        }
    };
    for (int t = 0; t < n_steps; t += 3) {
        step(t, a0, a1, a2);
        if (t + 1 < n_steps) step(t + 1, a2, a0, a1);
        if (t + 2 < n_steps) step(t + 2, a1, a2, a0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(384) void sconv3_thin3_f16(Thin3Params p) { thin3_body<_Float16>(p); }
__global__ __launch_bounds__(384) void sconv3_thin3_bf16(Thin3Params p) { thin3_body<__bf16>(p); }

}  // namespace

bool ie_sconv3_thin3_applicable(int nsrc, const int* cs, int Cout, int kh, int kw, int stride, int pad, int pad_mode, int act, bool residual, bool clamp,
                                bool out_f32, bool shuffle, int W) {
    static const int off = getenv("IE_SCONV_THIN3") ? !atoi(getenv("IE_SCONV_THIN3")) : 0;        // IE_SCONV_THIN3=0: A/B switch back to the tiled kernel
    if (off || nsrc < 1 || nsrc > MAXS || Cout < 1 || Cout > 4 || kh != 3 || kw != 3 || stride != 1 || pad != 1 || residual || clamp || !out_f32 || shuffle || W < 64) return false;
    if (pad_mode != IE_PAD_ZEROS && pad_mode != IE_PAD_REPLICATE) return false;
    if (act != IE_ACT_NONE && act != IE_ACT_RELU && act != IE_ACT_SIGMOID && act != IE_ACT_TANH) return false;
    for (int s = 0; s < nsrc; ++s)
        if (cs[s] != 64) return false;
    return true;
}

int ie_sconv3_thin3_launch(const void* const* srcs, int nsrc, const int* src_h, const int* src_w, const void* wpk, const float* bias, float* dst, int dtype, int N,
                           int H, int W, int Cout, int act, int pad_mode, hipStream_t st) {
    if (dtype != IE_F16 && dtype != IE_BF16) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3_thin3: f16 / bf16");
    if ((double)H * W * 128.0 >= 2147483647.0) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3_thin3: frame too large for 31-bit offsets");
    Thin3Params p;
    memset(&p, 0, sizeof(p));
    for (int s = 0; s < nsrc; ++s) {
        p.src[s] = srcs[s];
        p.sh[s] = (src_h && src_h[s] > 0) ? src_h[s] : H;
        p.sw[s] = (src_w && src_w[s] > 0) ? src_w[s] : W;
    }
    p.w = wpk; p.bias = bias; p.dst = dst; p.nsrc = nsrc; p.N = N; p.H = H; p.W = W; p.cout = Cout; p.act = act; p.replicate = pad_mode == IE_PAD_REPLICATE;
    p.strips = ie_cdiv(W, TW);
    const int lds = OFF_PART + 2 * (MAXS * TW * Cout * 4 + 256);
    {
        const int slots = lds <= 81920 ? 512 : 256;          // two workgroups per CU where the LDS allows
        const long long base = (long long)p.strips * N;
        int best_c = 1;
        double best = -1.0;
        for (int c = 1; c <= H && c <= 64; ++c) {
            const int rpb = ie_cdiv(H, c), cc = ie_cdiv(H, rpb);
            const long long blocks = base * cc;
            const long long rounds = (blocks + slots - 1) / slots;
            const double score = (double)blocks / (double)(rounds * slots) * ((double)rpb / (double)(rpb + 3));
            if (score > best + 1e-9) { best = score; best_c = cc; }
        }
        p.rows_per_block = ie_cdiv(H, best_c);
        p.chunks = ie_cdiv(H, p.rows_per_block);
    }
    const long long blocks = (long long)p.strips * p.chunks * N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "ie_sconv3_thin3: bad grid");
    typedef void (*Fn)(Thin3Params);
    const int fi = dtype == IE_F16 ? 0 : 1;
    static const Fn fns[2] = {sconv3_thin3_f16, sconv3_thin3_bf16};
    IE_GRANT_LDS(fns[fi], OFF_PART + 2 * (MAXS * TW * 4 * 4 + 256));
    void* args[] = {(void*)&p};
    IE_HIP_CHECK(hipLaunchKernel((const void*)fns[fi], dim3((unsigned)blocks), dim3(384), args, lds, st));
    return IE_OK;
}
