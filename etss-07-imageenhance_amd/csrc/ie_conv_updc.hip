// The last decoder level of a 32-channel U-Net in ONE launch (unet.py:75-112, 186-195): cat([skip, up]) -> Conv3x3(64 -> 32) + BN + ReLU -> Conv3x3(32 -> 32) + BN
// + ReLU -> Conv1x1(32 -> classes) (+ clamp) -> NCHW fp32.  With `bilinear` the x2 up-sampling (align_corners=True, unet.py:82-83) happens inside as well: the
// kernel reads the HALF-resolution tensor and the full-resolution `up` tensor never exists.
//
// Why: at 1080p this level is HBM-bound as separate launches -- resize (16 B/px read, 64 written), conv1 (128 read, 64 written: 4.9 TB/s measured), conv2 + head
// (64 read, 12 written): 348 bytes per pixel for 55 kFLOP.  Fused it moves 64 + 16 + 12 = 92 and is bound by the matrix pipe.
//
// Structure = ie_conv_pair.hip at 32 channels: 8 waves = 2 stages x (2 output-channel fragments x 2 halves of the strip's 8 pixel tiles); weights in registers for
// the whole launch (stage A 18 fragments: 2 K chunks [skip, up] x 9 taps; stage B 9); stage A marches down the strip on rows of an LDS ring, three rotating
// accumulator sets (ky = 0, 1, 2), and hands each finished row of the intermediate (ReLU, 16 bits) to stage B through a second ring; stage B parks its finished
// rows as whole pixel lines, and the 1x1 head contracts a parked row one step later (the parked row is the MFMA's A operand, so a lane ends up holding four
// consecutive pixels of one output plane).  One s_barrier per step.  124 of the strip's 128 computed columns are stored.
//
// The skip chunk of the ring arrives by LDS-DMA (stage A's waves).  The up chunk: by LDS-DMA too (up given at full resolution), or, with `bilinear`, written by
// stage B's waves -- they have half of stage A's MFMA work -- from a small ring of half-resolution rows (their own LDS-DMA): every 16-byte piece of the ring row
// (8 channels of one pixel) is a weighted sum of four half-resolution pieces with fp32 weights and one rounding (v_fma_mix).
#include "ie_conv_internal.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr int TW = 128, TS = 124, ROW_PX = TW;          // 128 computed columns, no halo columns: layer a is valid on columns 1 .. 126, layer b on 2 .. 125
constexpr int PXB = 64;                       // one pixel of a 32-channel 16-bit tensor = one K chunk of a pixel
constexpr int CH_ROW_B = ROW_PX * PXB;        // 8,448
constexpr int SLOT_A = 2 * CH_ROW_B;          // stage A's ring row: skip chunk, up chunk
constexpr unsigned OOB = 0x80000000u;
constexpr int PF = 3, NRA = PF + 1, NRH = 3;
constexpr int PARK_B = TW * PXB;              // 8,192
constexpr int LOW_PX = 80, LOW_ROW_B = LOW_PX * PXB, NRL = 4;      // half-resolution staging: five 1 KiB DMA pieces per row, four rows
constexpr int OFF_H = NRA * SLOT_A;                     //  67,584
constexpr int OFF_PARK = OFF_H + NRH * CH_ROW_B;        //  92,928
constexpr int OFF_LOW = OFF_PARK + 2 * PARK_B;          // 109,312
constexpr int LDS_B = OFF_LOW + NRL * LOW_ROW_B;

struct UpdcParams {
    const void* skip; const void* up; const void* wa; const float* ba; const void* wb; const float* bb; const void* tail_w; const float* tail_b; float* dst;
    int N, H, W, h, w, rows_per_block, strips, chunks, tail_c, tail_clamp;
    float lo, hi, sy, sx;
};

// w00 * a + w01 * b + w10 * c + w11 * d on the eight 16-bit values of a 16-byte piece, fp32 arithmetic, one rounding.  (Measured and dropped for fp16:
// v_fma_mix_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16, which read the halves in place and write the rounded half -- 8 instructions per dword instead of 8
// converts + 4 packed-fp32 FMAs + a pack, but slower on gfx950: the level ran 1,378 frames/s of U-Net-32 with them against 1,45x with this expression.)
template <typename T>
__device__ __forceinline__ ie_u32x4 blend8(ie_u32x4 va, ie_u32x4 vb, ie_u32x4 vc, ie_u32x4 vd, float w00, float w01, float w10, float w11) {
    ie_u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // one fixed chain of FMAs: left to contract the sum itself, the compiler chose differently in the three unrolled copies of the step -- the same pixel then
        // rounded differently depending on where its row fell in a chunk (found by the batch-independence test at 1080p)
        const float l = __builtin_fmaf(w11, ie_unpack_lo<T>(vd[k]), __builtin_fmaf(w10, ie_unpack_lo<T>(vc[k]), __builtin_fmaf(w01, ie_unpack_lo<T>(vb[k]), w00 * ie_unpack_lo<T>(va[k]))));
        const float u = __builtin_fmaf(w11, ie_unpack_hi<T>(vd[k]), __builtin_fmaf(w10, ie_unpack_hi<T>(vc[k]), __builtin_fmaf(w01, ie_unpack_hi<T>(vb[k]), w00 * ie_unpack_hi<T>(va[k]))));
        o[k] = ie_pack2<T>(l, u);
    }
    return o;
}
// fp16: the same sum as two v_dot2_f32_f16 per value pair -- (a, b) . (w00, w01) + (c, d) . (w10, w11), exact products, fp32 accumulation, one rounding of the
// result -- on operands paired by v_perm_b32: 9 instructions per dword instead of 17.  The four weights enter as fp16 (the blend's weights are in [0, 1]: 2^-12
// relative), which raises the rms error of an up-sampled value from 1.39e-4 to 1.94e-4 of full scale, its maximum unchanged (one fp16 ulp); the arithmetic of the
// blend is 6 % of the whole U-Net-32 forward (1,415 - 1,437 frames/s at batch 32 with it against 1,507 - 1,522 without any: profiles/r03/updc_experiments.log).
#ifndef IE_UPDC_DOT2
#define IE_UPDC_DOT2 1
#endif
#if IE_UPDC_DOT2
template <>
__device__ __forceinline__ ie_u32x4 blend8<_Float16>(ie_u32x4 va, ie_u32x4 vb, ie_u32x4 vc, ie_u32x4 vd, float w00, float w01, float w10, float w11) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 wab = __builtin_bit_cast(h2, ie_pack2<_Float16>(w00, w01)), wcd = __builtin_bit_cast(h2, ie_pack2<_Float16>(w10, w11));
    ie_u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned a = va[k], b = vb[k], c = vc[k], d = vd[k];
        const h2 ab_lo = __builtin_bit_cast(h2, __builtin_amdgcn_perm(b, a, 0x05040100u)), ab_hi = __builtin_bit_cast(h2, __builtin_amdgcn_perm(b, a, 0x07060302u));
        const h2 cd_lo = __builtin_bit_cast(h2, __builtin_amdgcn_perm(d, c, 0x05040100u)), cd_hi = __builtin_bit_cast(h2, __builtin_amdgcn_perm(d, c, 0x07060302u));
        const float l = __builtin_amdgcn_fdot2(ab_lo, wab, __builtin_amdgcn_fdot2(cd_lo, wcd, 0.f, false), false);
        const float u = __builtin_amdgcn_fdot2(ab_hi, wab, __builtin_amdgcn_fdot2(cd_hi, wcd, 0.f, false), false);
        o[k] = ie_pack2<_Float16>(l, u);
    }
    return o;
}
#endif

template <typename T, bool BILIN>
__device__ __forceinline__ void updc_body(const UpdcParams& p) {
    using frag = ie_u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int stg = wave >> 2, w4 = wave & 3, cf = w4 & 1, hf = w4 >> 1;      // stage; 16-cout fragment; half of the strip (pixel tiles 4 hf .. 4 hf + 3)

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = b;
    const int xc0 = strip * TS - 2;                   // image column of computed column 0
    const int H = p.H, W = p.W;
    const int ys = chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, H);
    const int n_steps = (ye - ys) + 8;

    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    };
    const size_t frame_b = (size_t)H * W * PXB;
    const size_t up_frame_b = BILIN ? (size_t)p.h * p.w * PXB : frame_b;
    const auto rsS = mk((const char*)p.skip + (size_t)n * frame_b, frame_b);
    const auto rsU = mk((const char*)p.up + (size_t)n * up_frame_b, up_frame_b);
    const size_t dst_b = (size_t)p.tail_c * H * W * 4;
    const auto rsD = mk((char*)p.dst + (size_t)n * dst_b, dst_b);

    // ---- this wave's weight fragments (stage A: 2 chunks x 9 taps, stage B: 9 taps) and 4 biases; images of ie_mconv_pack_weights (Cout 32: 32 rows per tap) ----
    frag wv[18];
    {
        const char* wp = (const char*)(stg ? p.wb : p.wa);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                wv[c * 9 + t] = (stg && c) ? frag{0u, 0u, 0u, 0u} : *(const frag*)(wp + ((size_t)(c * 9 + t) * 32 + cf * 16 + p16) * 64 + ((kq ^ ((p16 >> 1) & 3)) << 4));
    }
    ie_f32x4 bias4;
    {
        const float* bp = stg ? p.bb : p.ba;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = bp ? bp[16 * cf + 4 * kq + i] : 0.f;
    }

    // ---- stage A: LDS-DMA of the input rows.  Piece m = 16 pixels x 64 B of one chunk; position (lane & 3) of ring pixel q holds channel piece
    //      (lane & 3) ^ ((q >> 1) & 3).  Wave w4 loads pieces 2 w4, 2 w4 + 1 of the skip chunk (and of the up chunk when it comes at full resolution).  No halo
    //      columns: the outermost computed column of each layer is simply not used by the next (4 of 128 columns lost, no halo DMA and -- with the in-kernel
    //      up-sampling -- no third, nearly empty pass over the row for 8 halo pieces).
    unsigned body_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (2 * w4 + j) * 16 + (lane >> 2);
        const int x = xc0 + q;
        body_off[j] = (x >= 0 && x < W) ? (unsigned)(x * PXB + (((lane & 3) ^ ((q >> 1) & 3)) << 4)) : OOB;
    }
    auto issue = [&](int t) __attribute__((always_inline)) {        // input row of stage A's step t
        const int yi = ys - 2 + t;
        const unsigned rmask = (yi >= 0 && yi < H) ? 0u : OOB;
        const unsigned row_b = rmask ? 0u : (unsigned)yi * (unsigned)W * PXB;
        char* slot = smem + (t % NRA) * SLOT_A;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(slot + (2 * w4 + j) * 1024), 16,
                                                     (int)((row_b + body_off[j]) | rmask), 0, 0, 0);
        if constexpr (!BILIN) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(slot + CH_ROW_B + (2 * w4 + j) * 1024), 16,
                                                         (int)((row_b + body_off[j]) | rmask), 0, 0, 0);
        }
    };
    constexpr int NSRC_DMA = BILIN ? 1 : 2;

    // ---- fragment read offsets: lane (p16, kq) of the wave's tile j at tap column kx reads chunk piece kq of ring pixel 16 (4 hf + j) + p16 + kx - 1.  The swizzle
    //      term repeats every 8 pixels, so tile j's offset is tile 1's + 1024 (j - 1) -- an immediate -- except tile 0 (its own registers) and the last tile at
    //      kx = 2: pixels -1 and 128 do not exist, those reads are clamped into the row (they only feed the unused outermost column).
    auto lds_px = [&](int q) __attribute__((always_inline)) { return q * PXB + ((kq ^ ((q >> 1) & 3)) << 4); };
    int rd1[3], rd0[3], rd23;
    {
        auto fix = [&](int q) __attribute__((always_inline)) { return min(max(q, 0), TW - 1); };
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            rd1[kx] = lds_px(16 * (4 * hf + 1) + p16 + kx - 1);
            rd0[kx] = lds_px(fix(16 * (4 * hf) + p16 + kx - 1));
        }
        rd23 = lds_px(fix(16 * (4 * hf + 3) + p16 + 1));
    }
    // one row into the three live accumulator sets (ky = 0: the set this row starts takes the bias as its first C operand); NCK K chunks.
    // The order is pinned, fragment reads three ahead of their MFMAs (ie_conv_pair.hip: left alone the scheduler waits lgkmcnt(0) behind every read).
    auto row_mfma = [&](const char* slot, auto nck_tag, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
        constexpr int NCK = decltype(nck_tag)::value, NG = NCK * 12;
        frag bx[4];
        auto rdf = [&](int g) __attribute__((always_inline)) {         // g = (chunk, kx, tile), compile-time at every call
            const int c = g / 12, kx = (g % 12) >> 2, j = g & 3;
            const int off = j == 0 ? rd0[kx] : ((kx == 2 && j == 3) ? rd23 : rd1[kx] + 1024 * (j - 1));
            bx[g & 3] = *(const frag*)(slot + c * CH_ROW_B + off);
        };
        rdf(0); rdf(1); rdf(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c = g / 12, kx = (g % 12) >> 2, j = g & 3;
            if (g + 3 < NG) rdf(g + 3);
            accN[j] = ie_mfma16_raw<T>(wv[c * 9 + 0 * 3 + kx], bx[g & 3], (c == 0 && kx == 0) ? bias4 : accN[j]);
            accM[j] = ie_mfma16_raw<T>(wv[c * 9 + 1 * 3 + kx], bx[g & 3], accM[j]);
            accO[j] = ie_mfma16_raw<T>(wv[c * 9 + 2 * 3 + kx], bx[g & 3], accO[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using two = std::integral_constant<int, 2>;
    using one = std::integral_constant<int, 1>;

    // ---- epilogues: tile pair jj (the wave's tiles 2 jj, 2 jj + 1) of a finished row as 16-byte pieces (lanes l / l + 16 exchange halves): after the exchange
    //      even kq -> tile 2 jj, odd kq -> tile 2 jj + 1; the lane holds couts 16 cf + 8 (kq >> 1) .. + 7 of ONE pixel = channel piece 2 cf + (kq >> 1)
    int em_off;                    // pair 0; pair 1 is 32 pixels on (the swizzle repeats every 8 pixels)
    unsigned em_in = 0;            // bit jj: the lane's pixel of pair jj is inside the image
    {
        const int q = 16 * (4 * hf + (kq & 1)) + p16;
        em_off = q * PXB + (((2 * cf + (kq >> 1)) ^ ((q >> 1) & 3)) << 4);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int x = xc0 + q + 32 * jj;
            em_in |= (x >= 0 && x < W) ? (1u << jj) : 0u;
        }
    }
    auto put_row = [&](const ie_f32x4 (&acc)[4], char* base, unsigned keep) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const ie_u32x2 a = {ie_pack2_relu<T>(acc[2 * jj][0], acc[2 * jj][1]), ie_pack2_relu<T>(acc[2 * jj][2], acc[2 * jj][3])};
            const ie_u32x2 b2 = {ie_pack2_relu<T>(acc[2 * jj + 1][0], acc[2 * jj + 1][1]), ie_pack2_relu<T>(acc[2 * jj + 1][2], acc[2 * jj + 1][3])};
            const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
            const bool k = (keep >> jj) & 1u;
            *(ie_u32x4*)(base + em_off + jj * 32 * PXB) = ie_u32x4{k ? sx[0] : 0u, k ? sy[0] : 0u, k ? sx[1] : 0u, k ? sy[1] : 0u};
        }
    };

    ie_f32x4 a0[4], a1[4], a2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a0[j] = bias4; a1[j] = bias4; a2[j] = bias4; }

    if (stg == 0) {
        // Stage A, step t: input row ys - 2 + t; the intermediate's row m = ys - 3 + t is finished (accO) and written into hand-off slot t % NRH.  Rows and columns
        // of the intermediate outside the image are layer b's zero padding: written as zeros.
#pragma unroll
        for (int t = 0; t < PF; ++t) issue(t);
        if constexpr (BILIN) __builtin_amdgcn_s_barrier();          // stage B's prologue barrier (its staging rows have landed)
        auto step_a = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            ie_wait_vm_imm<(PF - 1) * 2 * NSRC_DMA>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            row_mfma(smem + (t % NRA) * SLOT_A, two{}, accN, accM, accO);
            issue(t + PF);
            const int m = ys - 3 + t;
            put_row(accO, smem + OFF_H + (t % NRH) * CH_ROW_B, (m >= 0 && m < H) ? em_in : 0u);
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_a(t, a0, a1, a2);
            if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0);
        }
    } else {
        // Stage B, step t: consumes the intermediate's row ys - 5 + t (stage A's step t - 2), finishing layer b's row ys - 6 + t; accN still holds the row finished at
        // step t - 1 (parked now), and the row parked at step t - 1 (ys - 8 + t) goes through the 1x1 head and leaves.
        // head: tiles 2 w4, 2 w4 + 1 of the parked row; A operand = the row (lane = pixel p16, channels 8 kq ..), B operand = the 1x1 weights [16][32] (lane = output
        // channel p16); D: lane (p16 = output channel, kq) holds pixels 4 kq .. 4 kq + 3 of the tile
        const frag tW = *(const frag*)((const char*)p.tail_w + p16 * PXB + kq * 16);
        const float tl_bias = p.tail_b[p16];
        int tl_lds[2];
        unsigned tl_col[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = 16 * (2 * w4 + t) + p16;
            tl_lds[t] = q * PXB + ((kq ^ ((q >> 1) & 3)) << 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qo = 16 * (2 * w4 + t) + 4 * kq + i, xo = xc0 + qo;
                tl_col[t][i] = (p16 < p.tail_c && qo >= 2 && qo <= TS + 1 && xo < W) ? (unsigned)((p16 * H * W + xo) * 4) : OOB;
            }
        }
        // park: the same 16-byte pieces as the hand-off rows, pixel line = 64 bytes
        int pk_off;
        {
            const int q = 16 * (4 * hf + (kq & 1)) + p16;
            pk_off = q * PXB + (((2 * cf + (kq >> 1)) ^ ((q >> 1) & 3)) << 4);
        }
        auto park_row = [&](const ie_f32x4 (&acc)[4], char* base) __attribute__((always_inline)) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const ie_u32x2 a = {ie_pack2_relu<T>(acc[2 * jj][0], acc[2 * jj][1]), ie_pack2_relu<T>(acc[2 * jj][2], acc[2 * jj][3])};
                const ie_u32x2 b2 = {ie_pack2_relu<T>(acc[2 * jj + 1][0], acc[2 * jj + 1][1]), ie_pack2_relu<T>(acc[2 * jj + 1][2], acc[2 * jj + 1][3])};
                const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
                *(ie_u32x4*)(base + pk_off + jj * 32 * PXB) = ie_u32x4{sx[0], sy[0], sx[1], sy[1]};
            }
        };
        // ---- bilinear: stage B's waves fill the up chunk of stage A's ring rows from a staging ring of half-resolution rows ------------------------------
        // Full-resolution row y, column x: fy = y * sy, y0 = floor(fy), wy = fy - y0, y1 = min(y0 + 1, h - 1), the same along x (torch's align_corners=True
        // source index, upsample_bilinear2d); value = sum of the four neighbours with fp32 weights, rounded once.  Rows / columns outside the image get zero
        // weights: they are conv a's zero padding.
        // A lane owns two 16-byte pieces of the ring row: pixel q = (e * 256 + 64 w4 + lane) >> 2, channel piece c = lane & 3, e = 0, 1.
        int lo0[2], lo1[2], rw[2];
        float cw0[2], cw1[2];
        unsigned low_src[2];
        int lxs = 0;
        if constexpr (BILIN) {
            const int xmin = max(xc0, 0);
            lxs = (int)((float)xmin * p.sx);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma clang fp contract(off)
                const int idx = e * 256 + w4 * 64 + lane, rq = idx >> 2, c = idx & 3, x = xc0 + rq;
                const bool ok = x >= 0 && x < W;
                const int xcl = min(max(x, 0), W - 1);
                const float fx = (float)xcl * p.sx;
                const int x0 = (int)fx, x1 = min(x0 + 1, p.w - 1);
                const float wx = fx - (float)x0;
                lo0[e] = (x0 - lxs) * PXB + c * 16;
                lo1[e] = (x1 - lxs) * PXB + c * 16;
                cw0[e] = ok ? 1.f - wx : 0.f;
                cw1[e] = ok ? wx : 0.f;
                rw[e] = CH_ROW_B + rq * PXB + ((c ^ ((rq >> 1) & 3)) << 4);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {          // staging pieces: wave w4 loads piece w4, wave 0 also piece 4 (16 pixels x 64 B each)
                const int px = (m ? 4 : w4) * 16 + (lane >> 2), lx = lxs + px;
                low_src[m] = lx < p.w ? (unsigned)(lx * PXB + (lane & 3) * 16) : OOB;
            }
        }
        auto low_rows = [&](int y, int& y0, int& y1, float& wy0, float& wy1) __attribute__((always_inline)) {      // wave-uniform
#pragma clang fp contract(off)       // fy is a ROUNDED product (torch: src = scale * dst); contracted into fy - y0 in only some of the step's unrolled copies, the same row
                                     // got a different weight depending on where it fell in a chunk (found by the batch-independence test at 1080p)
            const bool ok = y >= 0 && y < H;
            const int ycl = min(max(y, 0), H - 1);
            const float fy = (float)ycl * p.sy;
            y0 = (int)fy;
            y1 = min(y0 + 1, p.h - 1);
            const float wy = fy - (float)y0;
            wy0 = ok ? 1.f - wy : 0.f;
            wy1 = ok ? wy : 0.f;
        };
        auto issue_low = [&](int r) __attribute__((always_inline)) {        // half-resolution row r (inside the tensor) into staging slot r % NRL
            char* slot = smem + OFF_LOW + (r % NRL) * LOW_ROW_B;
            const unsigned row_b = (unsigned)r * (unsigned)p.w * PXB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(slot + w4 * 1024), 16, (int)(row_b + low_src[0]), 0, 0, 0);
            if (w4 == 0)          // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(slot + 4 * 1024), 16, (int)(row_b + low_src[1]), 0, 0, 0);
        };
        auto interp_row = [&](int y, char* aslot) __attribute__((always_inline)) {       // full-resolution row y -> the up chunk of ring slot `aslot`
            int y0, y1;
            float wy0, wy1;
            low_rows(y, y0, y1, wy0, wy1);
            const char* r0 = smem + OFF_LOW + (y0 % NRL) * LOW_ROW_B;
            const char* r1 = smem + OFF_LOW + (y1 % NRL) * LOW_ROW_B;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const ie_u32x4 va = *(const ie_u32x4*)(r0 + lo0[e]), vb = *(const ie_u32x4*)(r0 + lo1[e]);
                const ie_u32x4 vc = *(const ie_u32x4*)(r1 + lo0[e]), vd = *(const ie_u32x4*)(r1 + lo1[e]);
                const float w00 = wy0 * cw0[e], w01 = wy0 * cw1[e], w10 = wy1 * cw0[e], w11 = wy1 * cw1[e];
#if defined(IE_UPDC_EXP) && IE_UPDC_EXP == 1       // timing experiment: no blend arithmetic (wrong results)
                const ie_u32x4 o = va ^ vb ^ vc ^ vd;
                (void)w00; (void)w01; (void)w10; (void)w11;
#else
                const ie_u32x4 o = blend8<T>(va, vb, vc, vd, w00, w01, w10, w11);
#endif
                *(ie_u32x4*)(aslot + rw[e]) = o;
            }
        };
        constexpr int NST_B = 8;                       // the head's store instructions per step of a wave
        if constexpr (BILIN) {
            // prologue: four staging rows from the first one the rows ys - 2 .. ys + 2 need; then stage A's first three ring rows (ys - 2, ys - 1, ys)
            int y0, y1;
            float wa_, wb_;
            low_rows(ys - 2, y0, y1, wa_, wb_);
#pragma unroll
            for (int r = 0; r < NRL; ++r) issue_low(min(y0 + r, p.h - 1));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int t = 0; t < PF; ++t) interp_row(ys - 2 + t, smem + (t % NRA) * SLOT_A);
        }
        auto step_b = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            if constexpr (BILIN) {       // the staging row issued at step t - 2 has landed: behind it 2 x NST_B stores and step t - 1's DMA
                if (w4 == 0) ie_wait_vm_imm<2 * NST_B + 2>();
                else ie_wait_vm_imm<2 * NST_B + 1>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (BILIN) {
                interp_row(ys + 1 + t, smem + ((t + PF) % NRA) * SLOT_A);          // stage A's input row of step t + 3
                int y0, y1;
                float wa_, wb_;
                low_rows(ys + 3 + t, y0, y1, wa_, wb_);
                issue_low(y1);                                                        // the new staging row of step t + 2 (re-issued while it repeats: uniform counts)
            }
            const int yo = ys - 8 + t;
            const unsigned rmask = (yo >= ys && yo < ye) ? 0u : OOB;
            const unsigned trow_b = rmask ? 0u : (unsigned)yo * (unsigned)W * 4u;
            const char* islot = smem + OFF_PARK + ((t + 1) & 1) * PARK_B;
            frag ta[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) ta[i] = *(const frag*)(islot + tl_lds[i]);
            park_row(accN, smem + OFF_PARK + (t & 1) * PARK_B);
            // The 1x1 head of the row parked a step earlier and its stores come BEFORE this step's MFMA stream (round 4): at the head of a step the older stage holds the
            // matrix pipe anyway (finding 14), so everything of this stage that is not part of the stream belongs here (IE_UPDC_TAIL_LAST=1: behind it, A/B)
            auto head_1x1 = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ie_f32x4 o = {tl_bias, tl_bias, tl_bias, tl_bias};
                    o = ie_mfma16_raw<T>(ta[i], tW, o);
                    if (p.tail_clamp) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(o[e], p.lo), p.hi);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float oe = o[e];       // (__builtin_bit_cast straight on the vector ELEMENT o[e] compiled to element 0 four times: hipcc 7.2)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, oe), rsD, (int)((trow_b + tl_col[i][e]) | rmask), 0, 0);
                    }
                }
            };
#ifndef IE_UPDC_TAIL_LAST
#define IE_UPDC_TAIL_LAST 0
#endif
            if constexpr (!IE_UPDC_TAIL_LAST) { head_1x1(); __builtin_amdgcn_sched_barrier(0); }
            row_mfma(smem + OFF_H + ((t + 1) % NRH) * CH_ROW_B, one{}, accN, accM, accO);       // stage A's step t - 2 wrote slot (t - 2) % NRH == (t + 1) % NRH
            if constexpr (IE_UPDC_TAIL_LAST) head_1x1();
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_b(t, a0, a1, a2);
            if (t + 1 < n_steps) step_b(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_b(t + 2, a1, a2, a0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) void unet_updc32_head_f16(UpdcParams p) { updc_body<_Float16, false>(p); }
__global__ __launch_bounds__(512) void unet_updc32_head_bf16(UpdcParams p) { updc_body<__bf16, false>(p); }
__global__ __launch_bounds__(512) void unet_updc32_bilinear_head_f16(UpdcParams p) { updc_body<_Float16, true>(p); }
__global__ __launch_bounds__(512) void unet_updc32_bilinear_head_bf16(UpdcParams p) { updc_body<__bf16, true>(p); }

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------------------------------------------
IE_EXPORT int ie_unet_up_head_applicable(int C, int tail_cout, int H, int W, int up_h, int up_w) {
    static const int off = getenv("IE_UNET_UPHEAD") ? !atoi(getenv("IE_UNET_UPHEAD")) : 0;       // IE_UNET_UPHEAD=0: A/B switch back to separate launches
    if (off || C != 32 || tail_cout < 1 || tail_cout > 16 || H < 1 || W < 64 || (double)H * W * 64.0 >= 2147483647.0 || (double)H * W * tail_cout * 4.0 >= 2147483647.0)
        return 0;
    if (up_h == H && up_w == W) return 1;
    static const int nobil = getenv("IE_UNET_UPHEAD_BILINEAR") ? !atoi(getenv("IE_UNET_UPHEAD_BILINEAR")) : 0;       // =0: A/B, up-sample in its own launch
    return (!nobil && up_h >= 2 && up_w >= 2 && H == 2 * up_h && W == 2 * up_w) ? 1 : 0;
}

IE_EXPORT int ie_unet_up_head_nhwc(const void* skip, const void* up, int up_h, int up_w, const void* packed_w_a, const float* bias_a, const void* packed_w_b,
                                   const float* bias_b, const void* tail_w, const float* tail_bias, int tail_cout, int clamp, float clamp_lo, float clamp_hi,
                                   float* dst_nchw, int dtype, int N, int H, int W, void* stream) {
    if (!skip || !up || !packed_w_a || !packed_w_b || !tail_w || !tail_bias || !dst_nchw || N <= 0) IE_FAIL(IE_ERR_INVALID, "ie_unet_up_head_nhwc: null pointer / bad batch");
    if (dtype != IE_F16 && dtype != IE_BF16) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_unet_up_head_nhwc: f16 / bf16");
    if (!ie_unet_up_head_applicable(32, tail_cout, H, W, up_h, up_w))
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_unet_up_head_nhwc: 32 channels, W >= 64, `up` at the skip's resolution or exactly half of it -- ask ie_unet_up_head_applicable first");
    UpdcParams p;
    memset(&p, 0, sizeof(p));
    p.skip = skip; p.up = up; p.wa = packed_w_a; p.ba = bias_a; p.wb = packed_w_b; p.bb = bias_b; p.tail_w = tail_w; p.tail_b = tail_bias; p.dst = dst_nchw;
    p.N = N; p.H = H; p.W = W; p.h = up_h; p.w = up_w; p.tail_c = tail_cout; p.tail_clamp = clamp; p.lo = clamp_lo; p.hi = clamp_hi;
    p.sy = H > 1 ? (float)(up_h - 1) / (float)(H - 1) : 0.f;
    p.sx = W > 1 ? (float)(up_w - 1) / (float)(W - 1) : 0.f;
    p.strips = ie_cdiv(W, TS);
    {
        // row chunks: whole rounds of the chip's 256 workgroup slots (one 8-wave workgroup per CU), few apron rows (8 steps per chunk)
        const int slots = 256;
        const long long base = (long long)p.strips * N;
        int best_c = 1;
        double best = -1.0;
        for (int c = 1; c <= H && c <= 64; ++c) {
            const int rpb = ie_cdiv(H, c), cc = ie_cdiv(H, rpb);
            const long long blocks = base * cc;
            const long long rounds = (blocks + slots - 1) / slots;
            const double score = (double)blocks / (double)(rounds * slots) * ((double)rpb / (double)(rpb + 8));
            if (score > best + 1e-9) { best = score; best_c = cc; }
        }
        p.rows_per_block = ie_cdiv(H, best_c);
        p.chunks = ie_cdiv(H, p.rows_per_block);
    }
    const long long blocks = (long long)p.strips * p.chunks * N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "ie_unet_up_head_nhwc: bad grid");
    typedef void (*Fn)(UpdcParams);
    const int fi = (dtype == IE_F16 ? 0 : 1) + ((up_h != H || up_w != W) ? 2 : 0);
    static const Fn fns[4] = {unet_updc32_head_f16, unet_updc32_head_bf16, unet_updc32_bilinear_head_f16, unet_updc32_bilinear_head_bf16};
    IE_GRANT_LDS(fns[fi], LDS_B);
    void* args[] = {(void*)&p};
    IE_HIP_CHECK(hipLaunchKernel((const void*)fns[fi], dim3((unsigned)blocks), dim3(512), args, LDS_B, (hipStream_t)stream));
    return IE_OK;
}
