// Two stacked 3x3 convolutions 64 -> 64 -> 64 (+ bias + ReLU each) in ONE strip kernel, the intermediate never in HBM: RetinexNet's DecomNet body
// (retinexnet/model.py:28-33: five Conv3x3(64, 64) + ReLU, replicate padding) runs as two pairs + one single layer.
//
// Why: a 64 -> 64 layer at 1080p moves 256 bytes per pixel for 73.7 kFLOP -- 288 FLOP/B, under the 312 FLOP/B machine balance; the single-layer strip
// kernel (ie_conv_strip.hip) holds 1,034 TFLOP/s at 3.6 TB/s on it.  The pair moves the same 256 bytes for twice the work.  It is the structure of the
// Zero-DCE kernel K345 (ie_zdce_fused.hip) at 64 channels:
//   * 8 waves = 2 stages x 4 output-channel fragments; a wave owns 16 output channels of ITS layer on all 128 columns of the strip (8 pixel tiles) and
//     keeps that layer's 18 weight fragments (2 K chunks x 9 taps) in registers for the whole launch: 72 weight + 96 accumulator VGPRs;
//   * stage A marches down the strip on input rows DMA'd into an LDS ring (buffer_load ... lds, counted vmcnt waits), every pixel fragment feeding three
//     MFMAs (ky = 0, 1, 2) into three rotating accumulator sets, and writes each finished row of the intermediate (ReLU, 16-bit) into a second LDS ring
//     in exactly the swizzled [chunk][pixel][32 ch] image the fragment reads expect; stage B consumes it two steps later the same way;
//   * finished output rows are parked in LDS as whole 128-byte pixel lines and leave one step later, 1 KiB of whole lines per store instruction.
// One s_barrier per step orders everything.  Stage B needs one column of the intermediate either side: 126 of the 128 computed columns are stored.
//
// Padding.  zeros: rows / columns of the intermediate outside the image are written as zeros (they are layer b's padding, not layer a of padding).
// replicate: layer b pads the INTERMEDIATE by replication, so stage B reads it through clamped coordinates: column reads are clamped per lane (a table of
// 24 offsets), the first / last row is consumed twice (the ring keeps a row for the second read; rows outside the image are never written).
//
// Same packed weight images as ie_mconv_nhwc (ie_mconv_pack_weights, Cout 64, one 64-channel source); same roundings as the two-launch path (fp32
// accumulation in the same order, the intermediate rounded to 16 bits once), so the result is bit-identical to it (tests/test_mconv_gpu.py).
#include "ie_conv_internal.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr int TW = 128, TS = 126, ROW_PX = TW + 4;      // 128 body pixels, the two halo pixels, two spare ones (the halo DMA below covers all four)
constexpr int PXB = 64;                       // bytes of one 32-channel chunk of a pixel
constexpr int CH_ROW_B = ROW_PX * PXB;        // one ring row of one chunk
constexpr int SLOT_B = 2 * CH_ROW_B;          // one ring row (two chunks)
constexpr unsigned OOB = 0x80000000u;
constexpr int PF = 3, NRA = PF + 1, NRH = 3;
constexpr int OPB = 128;                      // output bytes per pixel
constexpr int OUT_ROW_B = TW * OPB;
constexpr int OFF_H = NRA * SLOT_B, OFF_OUT = OFF_H + NRH * SLOT_B, LDS_B = OFF_OUT + 2 * OUT_ROW_B;      // 67,584 + 50,688 + 32,768 = 151,040
constexpr int LDS_POOL_B = NRA * CH_ROW_B + NRH * SLOT_B + 3 * OUT_ROW_B;                               // 32 -> 64 -> 64 + pool: 33,792 + 50,688 + 49,152 = 133,632

struct PairParams {
    const void* src; const void* wa; const float* ba; const void* wb; const float* bb; void* dst;
    int N, H, W, rows_per_block, strips, chunks, replicate;
    int cout_b, act_b;       // thin second layer (CB == 16): its real output channels (4, 8, 12 or 16) and activation; dst is fp32 NHWC then
    void* pool;              // POOL: the 2x2 max-pooled copy of dst (NHWC 64 channels, N x H/2 x W/2)
    void* dst16;             // thin: optional second copy of the output as 16-bit NHWC (cout_b == 4), with dst16_c = 4 channels or padded to 8, or null
    int dst16_c;
};

__device__ __forceinline__ void wait_vm_cold(int n) {
    switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// CB = 64: the second layer is 64 -> 64 + ReLU, 16-bit output.  CB = 16: the second layer is thin (64 -> cout_b <= 16, any activation, fp32 output):
// its four waves split the strip by pixel tiles (two each) instead of by output channels, and a finished row leaves straight from the accumulators --
// lane (pixel, kq) holds channels 4 kq .. 4 kq + 3 of its pixel, 16 contiguous bytes of the fp32 NHWC row.
// CA = 32: the first layer is 32 -> 64 (one K chunk: U-Net-32's `down1`, unet.py:70-72 + 43-48).  POOL: the 64-channel output also leaves max-pooled 2x2 (three parked
// rows; a pooled row every second step from the two newest ones; values are ReLU outputs: compared as unsigned 16-bit integers).
template <typename T, int CB, int CA = 64, bool POOL = false>
__device__ __forceinline__ void sconv3x2_body(const PairParams& p) {
    constexpr bool THIN = CB == 16;
    constexpr int NCA = CA / 32, IPB = CA * 2;                     // stage A's K chunks; input bytes per pixel
    constexpr int SLOT_A = NCA * CH_ROW_B;                         // stage A's ring row
    constexpr int OFF_H = NRA * SLOT_A, OFF_OUT = OFF_H + NRH * SLOT_B, NPARK = POOL ? 3 : 2;
    static_assert(!POOL || (!THIN && CA == 32), "the pooled form exists for 32 -> 64 -> 64 (LDS)");
    using frag = ie_u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int stg = wave >> 2, cf = wave & 3;         // stage (0: layer a, 1: layer b), 16-cout fragment of the stage's layer

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = b;
    const int xc0 = strip * TS - 1;                   // image column of computed column 0
    const int H = p.H, W = p.W;
    const int ys = chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, H);
    const int n_steps = (ye - ys) + 8;
    const bool rep = p.replicate != 0;

    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    };
    const size_t frame_b = (size_t)H * W * IPB;
    const auto rsS = mk((const char*)p.src + (size_t)n * frame_b, frame_b);
    const size_t dframe_b = THIN ? (size_t)H * W * p.cout_b * 4 : (size_t)H * W * OPB;
    const int Hp = H >> 1, Wp = W >> 1;
    const size_t pool_b = (size_t)Hp * Wp * OPB;
    const auto rsP = mk(POOL ? (char*)p.pool + (size_t)n * pool_b : (char*)p.dst, POOL ? pool_b : 0);
    const auto rsD = mk((char*)p.dst + (size_t)n * dframe_b, dframe_b);

    // ---- this wave's 18 weight fragments and 4 biases ----------------------------------------------------------------------------------
    frag wv[18];
    {
        const char* wp = (const char*)(stg ? p.wb : p.wa);
        const int wrows = (THIN && stg) ? 16 : 64, wcf = (THIN && stg) ? 0 : cf;      // the thin layer's image has one 16-row fragment per (chunk, tap)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                wv[c * 9 + t] = (stg == 0 && c >= NCA) ? frag{0u, 0u, 0u, 0u}
                                                        : *(const frag*)(wp + ((size_t)(c * 9 + t) * wrows + wcf * 16 + p16) * 64 + ((kq ^ ((p16 >> 1) & 3)) << 4));
    }
    ie_f32x4 bias4;
    {
        const float* bp = stg ? p.bb : p.ba;
        const int c0 = (THIN && stg) ? 4 * kq : 16 * cf + 4 * kq, cmax = (THIN && stg) ? p.cout_b : 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = (bp && c0 + i < cmax) ? bp[c0 + i] : 0.f;
    }

    // ---- stage A: LDS-DMA of the input rows.  Piece m = 16 pixels x 64 B of one chunk; position (lane & 3) of ring pixel q holds channel piece
    //      (lane & 3) ^ ((q >> 1) & 3).  Wave cf loads pieces 2 cf, 2 cf + 1 of both chunks; wave 3 also the two halo pixels (ring pixels 128, 129).
    unsigned body_off[2], halo_off;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (2 * cf + j) * 16 + (lane >> 2);
        int x = xc0 + q;
        if (rep) x = min(max(x, 0), W - 1);
        body_off[j] = (x >= 0 && x < W) ? (unsigned)(x * IPB + (((lane & 3) ^ ((q >> 1) & 3)) << 4)) : OOB;
    }
    {
        // the two halo pixels (ring pixels 128, 129; (128 >> 1) & 3 == (129 >> 1) & 3 == 0: no swizzle) as ONE 4-byte-per-lane DMA over ring pixels
        // 128 .. 131: every lane active (the spare pixels read out of range = zeros).  A 16-byte-per-lane DMA would need `if (lane < 8)` around it,
        // and hipcc's structurizer then duplicated the DMA instructions BEHIND that divergent branch into both arms -- twice the vmcnt of the
        // counted waits, stale rows (round 3: found as 14 wrong pixels either end of a strip).
        const int hp = lane >> 4;
        int x = hp == 0 ? (xc0 - 1) : (xc0 + TW);
        if (rep) x = min(max(x, 0), W - 1);
        halo_off = (hp < 2 && x >= 0 && x < W) ? (unsigned)(x * IPB + (lane & 15) * 4) : OOB;
    }
    auto issue = [&](int t) __attribute__((always_inline)) {        // input row of stage A's step t
        int yi = ys - 2 + t;
        if (rep) yi = min(max(yi, 0), H - 1);
        const unsigned rmask = (yi >= 0 && yi < H) ? 0u : OOB;
        const unsigned row_b = rmask ? 0u : (unsigned)yi * (unsigned)W * IPB;
        char* slot = smem + (t % NRA) * SLOT_A;
#pragma unroll
        for (int c = 0; c < NCA; ++c) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(slot + c * CH_ROW_B + (2 * cf + j) * 1024), 16,
                                                         (int)((row_b + body_off[j] + c * 64) | rmask), 0, 0, 0);
            if (cf == 3)          // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(slot + c * CH_ROW_B + TW * PXB), 4,
                                                         (int)((row_b + halo_off + c * 64) | rmask), 0, 0, 0);
        }
    };
    const int LA = 2 * NCA + (cf == 3 ? NCA : 0);     // DMA instructions per step of this (stage A) wave

    // ---- fragment read offsets: lane (p16, kq) of pixel tile nt at tap column kx reads chunk piece kq of ring pixel 16 nt + p16 + kx - 1 ----
    auto lds_px = [&](int q) __attribute__((always_inline)) { return q * PXB + ((kq ^ ((q >> 1) & 3)) << 4); };
    // The swizzle term of a pixel repeats every 8 pixels, so tile nt's offset is tile 1's + 1024 (nt - 1): an immediate of the ds_read.  Exceptions, each
    // one lane of one tile: the column left of the strip / image (kx = 0, tile 0, p16 = 0) and the one right of it (kx = 2).  Stage A reads the halo
    // pixels 128 / 129 there.  Stage B reads the intermediate, which has no halo: its outermost columns are dropped, and replicate padding replaces
    // the column right of the image's last one, q = qhi + 1, by qhi -- in whichever tile that falls -- and the one left of column 0 by column 0.
    // So: kx = 0 and 1 from one base register each (+ one register for the first tile's exception), kx = 2 from a table of eight.  (A full 3 x 8
    // table cost 24 address VGPRs and the compiler then had none left to read fragments ahead of their MFMAs: 59 % MFMA-busy -> see DESIGN 5.)
    int rd_base[2], rd_first, rd2[8];
    {
        const int qlo = max(0, -xc0), qhi = max(min(TW - 1, W - 1 - xc0), qlo);
        auto fix = [&](int q) __attribute__((always_inline)) {
            if (stg == 0) return q < 0 ? TW : (q >= TW ? TW + 1 : q);
            if (rep) return min(max(q, qlo), qhi);
            return min(max(q, 0), TW - 1);
        };
        rd_base[0] = lds_px(16 + p16 - 1);
        rd_base[1] = lds_px(16 + p16);
        rd_first = lds_px(fix(p16 - 1));
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) rd2[nt] = lds_px(fix(16 * nt + p16 + 1));
    }
    // thin stage B: wave cf covers tiles 2 cf, 2 cf + 1; six offsets (kx, tile), the same clamping rules
    int rdt[3][2];
    if constexpr (THIN) {
        const int qlo = max(0, -xc0), qhi = max(min(TW - 1, W - 1 - xc0), qlo);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = 16 * (2 * cf + j) + p16 + kx - 1;
                rdt[kx][j] = lds_px(rep ? min(max(q, qlo), qhi) : min(max(q, 0), TW - 1));
            }
    }
    auto row_mfma_t = [&](const char* slot, ie_f32x4 (&accN)[2], ie_f32x4 (&accM)[2], ie_f32x4 (&accO)[2]) __attribute__((always_inline)) {
        frag bx[4];
        auto rdf = [&](int g) __attribute__((always_inline)) {         // g = (chunk, kx, tile of the wave)
            const int c = g / 6, kx = (g % 6) >> 1, j = g & 1;
            bx[g & 3] = *(const frag*)(slot + c * CH_ROW_B + rdt[kx][j]);
        };
        rdf(0); rdf(1); rdf(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int c = g / 6, kx = (g % 6) >> 1, j = g & 1;
            if (g + 3 < 12) rdf(g + 3);
            accN[j] = ie_mfma16_raw<T>(wv[c * 9 + 0 * 3 + kx], bx[g & 3], (c == 0 && kx == 0) ? bias4 : accN[j]);
            accM[j] = ie_mfma16_raw<T>(wv[c * 9 + 1 * 3 + kx], bx[g & 3], accM[j]);
            accO[j] = ie_mfma16_raw<T>(wv[c * 9 + 2 * 3 + kx], bx[g & 3], accO[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one row into the three live accumulator sets (ky = 0: the set this row starts takes the bias as its first C operand)
    // The order is pinned (sched_barrier after every fragment's three MFMAs) with the fragment reads three ahead of their use: left to itself the
    // scheduler chained the 144 MFMAs per accumulator and waited lgkmcnt(0) behind every read.
    auto row_mfma = [&](const char* slot, auto nck_tag, ie_f32x4 (&accN)[8], ie_f32x4 (&accM)[8], ie_f32x4 (&accO)[8]) __attribute__((always_inline)) {
        constexpr int NG = decltype(nck_tag)::value * 24;
        frag bx[4];
        auto rdf = [&](int g) __attribute__((always_inline)) {         // g = (chunk, kx, tile), compile-time at every call
            const int c = g / 24, kx = (g % 24) >> 3, nt = g & 7;
            const int off = kx == 2 ? rd2[nt] : ((kx == 0 && nt == 0) ? rd_first : rd_base[kx] + 1024 * (nt - 1));
            bx[g & 3] = *(const frag*)(slot + c * CH_ROW_B + off);
        };
        rdf(0); rdf(1); rdf(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c = g / 24, kx = (g % 24) >> 3, nt = g & 7;
            if (g + 3 < NG) rdf(g + 3);
            accN[nt] = ie_mfma16_raw<T>(wv[c * 9 + 0 * 3 + kx], bx[g & 3], (c == 0 && kx == 0) ? bias4 : accN[nt]);
            accM[nt] = ie_mfma16_raw<T>(wv[c * 9 + 1 * 3 + kx], bx[g & 3], accM[nt]);
            accO[nt] = ie_mfma16_raw<T>(wv[c * 9 + 2 * 3 + kx], bx[g & 3], accO[nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- epilogues: tile pair j (tiles 2j, 2j + 1) of a finished row as 16-byte pieces (lanes l / l + 16 exchange halves) ------------------
    // after the exchange: even kq -> tile 2j, odd kq -> tile 2j + 1; the lane holds couts 16 cf + 8 (kq >> 1) .. + 7 of ONE pixel
    int em_off, pk_off;            // tile pair 0; pair j is 32 pixels on (the swizzles repeat every 8 pixels): + j * 32 * PXB / + j * 32 * OPB
    unsigned em_in = 0;            // bit j: the lane's pixel of pair j is inside the image
    {
        const int q = 16 * (kq & 1) + p16;
        em_off = (cf >> 1) * CH_ROW_B + q * PXB + (((2 * (cf & 1) + (kq >> 1)) ^ ((q >> 1) & 3)) << 4);
        pk_off = q * OPB + (((2 * cf + (kq >> 1)) ^ (q & 7)) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = xc0 + q + 32 * j;
            em_in |= (x >= 0 && x < W) ? (1u << j) : 0u;
        }
    }
    auto put_row = [&](const ie_f32x4 (&acc)[8], char* base, int off0, int stride, unsigned keep) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const ie_u32x2 a = {ie_pack2_relu<T>(acc[2 * j][0], acc[2 * j][1]), ie_pack2_relu<T>(acc[2 * j][2], acc[2 * j][3])};
            const ie_u32x2 b2 = {ie_pack2_relu<T>(acc[2 * j + 1][0], acc[2 * j + 1][1]), ie_pack2_relu<T>(acc[2 * j + 1][2], acc[2 * j + 1][3])};
            const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
            const bool k = (keep >> j) & 1u;
            *(ie_u32x4*)(base + off0 + j * stride) = ie_u32x4{k ? sx[0] : 0u, k ? sy[0] : 0u, k ? sx[1] : 0u, k ? sy[1] : 0u};
        }
    };
    // stage B's stores: instruction i of wave cf = pixels 32 cf + 8 i .. + 7 of the parked row, lane = (pixel, 16-byte piece): 1 KiB of whole lines
    unsigned st_col[4];
    int st_lds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = 32 * cf + 8 * i + (lane >> 3), c = lane & 7, x = xc0 + q;
        st_col[i] = (q >= 1 && q <= TS && x < W) ? (unsigned)(x * OPB + c * 16) : OOB;       // x >= 0 there: xc0 + 1 >= 0
        st_lds[i] = q * OPB + ((c ^ (q & 7)) << 4);
    }

    // POOL: lane (of stage B's 256) owns pieces e = 0, 1 of the pooled row: pooled pixel xp = computed columns 2 xp + 1, 2 xp + 2 (image columns xc0 + 1 + 2 xp, even), piece pc
    int pl_lds[2][2] = {{0, 0}, {0, 0}};
    unsigned pl_col[2] = {OOB, OOB};
    if constexpr (POOL) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int pi = e * 256 + cf * 64 + lane, xp = pi >> 3, pc = pi & 7, q0 = 2 * xp + 1, q1 = 2 * xp + 2;
            pl_lds[e][0] = q0 * OPB + ((pc ^ (q0 & 7)) << 4);
            pl_lds[e][1] = (q1 < TW ? q1 : q0) * OPB + ((pc ^ ((q1 < TW ? q1 : q0) & 7)) << 4);
            const int xcol = ((xc0 + 1) >> 1) + xp;
            pl_col[e] = (q1 <= TS && xcol < (p.W >> 1)) ? (unsigned)(xcol * OPB + pc * 16) : OOB;
        }
    }

    ie_f32x4 a0[8], a1[8], a2[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { a0[nt] = bias4; a1[nt] = bias4; a2[nt] = bias4; }

    if (stg == 0) {
        // Stage A, step t: input row ys - 2 + t; the intermediate's row m = ys - 3 + t is finished (accO) and written into hand-off slot t % NRH
#pragma unroll
        for (int t = 0; t < PF; ++t) issue(t);
        auto step_a = [&](int t, ie_f32x4 (&accN)[8], ie_f32x4 (&accM)[8], ie_f32x4 (&accO)[8]) __attribute__((always_inline)) {
            if (cf == 3) ie_wait_vm_hot<(PF - 1) * 3 * NCA>((PF - 1) * 3 * NCA, [](int n_) __attribute__((always_inline)) { wait_vm_cold(n_); });
            else ie_wait_vm_hot<(PF - 1) * 2 * NCA>((PF - 1) * 2 * NCA, [](int n_) __attribute__((always_inline)) { wait_vm_cold(n_); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            row_mfma(smem + (t % NRA) * SLOT_A, std::integral_constant<int, NCA>{}, accN, accM, accO);
            issue(t + PF);
            const int m = ys - 3 + t;
            const bool row_in = m >= 0 && m < H;
            // zeros: the padding of layer b (rows / columns outside the image) is written as zeros.  replicate: rows outside are not written at all
            // (stage B re-reads the edge row) and columns outside are never read (stage B clamps its column reads)
            if (!rep || row_in) put_row(accO, smem + OFF_H + (t % NRH) * SLOT_B, em_off, 32 * PXB, (rep || row_in) ? (rep ? 0xfu : em_in) : 0u);
        };
        (void)LA;
        for (int t = 0; t < n_steps; t += 3) {
            step_a(t, a0, a1, a2);
            if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0);
        }
    } else if constexpr (THIN) {
        // Thin stage B, step t: the same schedule; accN holds output row ys - 7 + t (finished at step t - 1), which leaves now, straight from the registers
        unsigned st_f32[2], st_16[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 16 * (2 * cf + j) + p16, x = xc0 + q;
            const bool ok = q >= 1 && q <= TS && x < W;
            st_f32[j] = (ok && 4 * kq < p.cout_b) ? (unsigned)(x * p.cout_b * 4 + kq * 16) : OOB;
            st_16[j] = (ok && kq == 0) ? (unsigned)(x * 2 * p.dst16_c) : OOB;
        }
        const size_t f16_b = (size_t)H * W * 2 * p.dst16_c;
        const bool wide16 = p.dst16_c == 8;
        const auto rsD16 = mk(p.dst16 ? (char*)p.dst16 + (size_t)n * f16_b : (char*)p.dst, p.dst16 ? f16_b : 0);
        const bool dual = p.dst16 != nullptr;
        const int act = p.act_b;
        ie_f32x4 t0[2], t1[2], t2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { t0[j] = bias4; t1[j] = bias4; t2[j] = bias4; }
        auto step_bt = [&](int t, ie_f32x4 (&accN)[2], ie_f32x4 (&accM)[2], ie_f32x4 (&accO)[2]) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            {
                const int yo = ys - 7 + t;
                const unsigned rmask = (yo >= ys && yo < ye) ? 0u : OOB;
                const unsigned yrow = rmask ? 0u : (unsigned)yo * (unsigned)W;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    ie_f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {            // the hardware exp2 / rcp forms ie_mconv_nhwc's epilogue uses; `act` is wave-uniform
                        const float a = accN[j][i];
                        v[i] = act == IE_ACT_SIGMOID ? ie_sigmoid_fast(a) : act == IE_ACT_TANH ? ie_tanh_fast(a) : act == IE_ACT_RELU ? fmaxf(a, 0.f) : a;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ie_u32x4, v), rsD, (int)((yrow * (unsigned)(p.cout_b * 4) + st_f32[j]) | rmask), 0, 0);
                    if (dual) {          // wave-uniform
                        const unsigned o16 = (yrow * (unsigned)(2 * p.dst16_c) + st_16[j]) | rmask;
                        if (wide16) __builtin_amdgcn_raw_buffer_store_b128(ie_u32x4{ie_pack2<T>(v[0], v[1]), ie_pack2<T>(v[2], v[3]), 0u, 0u}, rsD16, (int)o16, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b64(ie_u32x2{ie_pack2<T>(v[0], v[1]), ie_pack2<T>(v[2], v[3])}, rsD16, (int)o16, 0, 0);
                    }
                }
            }
            int mc = ys - 5 + t;
            if (rep) mc = min(max(mc, 0), H - 1);
            const int ta = mc - ys + 3;
            row_mfma_t(smem + OFF_H + ((ta < 0 ? 0 : ta) % NRH) * SLOT_B, accN, accM, accO);
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_bt(t, t0, t1, t2);
            if (t + 1 < n_steps) step_bt(t + 1, t2, t0, t1);
            if (t + 2 < n_steps) step_bt(t + 2, t1, t2, t0);
        }
    } else {
        // Stage B, step t: consumes the intermediate's row mc = clamp(ys - 5 + t) (written by stage A at its step mc - ys + 3), finishing output row
        // ys - 6 + t; accN still holds the row finished at step t - 1 (parked now), the row parked at step t - 1 leaves now.
        auto step_b = [&](int t, ie_f32x4 (&accN)[8], ie_f32x4 (&accM)[8], ie_f32x4 (&accO)[8]) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            {
                const int yo = ys - 8 + t;
                const unsigned rmask = (yo >= ys && yo < ye) ? 0u : OOB;
                const unsigned row_b = rmask ? 0u : (unsigned)yo * (unsigned)W * OPB;
                const char* islot = smem + OFF_OUT + ((t + NPARK - 1) % NPARK) * OUT_ROW_B;          // parked at step t - 1
                ie_u32x4 ln[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ln[i] = *(const ie_u32x4*)(islot + st_lds[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(ln[i], rsD, (int)((row_b + st_col[i]) | rmask), 0, 0);
                if constexpr (POOL) {       // rows yo - 1 (parked at step t - 2) and yo make pooled row (yo - 1) / 2 when yo is odd (ys is even)
                    const char* jslot = smem + OFF_OUT + ((t + NPARK - 2) % NPARK) * OUT_ROW_B;
                    const bool ok = !rmask && (yo & 1) && yo - 1 >= ys && (yo >> 1) < Hp;
                    const unsigned pmask = ok ? 0u : OOB;
                    const unsigned prow_b = ok ? (unsigned)(yo >> 1) * (unsigned)Wp * OPB : 0u;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        typedef __attribute__((ext_vector_type(8))) short i16x8;      // ReLU outputs: non-negative floats order as signed integers (-0 loses)
                        const i16x8 v0 = *(const i16x8*)(jslot + pl_lds[e][0]), v1 = *(const i16x8*)(jslot + pl_lds[e][1]);
                        const i16x8 v2 = *(const i16x8*)(islot + pl_lds[e][0]), v3 = *(const i16x8*)(islot + pl_lds[e][1]);
                        const i16x8 pv = __builtin_elementwise_max(__builtin_elementwise_max(v0, v1), __builtin_elementwise_max(v2, v3));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ie_u32x4, pv), rsP, (int)((prow_b + pl_col[e]) | pmask), 0, 0);
                    }
                }
            }
            put_row(accN, smem + OFF_OUT + (t % NPARK) * OUT_ROW_B, pk_off, 32 * OPB, 0xfu);
            int mc = ys - 5 + t;
            if (rep) mc = min(max(mc, 0), H - 1);
            const int ta = mc - ys + 3;                       // stage A's step that wrote it (>= 2)
            row_mfma(smem + OFF_H + ((ta < 0 ? 0 : ta) % NRH) * SLOT_B, std::integral_constant<int, 2>{}, accN, accM, accO);
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_b(t, a0, a1, a2);
            if (t + 1 < n_steps) step_b(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_b(t + 2, a1, a2, a0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) void sconv3x2_f16_c64_c64_c64_relu(PairParams p) { sconv3x2_body<_Float16, 64>(p); }
__global__ __launch_bounds__(512) void sconv3x2_bf16_c64_c64_c64_relu(PairParams p) { sconv3x2_body<__bf16, 64>(p); }
__global__ __launch_bounds__(512) void sconv3x2_f16_c32_c64_c64_relu_pool(PairParams p) { sconv3x2_body<_Float16, 64, 32, true>(p); }
__global__ __launch_bounds__(512) void sconv3x2_bf16_c32_c64_c64_relu_pool(PairParams p) { sconv3x2_body<__bf16, 64, 32, true>(p); }
__global__ __launch_bounds__(512) void sconv3x2_f16_c64_c64_thin_f32(PairParams p) { sconv3x2_body<_Float16, 16>(p); }
__global__ __launch_bounds__(512) void sconv3x2_bf16_c64_c64_thin_f32(PairParams p) { sconv3x2_body<__bf16, 16>(p); }

}  // namespace

// Two Conv3x3(64 -> 64, stride 1, padding 1 zeros | replicate) + bias + ReLU layers in one launch.  w_a / w_b: ie_mconv_pack_weights images of the two layers
// (one 64-channel source each).  src / dst: NHWC, 64 channels, 16-bit.
// cout_b = 64: as above.  cout_b in {4, 8, 12, 16}: the second layer is thin -- dst is fp32 NHWC with cout_b channels, act_b its activation (none / relu /
// sigmoid / tanh), dst16 an optional 16-bit NHWC copy padded to 8 channels (cout_b == 4).
int ie_sconv3x2_launch(const void* src, const void* w_a, const float* bias_a, const void* w_b, const float* bias_b, void* dst, int dtype, int N, int H, int W,
                       int pad_mode, int cout_b, int act_b, void* dst16, int dst16_c, void* pool_dst, hipStream_t st) {
    if (!src || !w_a || !w_b || !dst || N <= 0 || H < 1 || W < 64) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3x2: 64 channels, W >= 64");
    const bool thin = cout_b != 64, pool = pool_dst != nullptr;       // pool: the 32 -> 64 -> 64 form (src has 32 channels)
    if (pool && (thin || H < 2)) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3x2: the pooled form has 64 output channels and H >= 2");
    if (thin && (cout_b < 4 || cout_b > 16 || (cout_b & 3) || (dst16 && cout_b != 4)))
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3x2: thin second layer needs 4, 8, 12 or 16 output channels (4 with a 16-bit copy), got %d", cout_b);
    if (thin && act_b != IE_ACT_NONE && act_b != IE_ACT_RELU && act_b != IE_ACT_SIGMOID && act_b != IE_ACT_TANH)
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3x2: activation %d of the thin layer", act_b);
    if (dtype != IE_F16 && dtype != IE_BF16) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3x2: f16 / bf16");
    if ((double)H * W * 128.0 >= 2147483647.0) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3x2: frame too large for 31-bit offsets");
    PairParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.wa = w_a; p.ba = bias_a; p.wb = w_b; p.bb = bias_b; p.dst = dst; p.N = N; p.H = H; p.W = W; p.replicate = pad_mode == IE_PAD_REPLICATE;
    p.cout_b = cout_b; p.act_b = act_b; p.dst16 = thin ? dst16 : nullptr; p.dst16_c = dst16_c == 4 ? 4 : 8; p.pool = pool_dst;
    p.strips = ie_cdiv(W, TS);
    {
        // row chunks: whole rounds of the chip's 256 workgroup slots (one 8-wave workgroup per CU), few apron rows (8 steps per chunk)
        const int slots = 256;
        const long long base = (long long)p.strips * N;
        int best_c = 1;
        double best = -1.0;
        for (int c = 1; c <= H && c <= 64; ++c) {
            int rpb = ie_cdiv(H, c);
            if (pool) rpb += rpb & 1;                  // pooled rows are made of a chunk's own row pairs
            const int cc = ie_cdiv(H, rpb);
            const long long blocks = base * cc;
            const long long rounds = (blocks + slots - 1) / slots;
            const double score = (double)blocks / (double)(rounds * slots) * ((double)rpb / (double)(rpb + 8));
            if (score > best + 1e-9) { best = score; best_c = cc; }
        }
        p.rows_per_block = ie_cdiv(H, best_c);
        if (pool) p.rows_per_block += p.rows_per_block & 1;
        p.chunks = ie_cdiv(H, p.rows_per_block);
    }
    const long long blocks = (long long)p.strips * p.chunks * N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "ie_sconv3x2: bad grid");
    typedef void (*Fn)(PairParams);
    const int fi = (dtype == IE_F16 ? 0 : 1) + (pool ? 4 : thin ? 2 : 0);
    static const Fn fns[6] = {sconv3x2_f16_c64_c64_c64_relu, sconv3x2_bf16_c64_c64_c64_relu, sconv3x2_f16_c64_c64_thin_f32, sconv3x2_bf16_c64_c64_thin_f32,
                              sconv3x2_f16_c32_c64_c64_relu_pool, sconv3x2_bf16_c32_c64_c64_relu_pool};
    const Fn fn = fns[fi];
    const int lds = pool ? LDS_POOL_B : LDS_B;
    IE_GRANT_LDS(fn, lds);
    void* args[] = {(void*)&p};
    IE_HIP_CHECK(hipLaunchKernel((const void*)fn, dim3((unsigned)blocks), dim3(512), args, lds, st));
    return IE_OK;
}

// ---- C ABI -------------------------------------------------------------------------------------------------------------------------------
IE_EXPORT int ie_mconv_pair_applicable(int cin, int cmid, int cout, int kh, int kw, int stride, int pad, int pad_mode, int act_a, int act_b, int W) {
    static const int off = getenv("IE_SCONV_PAIR") ? !atoi(getenv("IE_SCONV_PAIR")) : 0;       // IE_SCONV_PAIR=0: A/B switch back to two launches
    return (!off && cin == 64 && cmid == 64 && cout == 64 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && (pad_mode == IE_PAD_ZEROS || pad_mode == IE_PAD_REPLICATE) &&
            act_a == IE_ACT_RELU && act_b == IE_ACT_RELU && W >= 64) ? 1 : 0;
}

IE_EXPORT int ie_mconv_pair_nhwc(const void* src, const void* packed_w_a, const float* bias_a, const void* packed_w_b, const float* bias_b, void* dst,
                                 int dtype, int N, int H, int W, int pad_mode, void* stream) {
    if (pad_mode != IE_PAD_ZEROS && pad_mode != IE_PAD_REPLICATE) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pair_nhwc: pad_mode %d", pad_mode);
    return ie_sconv3x2_launch(src, packed_w_a, bias_a, packed_w_b, bias_b, dst, dtype, N, H, W, pad_mode, 64, IE_ACT_RELU, nullptr, 8, nullptr, (hipStream_t)stream);
}

// U-Net-32's second encoder level (unet.py:70-72: MaxPool after DoubleConv(32, 64)): Conv3x3(32 -> 64) + ReLU, Conv3x3(64 -> 64) + ReLU, and the 2x2 max-pooled copy
// of the result, in one launch.  src: NHWC 32 channels; dst: NHWC 64 channels (the skip tensor); pool_dst: NHWC 64 channels, N x floor(H/2) x floor(W/2).
IE_EXPORT int ie_mconv_pair_pool_applicable(int cin, int cmid, int cout, int kh, int kw, int stride, int pad, int pad_mode, int act_a, int act_b, int H, int W) {
    static const int off = getenv("IE_SCONV_PAIR_POOL") ? !atoi(getenv("IE_SCONV_PAIR_POOL")) : 0;
    return (!off && cin == 32 && cmid == 64 && cout == 64 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && (pad_mode == IE_PAD_ZEROS || pad_mode == IE_PAD_REPLICATE) &&
            act_a == IE_ACT_RELU && act_b == IE_ACT_RELU && W >= 64 && H >= 2) ? 1 : 0;
}

IE_EXPORT int ie_mconv_pair_pool_nhwc(const void* src, const void* packed_w_a, const float* bias_a, const void* packed_w_b, const float* bias_b, void* dst, void* pool_dst,
                                      int dtype, int N, int H, int W, int pad_mode, void* stream) {
    if (pad_mode != IE_PAD_ZEROS && pad_mode != IE_PAD_REPLICATE) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pair_pool_nhwc: pad_mode %d", pad_mode);
    if (!pool_dst) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pair_pool_nhwc: pool_dst is null");
    return ie_sconv3x2_launch(src, packed_w_a, bias_a, packed_w_b, bias_b, dst, dtype, N, H, W, pad_mode, 64, IE_ACT_RELU, nullptr, 8, pool_dst, (hipStream_t)stream);
}

IE_EXPORT int ie_mconv_pair_thin_applicable(int cin, int cmid, int cout, int kh, int kw, int stride, int pad, int pad_mode, int act_a, int act_b, int W) {
    static const int off = getenv("IE_SCONV_PAIR") ? !atoi(getenv("IE_SCONV_PAIR")) : 0;
    return (!off && cin == 64 && cmid == 64 && cout >= 4 && cout <= 16 && !(cout & 3) && kh == 3 && kw == 3 && stride == 1 && pad == 1 &&
            (pad_mode == IE_PAD_ZEROS || pad_mode == IE_PAD_REPLICATE) && act_a == IE_ACT_RELU &&
            (act_b == IE_ACT_NONE || act_b == IE_ACT_RELU || act_b == IE_ACT_SIGMOID || act_b == IE_ACT_TANH) && W >= 64) ? 1 : 0;
}

IE_EXPORT int ie_mconv_pair_thin_nhwc(const void* src, const void* packed_w_a, const float* bias_a, const void* packed_w_b, const float* bias_b, float* dst_f32,
                                      void* dst16, int dst16_channels, int dtype, int N, int H, int W, int cout_b, int act_b, int pad_mode, void* stream) {
    if (pad_mode != IE_PAD_ZEROS && pad_mode != IE_PAD_REPLICATE) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pair_thin_nhwc: pad_mode %d", pad_mode);
    if (cout_b == 64) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pair_thin_nhwc: 64 output channels is ie_mconv_pair_nhwc");
    if (dst16 && dst16_channels != 4 && dst16_channels != 8) IE_FAIL(IE_ERR_INVALID, "ie_mconv_pair_thin_nhwc: the 16-bit copy has 4 or 8 channels");
    return ie_sconv3x2_launch(src, packed_w_a, bias_a, packed_w_b, bias_b, dst_f32, dtype, N, H, W, pad_mode, cout_b, act_b, dst16, dst16_channels, nullptr, (hipStream_t)stream);
}
