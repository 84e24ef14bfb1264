// Strip form of the 16-bit matrix-core 3x3 convolution for the narrow layers of the generators (gfx950): Cin = 32 or 64 (one tensor,
// or the virtual concat of two 32-channel tensors), Cout = 32 or 64, stride 1, zero or replicate padding 1 -- the full- and half-resolution layers
// of U-Net-32 / U-Net-64 / RetinexNet (unet.py:43-48, retinexnet/model.py:28-40), where the tiled kernel (ie_conv_mfma.hip) spends as
// long in its per-tile epilogue as in its MFMAs.  Same semantics and the same packed weight image as ie_mconv_nhwc.
//
// It is the Zero-DCE strip machinery (ie_zdce_kernels.hip / ie_zdce_fused.hip, where the 64 -> 32 stage of K345 holds 79 % of the
// MFMA pipe): a workgroup marches down a 128-column strip; ALL weights of the wave's 16 output channels stay in registers for the
// whole kernel (9 taps x Cin/32 fragments = 36 or 72 VGPRs); every input row is DMA'd once into an LDS ring (buffer_load ... lds,
// the descriptor's range check supplies the zero padding) and each ds_read_b128 pixel fragment feeds three MFMAs (ky = 0, 1, 2) into
// three rotating accumulator sets, so an output row is finished -- and stored -- every step instead of a whole tile at the end.
// wave = (cout fragment, column half): Cout / 16 x 2 waves per workgroup (8 waves for 64 output channels).
#include "ie_conv_internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int TW = 128;
constexpr int ROW_PX = TW + 2;
constexpr int PXB = 64;                    // bytes of one 32-channel chunk of a pixel
constexpr int SRC_ROW_B = ROW_PX * PXB;    // one ring row of one chunk
constexpr unsigned OOB = 0x80000000u;
// ring slots (input rows resident in LDS): 4, or 3 with four K chunks (LDS: 4 chunks x 8.3 KB per row) and for 64 -> 32 channels, whose
// four-wave workgroup then fits twice on a CU (66 KB instead of 83 KB of LDS: one wave per SIMD is too few to hide the row barrier)
#ifndef IE_SC_EARLY_ST
#define IE_SC_EARLY_ST 1
#endif
#ifndef IE_SC_FIRST_EARLY
#define IE_SC_FIRST_EARLY 1
#endif
#ifndef IE_SC_EARLY_ALL
#define IE_SC_EARLY_ALL 0
#endif
#ifndef IE_SC_AHEAD4
#define IE_SC_AHEAD4 2
#endif
#ifndef IE_SC_PINNED
#define IE_SC_PINNED 1      // explicit read-ahead + pinned MFMA order in the strip kernels' row loop (0: the sched_group_barrier pipeline of rounds 1 - 2)
#endif
#ifndef IE_SC_NR_C64O32
#define IE_SC_NR_C64O32 3
#endif
#ifndef IE_SC_RING4
#define IE_SC_RING4 2        // four-chunk kernels: bias through LDS + a pinned fragment ring read IE_SC_RING4 fragments ahead (0: the round-2 pipeline, one fragment register)
#endif
template <int NCH, int NCF> struct SGeom { static constexpr int NR = NCH == 4 ? 3 : ((NCH == 2 && NCF == 2) ? IE_SC_NR_C64O32 : 4), PF = NR - 1; };

struct SConvParams {
    const void* src[2];      // the one or two source tensors (virtual concat)
    int tps[2];              // their pixel strides in bytes
    int th[2], tw[2];        // their stored sizes: (H, W), or smaller = read through a nearest-neighbour up-sampling to (H, W)
    int tsel[4], coff[4];    // K chunk c: which tensor, and the chunk's byte offset inside a pixel
    const void* wpk;         // ie_mconv_pack_weights image: [chunk][tap][16 NCF rows][32 k], 16-byte pieces swizzled by (row >> 1) & 3
    const float* bias;       // [Cout] or null
    void* dst;               // NHWC T, Cout channels
    void* pool;              // POOL kernels: the 2x2 max-pooled copy of dst, NHWC T [N, H/2, W/2, Cout] (nn.MaxPool2d(2) behind the conv: unet.py:70-72)
    // TAIL kernels: a 1x1 conv (tail_c <= 16 output channels, optional clamp) of the ReLU'd rows while they sit in LDS; only ITS result leaves,
    // NCHW fp32 [N, tail_c, H, W] -- the U-Net head `outc` behind the last DoubleConv (unet.py:186-195).  dst is not written.
    const void* tail_w;      // [16][Cout] T, rows >= tail_c zero
    const float* tail_b;     // [16]
    float* tail_dst;
    int tail_c, tail_clamp;
    float tail_lo, tail_hi;
    int N, H, W, rows_per_block, strips, chunks;
    // FIRST kernels: the conv's input rows are not read from HBM but computed in the kernel, one row ahead, by the layer in front of it --
    // the network's first conv (8-channel image -> 32 NCH channels, 3x3, zero padding, bias + ReLU; unet.py:43-48 `inc`): the first DoubleConv
    // in one launch, its intermediate never in HBM.
    const void* img;         // NHWC 8-channel image (16 bytes per pixel)
    const void* w1;          // ie_mconv_pack_weights image of the first conv (thin form: [ky][32 NCH rows][k = kx * 8 + channel])
    const float* b1;         // [32 NCH] or null
    int sub;                 // 1, or 2 = stride 2: every row / column is computed, the even ones are stored (Ho = ceil(H/2), Wo = ceil(W/2))
    int replicate;           // padding_mode="replicate": out-of-image rows / columns are clamped to the edge instead of reading zeros
    float slope;
};

__device__ __forceinline__ void wait_vm(int n) {
#define IE_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
    IE_W(1) IE_W(2) IE_W(3) IE_W(4) IE_W(5) IE_W(6) IE_W(7) IE_W(8) IE_W(9) IE_W(10) IE_W(11) IE_W(12) IE_W(13) IE_W(14) IE_W(15) IE_W(16)
    IE_W(17) IE_W(18) IE_W(19) IE_W(20) IE_W(21) IE_W(22) IE_W(23) IE_W(24) IE_W(25) IE_W(26) IE_W(27) IE_W(28) IE_W(29) IE_W(30)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef IE_W
}

template <typename T, int NCH, int NCF, int ACT, bool POOL = false, bool TAIL = false, bool SUB2 = false, bool FIRST = false>
__device__ __forceinline__ void sconv3_body(const SConvParams& p) {
    using frag = ie_u32x4;
    constexpr int NR = SGeom<NCH, NCF>::NR, PF = SGeom<NCH, NCF>::PF;
    constexpr int NW = NCF * 2;                   // waves
    constexpr int PPW = 8 / NW > 0 ? 8 / NW : 1;  // body DMA pieces (16 pixels x 64 B) per wave, chunk and row
    constexpr int SLOT_B = SRC_ROW_B * NCH;
    constexpr int OPB = 32 * NCF;                 // output bytes per pixel
    // Finished rows are parked in LDS as whole pixel lines and leave one step later, 1 KiB of contiguous lines per store instruction: a
    // wave owns 16 of a pixel's 32 / 64 output channels, and 32-byte pieces of a line cost the write path as much as the whole line
    // (profiles/r02/k12_experiments_2.log).  [px][OPB bytes], the 16-byte pieces of pixel q XOR-swizzled by q (bank spread).
    constexpr int OUT_ROW_B = TW * OPB;
    constexpr int OFF_OUT = NR * SLOT_B;
    constexpr int NPC = OPB / 16;                 // 16-byte pieces per pixel (4 or 8)
    constexpr int PPI = 64 / NPC;                 // pixels per store instruction (16 or 8)
    constexpr int SPW = (TW / PPI) / NW;          // store instructions per wave and row (2)
    // POOL: a third parked row, so that rows 2k and 2k+1 are both in LDS when the later one leaves; every lane then owns one 16-byte
    // piece of the 64 pooled pixels (NW x 64 lanes = 64 pixels x NPC pieces) and the pooled row leaves as whole lines too.  The
    // activation is ReLU, so the packed 16-bit values order like unsigned integers.
    static_assert(!POOL || ACT == IE_ACT_RELU, "the fused max-pool compares non-negative values as integers");
    static_assert(NW * 64 == (TW / 2) * NPC, "one pooled piece per lane");
    static_assert(!TAIL || (ACT == IE_ACT_RELU && !POOL), "the 1x1 tail follows a ReLU layer");
    constexpr int TKB = NCF / 2;                  // TAIL: 32-channel blocks of the 1x1 conv's K (= this conv's Cout)
    constexpr int TT = 8 / NW;                    // TAIL: 16-pixel tiles of the strip per wave; TT * TKB == 2
    constexpr int NST = TAIL ? TT : SPW + (POOL ? 1 : 0);       // store instructions a wave issues per step
    // FIRST: image rows (132 pixels x 16 bytes, three 1-KiB DMA instructions) in their own ring; image row k = ys - 2 + k; the input row of
    // step it + 1 (x1 row ys + it) comes from image rows k = it + 1 .. it + 3 while the DMA of row it + 3 + PF is in flight
    constexpr int NF1 = 2 * NCH, IMG_SLOT_B = 3072, NRI = PF + 3;
    constexpr int OFF_IMG = OFF_OUT + (POOL ? 3 : 2) * OUT_ROW_B;
    static_assert(!FIRST || (NW == 2 * NF1 && ACT == IE_ACT_RELU && !SUB2 && !TAIL), "FIRST: one conv1 fragment and half of the nine tiles per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int ph = wave & 1, cf = wave >> 1;

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = b;
    const int x0 = strip * TW;
    const int H = p.H, W = p.W;
    const int ys = chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, H);
    const int n_it = (ye - ys) + 2;               // input rows ys-1 .. ye

    // ---- buffer descriptors, one frame each -----------------------------------------------------------------------------
    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    };
    const size_t fb0 = (size_t)p.th[0] * p.tw[0] * p.tps[0], fb1 = (size_t)p.th[1] * p.tw[1] * p.tps[1];
    const auto rs0 = mk((const char*)p.src[0] + (size_t)n * fb0, fb0);
    const auto rs1 = mk((const char*)p.src[1] + (size_t)n * fb1, fb1);
    const int sub = p.sub, Wo = (W + sub - 1) / sub;
    const size_t fbo = (size_t)((H + sub - 1) / sub) * Wo * OPB;
    const auto rsd = mk((char*)p.dst + (size_t)n * fbo, fbo);
    const int Hp = H >> 1, Wp = W >> 1;
    const size_t fbp = (size_t)Hp * Wp * OPB;
    const auto rsp = mk(POOL ? (char*)p.pool + (size_t)n * fbp : (char*)p.dst, POOL ? fbp : 0);
    const size_t fbt = (size_t)p.tail_c * H * W * 4;
    const auto rst = mk(TAIL ? (char*)p.tail_dst + (size_t)n * fbt : (char*)p.dst, TAIL ? fbt : 0);

    // ---- weights of this wave's 16 output channels into registers ---------------------------------------------------------
    frag wA[NCH * 9];
    {
        const char* wp = (const char*)p.wpk;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                wA[c * 9 + t] = *(const frag*)(wp + ((size_t)(c * 9 + t) * (16 * NCF) + cf * 16 + p16) * 64 + ((kq ^ ((p16 >> 1) & 3)) << 4));
    }
    // Four K chunks (144 weight + 48 accumulator VGPRs): the bias does not ride in registers as the first C operand but waits in LDS and joins when a row is
    // parked -- the four registers it frees are the second slot of the fragment ring below (IE_SC_RING4).
    constexpr bool LDS_BIAS = NCH == 4 && IE_SC_RING4 > 0 && !FIRST && !SUB2;
    constexpr int OFF_BIAS = OFF_IMG;          // (no image ring in these kernels) 16 NCF floats behind the parked rows
    ie_f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LDS_BIAS) {
        if (tid < 16 * NCF) *(float*)(smem + OFF_BIAS + tid * 4) = p.bias ? p.bias[tid] : 0.f;      // published by the first step's barrier, read from step 2 on
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = p.bias ? p.bias[16 * cf + 4 * kq + i] : 0.f;
    }

    // ---- DMA geometry: piece m = 16 pixels x 64 B of one chunk; 16-byte position (lane & 3) of ring pixel q holds channel piece
    //      (lane & 3) ^ ((q >> 1) & 3), so the B-operand reads are bank-conflict free at every tap offset ------------------------
    // F.interpolate(size=(H, W)) in front of the conv, folded into the addressing: source column = min(floor(x * in / out), in - 1), per tensor
    const float fx0 = (float)p.tw[0] / (float)W, fx1 = (float)p.tw[1] / (float)W, fy0 = (float)p.th[0] / (float)H, fy1 = (float)p.th[1] / (float)H;
    auto map_x = [&](int x, int t) __attribute__((always_inline)) {
        return x < 0 ? -1 : (p.tw[t] == W ? x : min((int)floorf(x * (t ? fx1 : fx0)), p.tw[t] - 1));
    };
    int px_q[2][PPW];       // source pixel of this lane in piece j, per tensor (-1: outside the image)
    int px_s[PPW];          // source 16-byte piece
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = (PPW * wave + j) * 16 + (lane >> 2);
        const int x = (x0 + q < W) ? x0 + q : (p.replicate ? W - 1 : -1);
        px_q[0][j] = map_x(x, 0); px_q[1][j] = map_x(x, 1);
        px_s[j] = ((lane & 3) ^ ((q >> 1) & 3)) * 16;
    }
    int halo_x[2];
    {
        int x = (lane < 4) ? (x0 - 1) : (x0 + TW);
        if (p.replicate) x = min(max(x, 0), W - 1);
        x = (lane < 8 && x >= 0 && x < W) ? x : -1;
        halo_x[0] = map_x(x, 0); halo_x[1] = map_x(x, 1);
    }
    auto issue_row = [&](int it) __attribute__((always_inline)) {
        int yi = ys - 1 + it;
        if (p.replicate) yi = min(max(yi, 0), H - 1);
        const bool row_ok = (yi >= 0) && (yi < H) && (it < n_it);
        char* slot = smem + (it % NR) * SLOT_B;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool t0 = p.tsel[c] == 0;
            const int ti = t0 ? 0 : 1;
            const unsigned ps = (unsigned)p.tps[ti], co = (unsigned)p.coff[c];
            const int ysrc = p.th[ti] == H ? yi : min((int)floorf(yi * (ti ? fy1 : fy0)), p.th[ti] - 1);
            const unsigned row_b = (unsigned)ysrc * (unsigned)p.tw[ti] * ps;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                if (PPW * wave + j >= 8) break;
                const unsigned off = (row_ok && px_q[ti][j] >= 0) ? row_b + (unsigned)px_q[ti][j] * ps + co + (unsigned)px_s[j] : OOB;
                char* dstb = slot + c * SRC_ROW_B + (PPW * wave + j) * 1024;
                if (t0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)dstb, 16, (int)off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)dstb, 16, (int)off, 0, 0, 0);
            }
            if (wave == NW - 1) {          // halo columns: (128 >> 1) & 3 == (129 >> 1) & 3 == 0, position == channel piece
                const unsigned off = (row_ok && halo_x[ti] >= 0) ? row_b + (unsigned)halo_x[ti] * ps + co + (unsigned)((lane & 3) * 16) : OOB;
                char* dstb = slot + c * SRC_ROW_B + TW * PXB;
                if (lane < 8) {
                    if (t0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)dstb, 16, (int)off, 0, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)dstb, 16, (int)off, 0, 0, 0);
                }
            }
        }
    };
    const int L = FIRST ? (wave < 3 ? 1 : 0) : (PPW * wave < 8 ? PPW : 0) * NCH + (wave == NW - 1 ? NCH : 0);
    const int wait_n = (PF - 1) * L;

    // ---- FIRST: the image ring and the first conv --------------------------------------------------------------------------------
    const size_t fbi = (size_t)H * W * 16;
    const auto rsi = mk(FIRST ? (const char*)p.img + (size_t)n * fbi : (const char*)p.dst, FIRST ? fbi : 0);
    auto issue_img = [&](int k) __attribute__((always_inline)) {
        if (wave >= 3) return;
        const int m = ys - 2 + k, j = 64 * wave + lane, x = x0 - 2 + j;        // image row, pixel of the slot, image column
        const bool ok = m >= 0 && m < H && j < 132 && x >= 0 && x < W;
        const unsigned off = ok ? (unsigned)((m * W + x) * 16) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsi, (__attribute__((address_space(3))) void*)(smem + OFF_IMG + (k % NRI) * IMG_SLOT_B + wave * 1024), 16,
                                                 (int)off, 0, 0, 0);
    };
    // this wave: conv1 cout fragment f1 (16 of the 32 NCH channels), tiles 5 tg1 .. 5 tg1 + 4 of the nine 16-pixel tiles of a ring row (eight of
    // the body, one with the two halo pixels).  K = 32 = (kx 0..3) x 8 channels (the fourth tap has zero weights): lane (p16, kq) reads the 16
    // bytes of image pixel x - 1 + kq, one MFMA per kernel row -- the thin kernel's contraction (same order: ky 0, 1, 2; bias and ReLU behind it)
    const int f1 = wave % NF1, tg1 = wave / NF1;
    frag w1f[3];
    float b1v[4] = {0.f, 0.f, 0.f, 0.f};
    int img_off[5], ring_off[5];
    unsigned colok = 0;
    if constexpr (FIRST) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
            w1f[ky] = *(const frag*)((const char*)p.w1 + ((size_t)(ky * 16 * NF1 + 16 * f1 + p16)) * 64 + ((kq ^ ((p16 >> 1) & 3)) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) b1v[i] = p.b1 ? p.b1[16 * f1 + 4 * kq + i] : 0.f;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int t = 5 * tg1 + u;
            const int q = t < 8 ? 16 * t + p16 : 128 + (p16 & 1);                  // ring pixel (tile 8: lanes p16 >= 2 shadow the two halo pixels, no write)
            const int rel = q < 128 ? q : (q == 128 ? -1 : 128);                     // column relative to x0
            img_off[u] = (rel + 1 + kq) * 16;
            const int c = f1 >> 1, piece = 2 * (f1 & 1) + (kq >> 1);
            ring_off[u] = c * SRC_ROW_B + q * PXB + ((piece ^ ((q >> 1) & 3)) << 4) + (kq & 1) * 8;
            const int x = x0 + rel;
            if (x >= 0 && x < W && t < 9 && (t < 8 || p16 < 2)) colok |= 1u << u;
        }
    }
    auto conv1_row = [&](int it) __attribute__((always_inline)) {          // x1 row ys + it -> ring slot (it + 1) % NR, from image rows k = it + 1 .. it + 3
        const int r = ys + it;
        const bool rv = r >= 0 && r < H;                                     // rows outside the image are the second conv's zero padding, not conv1 of padding
        char* slot = smem + ((it + 1) % NR) * SLOT_B;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (5 * tg1 + u >= 9) break;
            ie_f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const frag bx = *(const frag*)(smem + OFF_IMG + ((it + 1 + ky) % NRI) * IMG_SLOT_B + img_off[u]);
                a = ie_mfma16_raw<T>(w1f[ky], bx, a);
            }
            ie_u32x2 o = {ie_pack2_relu<T>(a[0] + b1v[0], a[1] + b1v[1]), ie_pack2_relu<T>(a[2] + b1v[2], a[3] + b1v[3])};
            if (!rv || !((colok >> u) & 1u)) o = ie_u32x2{0u, 0u};
            if (5 * tg1 + u < 8 || p16 < 2) *(ie_u32x2*)(slot + ring_off[u]) = o;
        }
    };

    // ---- LDS read offsets of the B operand ------------------------------------------------------------------------------------
    auto lds_px = [&](int q) __attribute__((always_inline)) { return q * PXB + ((kq ^ ((q >> 1) & 3)) << 4); };
    int rd_base[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) rd_base[kx] = lds_px(64 * ph + 16 + p16 + kx - 1) - 1024;
    int rd_first, rd_last;
    {
        const int xf = 64 * ph + p16 - 1;
        rd_first = lds_px(xf < 0 ? TW : xf);
        const int xl = 64 * ph + 48 + p16 + 1;
        rd_last = lds_px(xl >= TW ? TW + 1 : xl);
    }

    // ---- epilogue -----------------------------------------------------------------------------------------------------------------
    auto swz = [&](int q) __attribute__((always_inline)) { return NPC == 8 ? (q & 7) : ((q >> 1) & 3); };
    // lanes l / l+16 exchange halves (v_permlane16_swap), so a lane parks 16 bytes (8 output channels) of ONE pixel: even kq -> tile 2j,
    // odd kq -> tile 2j+1; ds_write_b128 serves 8 lanes per LDS cycle = 8 consecutive pixels x 16 bytes, every bank once
    int park_off0;              // j = 1 is 32 pixels on: the same swizzle, a constant 32 * OPB bytes away (kept out of a register)
    {
        const int q = 64 * ph + 16 * (kq & 1) + p16, c = 2 * cf + (kq >> 1);
        park_off0 = q * OPB + ((c ^ swz(q)) << 4);
    }
    unsigned st_col[SPW];       // store instruction j of this wave: PPI pixels x OPB bytes, lane = (pixel, 16-byte piece)
    int st_lds0;                // instruction j reads PPI pixels on: the same swizzle (PPI is 8 or 16), a constant PPI * OPB bytes away
    static_assert(SPW == 2 && (PPI == 8 || PPI == 16), "store geometry");
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        const int q = (SPW * wave + j) * PPI + lane / NPC, c = lane % NPC, xo = x0 + q;
        st_col[j] = (xo < W && (sub == 1 || !(xo & 1))) ? (unsigned)((xo / sub) * OPB + c * 16) : OOB;
        if (j == 0) st_lds0 = q * OPB + ((c ^ swz(q)) << 4);
    }
    // POOL: this lane's piece of the pooled row: pooled pixel xp (source pixels 2xp, 2xp+1 of the strip), 16-byte piece pc
    const int pl_xp = tid / NPC, pl_c = tid % NPC;
    const int pl_lds0 = (2 * pl_xp) * OPB + ((pl_c ^ swz(2 * pl_xp)) << 4), pl_lds1 = (2 * pl_xp + 1) * OPB + ((pl_c ^ swz(2 * pl_xp + 1)) << 4);
    const unsigned pl_col = ((x0 >> 1) + pl_xp < Wp) ? (unsigned)(((x0 >> 1) + pl_xp) * OPB + pl_c * 16) : OOB;
    // TAIL: the parked row is the A operand (a lane = one pixel, 8 channels), the 1x1 weights the B operand (a lane = one output channel):
    // D leaves with lane (p16 = output channel, kq) holding pixels 4kq .. 4kq+3 of the tile -- 16 contiguous bytes of an NCHW plane
    frag tW[TKB];
    float tl_bias = 0.f;
    int tl_lds[TT][TKB];
    unsigned tl_col[TT];
    if constexpr (TAIL) {
#pragma unroll
        for (int bk = 0; bk < TKB; ++bk) tW[bk] = *(const frag*)((const char*)p.tail_w + p16 * OPB + bk * 64 + kq * 16);
        tl_bias = p.tail_b[p16];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int q = 16 * (TT * wave + t) + p16;
#pragma unroll
            for (int bk = 0; bk < TKB; ++bk) tl_lds[t][bk] = q * OPB + (((4 * bk + kq) ^ swz(q)) << 4);
            const int xo = x0 + 16 * (TT * wave + t) + 4 * kq;          // W % 4 == 0: the four pixels are inside or outside together
            tl_col[t] = (p16 < p.tail_c && xo < W) ? (unsigned)((p16 * H * W + xo) * 4) : OOB;
        }
    }
    auto park = [&](const ie_f32x4 (&acc_)[4], char* oslot) __attribute__((always_inline)) {
        ie_u32x2 pk[4];
        ie_f32x4 acc[4];
        if constexpr (LDS_BIAS) {
            const ie_f32x4 bv = *(const ie_f32x4*)(smem + OFF_BIAS + (16 * cf + 4 * kq) * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = acc_[nt] + bv;
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = acc_[nt];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if constexpr (ACT == IE_ACT_RELU) {
                pk[nt][0] = ie_pack2_relu<T>(acc[nt][0], acc[nt][1]);
                pk[nt][1] = ie_pack2_relu<T>(acc[nt][2], acc[nt][3]);
            } else if constexpr (ACT == IE_ACT_LRELU) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[nt][i] > 0.f ? acc[nt][i] : acc[nt][i] * p.slope;
                pk[nt][0] = ie_pack2<T>(v[0], v[1]);
                pk[nt][1] = ie_pack2<T>(v[2], v[3]);
            } else {
                pk[nt][0] = ie_pack2<T>(acc[nt][0], acc[nt][1]);
                pk[nt][1] = ie_pack2<T>(acc[nt][2], acc[nt][3]);
            }
        }
#ifdef IE_PARK_B64       // A/B: 8-byte pieces straight from the accumulators (two-way bank conflicts between pixels q and q+8)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int q = 64 * ph + 16 * nt + p16, c = 2 * cf + (kq >> 1);
            *(ie_u32x2*)(oslot + q * OPB + ((c ^ swz(q)) << 4) + (kq & 1) * 8) = pk[nt];
        }
#else
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const auto sx = __builtin_amdgcn_permlane16_swap(pk[2 * j][0], pk[2 * j + 1][0], false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(pk[2 * j][1], pk[2 * j + 1][1], false, false);
            *(ie_u32x4*)(oslot + park_off0 + j * 32 * OPB) = ie_u32x4{sx[0], sy[0], sx[1], sy[1]};
        }
#endif
    };

    // ---- one input row into the three live accumulator sets (output rows yi+1, yi, yi-1) -------------------------------------------
    auto row_mfma = [&](int it, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
        const char* slot = smem + (it % NR) * SLOT_B;
        if constexpr (SUB2) {
            // stride 2: only the even output rows leave.  An odd input row feeds the even rows above and below it (accN, accO), an even one
            // its own row (accM): 1.5 of the 3 MFMAs per fragment on average; the skipped sets hold odd rows that are parked but never stored.
            const bool odd = ((ys - 1 + it) & 1) != 0;
            if (odd) {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int off = (kx == 0 && nt == 0) ? rd_first : (kx == 2 && nt == 3) ? rd_last : rd_base[kx] + 1024 * nt;
                            const frag bx = *(const frag*)(slot + c * SRC_ROW_B + off);
                            accN[nt] = ie_mfma16_raw<T>(wA[c * 9 + 0 * 3 + kx], bx, (c == 0 && kx == 0) ? bias4 : accN[nt]);
                            accO[nt] = ie_mfma16_raw<T>(wA[c * 9 + 2 * 3 + kx], bx, accO[nt]);
                        }
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int f = 0; f < NCH * 12 - 3; ++f) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int off = (kx == 0 && nt == 0) ? rd_first : (kx == 2 && nt == 3) ? rd_last : rd_base[kx] + 1024 * nt;
                            const frag bx = *(const frag*)(slot + c * SRC_ROW_B + off);
                            accM[nt] = ie_mfma16_raw<T>(wA[c * 9 + 1 * 3 + kx], bx, accM[nt]);
                        }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int f = 0; f < NCH * 12 - 4; ++f) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            return;
        }
        if constexpr (IE_SC_PINNED && NCH < 4) {
        // Pinned order (round 3): the fragment reads run AHEAD fragments before their three MFMAs, through a small ring of fragment registers, and a
        // sched_barrier after every fragment keeps it that way.  The sched_group_barrier pipeline below asked for the same thing, but the listing
        // showed `ds_read_b128 ; s_waitcnt lgkmcnt(0) ; 3 x v_mfma` into ONE reused register quad for every fragment of every strip kernel: the full
        // LDS latency in front of each MFMA triplet (53 % MFMA-busy on 64 -> 64).
        {
            constexpr int AHEAD = 3;
            constexpr int NG = NCH * 12;
            frag bx[AHEAD + 1];
            auto rdf = [&](int g) __attribute__((always_inline)) {       // g = (chunk, kx, tile): compile-time at every call
                const int c = g / 12, kx = (g % 12) >> 2, nt = g & 3;
                const int off = (kx == 0 && nt == 0) ? rd_first : (kx == 2 && nt == 3) ? rd_last : rd_base[kx] + 1024 * nt;
                bx[g % (AHEAD + 1)] = *(const frag*)(slot + c * SRC_ROW_B + off);
            };
#pragma unroll
            for (int g = 0; g < AHEAD; ++g) rdf(g);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int c = g / 12, kx = (g % 12) >> 2, nt = g & 3;
                if (g + AHEAD < NG) rdf(g + AHEAD);
                const frag b_ = bx[g % (AHEAD + 1)];
                accN[nt] = ie_mfma16_raw<T>(wA[c * 9 + 0 * 3 + kx], b_, (c == 0 && kx == 0) ? bias4 : accN[nt]);
                accM[nt] = ie_mfma16_raw<T>(wA[c * 9 + 1 * 3 + kx], b_, accM[nt]);
                accO[nt] = ie_mfma16_raw<T>(wA[c * 9 + 2 * 3 + kx], b_, accO[nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        } else if constexpr (LDS_BIAS) {      // four K chunks: a ring of TWO fragment registers, the read one fragment ahead of its three MFMAs, order pinned
            constexpr int NG = NCH * 12, RA = IE_SC_RING4;       // RA fragments ahead, RA + 1 ring slots
            frag bx[RA + 1];
            auto rdf = [&](int g) __attribute__((always_inline)) {
                const int c = g / 12, kx = (g % 12) >> 2, nt = g & 3;
                const int off = (kx == 0 && nt == 0) ? rd_first : (kx == 2 && nt == 3) ? rd_last : rd_base[kx] + 1024 * nt;
                bx[g % (RA + 1)] = *(const frag*)(slot + c * SRC_ROW_B + off);
            };
#pragma unroll
            for (int g = 0; g < RA; ++g) rdf(g);
            __builtin_amdgcn_sched_barrier(0);
            const ie_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int c = g / 12, kx = (g % 12) >> 2, nt = g & 3;
                if (g + RA < NG) rdf(g + RA);
                const frag b_ = bx[g % (RA + 1)];
                accN[nt] = ie_mfma16_raw<T>(wA[c * 9 + 0 * 3 + kx], b_, (c == 0 && kx == 0) ? zero4 : accN[nt]);
                accM[nt] = ie_mfma16_raw<T>(wA[c * 9 + 1 * 3 + kx], b_, accM[nt]);
                accO[nt] = ie_mfma16_raw<T>(wA[c * 9 + 2 * 3 + kx], b_, accO[nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {      // four K chunks without IE_SC_RING4: no registers for a fragment ring (256 VGPRs, 2 spilled with one fragment of look-ahead): the round-2 pipeline
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int off = (kx == 0 && nt == 0) ? rd_first : (kx == 2 && nt == 3) ? rd_last : rd_base[kx] + 1024 * nt;
                    const frag bx = *(const frag*)(slot + c * SRC_ROW_B + off);
                    accN[nt] = ie_mfma16_raw<T>(wA[c * 9 + 0 * 3 + kx], bx, (c == 0 && kx == 0) ? bias4 : accN[nt]);
                    accM[nt] = ie_mfma16_raw<T>(wA[c * 9 + 1 * 3 + kx], bx, accM[nt]);
                    accO[nt] = ie_mfma16_raw<T>(wA[c * 9 + 2 * 3 + kx], bx, accO[nt]);
                }
        // pixel fragments read ahead of the MFMAs that consume them: 3, or 2 with four K chunks (4 VGPRs each; that kernel has none to spare)
        constexpr int AHEAD = NCH == 4 ? IE_SC_AHEAD4 : 3;
        __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
        for (int f = 0; f < NCH * 12 - AHEAD; ++f) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * AHEAD, 0);
        }
    };
    auto step = [&](int it, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
        // ops younger than row `it`'s DMA: the (PF-1) L loads of the rows behind it and the SPW (possibly out-of-range) stores every step issues
        // steady-state counts per wave class (a regular wave; the wave that also loads the halo columns): compile-time, tested first
        constexpr int HOT0 = FIRST ? (PF - 1) + NST * PF : (PF - 1) * PPW * NCH + NST * PF;
        constexpr int HOT1 = FIRST ? -1 : (PF - 1) * (PPW * NCH + NCH) + NST * PF;
        if (L) ie_wait_vm_hot<HOT0, HOT1>(wait_n + NST * (it < PF ? it : PF), [](int n_) __attribute__((always_inline)) { wait_vm(n_); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's parked pieces of the previous step
        __builtin_amdgcn_s_barrier();
        if constexpr (FIRST) issue_img(it + 3 + PF);
        else issue_row(it + PF);
        // the row parked in the previous step (finished two steps ago)
        const int yo = ys - 4 + it;
        const unsigned rmask = ((yo >= ys) && (yo < ye) && (sub == 1 || !(yo & 1))) ? 0u : OOB;
        const unsigned row_b = rmask ? 0u : (unsigned)(yo / sub) * (unsigned)Wo * OPB;
        const char* islot = smem + OFF_OUT + (POOL ? (it + 2) % 3 : (it + 1) & 1) * OUT_ROW_B;
        ie_u32x4 ln[SPW];
        frag ta[TT][TKB];
        if constexpr (TAIL) {
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int bk = 0; bk < TKB; ++bk) ta[t][bk] = *(const frag*)(islot + tl_lds[t][bk]);
        } else {
#pragma unroll
            for (int j = 0; j < SPW; ++j) ln[j] = *(const ie_u32x4*)(islot + st_lds0 + j * PPI * OPB);
        }
        // POOL: rows yo-1 (parked two steps ago) and yo make pooled row (yo-1)/2 when yo is odd (ys is even)
        ie_u16x8 pv = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned pmask = OOB, prow_b = 0u;
        if constexpr (POOL) {
            if ((yo & 1) && (yo - 1 >= ys) && (yo < ye)) {
                const char* jslot = smem + OFF_OUT + ((it + 1) % 3) * OUT_ROW_B;
                const ie_u16x8 v0 = *(const ie_u16x8*)(jslot + pl_lds0), v1 = *(const ie_u16x8*)(jslot + pl_lds1);
                const ie_u16x8 v2 = *(const ie_u16x8*)(islot + pl_lds0), v3 = *(const ie_u16x8*)(islot + pl_lds1);
                pv = __builtin_elementwise_max(__builtin_elementwise_max(v0, v1), __builtin_elementwise_max(v2, v3));
                pmask = 0u;
                prow_b = (unsigned)(yo >> 1) * (unsigned)Wp * OPB;
            }
        }
        // the row finished in the previous step is parked before its accumulator set is restarted by this step's MFMAs
        park(accN, smem + OFF_OUT + (POOL ? it % 3 : it & 1) * OUT_ROW_B);
        // four K chunks: 144 weight + 48 accumulator VGPRs leave no room to carry the row's 8 store registers across the MFMAs (they spilled)
        if constexpr (FIRST && IE_SC_FIRST_EARLY) {      // next step's input row, ahead of this row's MFMAs (its LDS-read -> MFMA -> write chains then run under them)
            __builtin_amdgcn_sched_barrier(0);
            conv1_row(it);
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr bool EARLY_ST = IE_SC_EARLY_ST && (NCH == 4 || IE_SC_EARLY_ALL) && !TAIL;
        if constexpr (EARLY_ST) {
#pragma unroll
            for (int j = 0; j < SPW; ++j) __builtin_amdgcn_raw_buffer_store_b128(ln[j], rsd, (int)((row_b + st_col[j]) | rmask), 0, 0);
        }
        row_mfma(it, accN, accM, accO);
        if constexpr (EARLY_ST) {
        } else if constexpr (TAIL) {
            const unsigned trow_b = rmask ? 0u : (unsigned)yo * (unsigned)W * 4u;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                ie_f32x4 o = {tl_bias, tl_bias, tl_bias, tl_bias};
#pragma unroll
                for (int bk = 0; bk < TKB; ++bk) o = ie_mfma16_raw<T>(ta[t][bk], tW[bk], o);
                if (p.tail_clamp) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = fminf(fmaxf(o[i], p.tail_lo), p.tail_hi);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ie_u32x4, o), rst, (int)((trow_b + tl_col[t]) | rmask), 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < SPW; ++j) __builtin_amdgcn_raw_buffer_store_b128(ln[j], rsd, (int)((row_b + st_col[j]) | rmask), 0, 0);
        }
        if constexpr (POOL)       // every step issues it (out of range when there is no pooled row): the counted waits stay uniform
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ie_u32x4, pv), rsp, (int)((prow_b + pl_col) | pmask), 0, 0);
        if constexpr (FIRST && !IE_SC_FIRST_EARLY) conv1_row(it);      // next step's input row; the ring slot it fills was last read two steps ago
    };

    if constexpr (FIRST) {
#pragma unroll
        for (int k = 0; k < PF + 3; ++k) issue_img(k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        conv1_row(-1);                           // x1 row ys - 1 for step 0 (the barrier at the head of step 0 publishes it)
    } else {
#pragma unroll
        for (int it = 0; it < PF; ++it) issue_row(it);
    }
    ie_f32x4 a0[4], a1[4], a2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { a0[nt] = bias4; a1[nt] = bias4; a2[nt] = bias4; }
    const int n_steps = n_it + 2;                 // two more steps: park, then store, the last finished row
    for (int it = 0; it < n_steps; it += 3) {
        step(it, a0, a1, a2);
        if (it + 1 < n_steps) step(it + 1, a2, a0, a1);
        if (it + 2 < n_steps) step(it + 2, a1, a2, a0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

typedef void (*SConvFn)(SConvParams);
// Concrete kernels (hipcc 7.2 dropped host stubs of kernel templates in this file; readable names in rocprof traces as a bonus).
#define IE_DEF_SCONV_PLAIN(TAG, T, NCH, NCF)                                                                                        \
    __global__ __launch_bounds__(NCF * 128) void sconv3_##TAG##_none(SConvParams p) { sconv3_body<T, NCH, NCF, IE_ACT_NONE>(p); }    \
    __global__ __launch_bounds__(NCF * 128) void sconv3_##TAG##_relu(SConvParams p) { sconv3_body<T, NCH, NCF, IE_ACT_RELU>(p); }    \
    __global__ __launch_bounds__(NCF * 128) void sconv3_##TAG##_lrelu(SConvParams p) { sconv3_body<T, NCH, NCF, IE_ACT_LRELU>(p); }
#define IE_DEF_SCONV(TAG, T, NCH, NCF)                                                                                              \
    IE_DEF_SCONV_PLAIN(TAG, T, NCH, NCF)                                                                                            \
    __global__ __launch_bounds__(NCF * 128) void sconv3_##TAG##_relu_pool(SConvParams p) { sconv3_body<T, NCH, NCF, IE_ACT_RELU, true>(p); } \
    __global__ __launch_bounds__(NCF * 128) void sconv3_##TAG##_relu_tail(SConvParams p) { sconv3_body<T, NCH, NCF, IE_ACT_RELU, false, true>(p); }
// the first DoubleConv of a U-Net (3 -> 32 / 64 -> 32 / 64 channels, + pooled copy) in one launch
__global__ __launch_bounds__(256) void sconv3_f16_c32_o32_relu_pool_first(SConvParams p) { sconv3_body<_Float16, 1, 2, IE_ACT_RELU, true, false, false, true>(p); }
__global__ __launch_bounds__(256) void sconv3_bf16_c32_o32_relu_pool_first(SConvParams p) { sconv3_body<__bf16, 1, 2, IE_ACT_RELU, true, false, false, true>(p); }
__global__ __launch_bounds__(512) void sconv3_f16_c64_o64_relu_pool_first(SConvParams p) { sconv3_body<_Float16, 2, 4, IE_ACT_RELU, true, false, false, true>(p); }
__global__ __launch_bounds__(512) void sconv3_bf16_c64_o64_relu_pool_first(SConvParams p) { sconv3_body<__bf16, 2, 4, IE_ACT_RELU, true, false, false, true>(p); }
// stride 2 with the odd rows' MFMAs skipped: the shape RetinexNet uses (retinexnet/model.py:60-62); other stride-2 shapes run the plain kernels
// (every row computed, the even ones stored)
__global__ __launch_bounds__(512) void sconv3_f16_c64_o64_relu_s2(SConvParams p) { sconv3_body<_Float16, 2, 4, IE_ACT_RELU, false, false, true>(p); }
__global__ __launch_bounds__(512) void sconv3_bf16_c64_o64_relu_s2(SConvParams p) { sconv3_body<__bf16, 2, 4, IE_ACT_RELU, false, false, true>(p); }
IE_DEF_SCONV(f16_c32_o32, _Float16, 1, 2)
IE_DEF_SCONV(f16_c32_o64, _Float16, 1, 4)
IE_DEF_SCONV(f16_c64_o32, _Float16, 2, 2)
IE_DEF_SCONV(f16_c64_o64, _Float16, 2, 4)
// four K chunks: plain forms only -- the fused-pool / fused-head forms of this shape spilled 18 / 26 VGPRs into the counted-vmcnt loop (a reload
// there drains the LDS-DMA queue); ie_sconv3_fused_ok() sends such calls to the two-launch path
IE_DEF_SCONV_PLAIN(f16_c128_o64, _Float16, 4, 4)
IE_DEF_SCONV(bf16_c32_o32, __bf16, 1, 2)
IE_DEF_SCONV(bf16_c32_o64, __bf16, 1, 4)
IE_DEF_SCONV(bf16_c64_o32, __bf16, 2, 2)
IE_DEF_SCONV(bf16_c64_o64, __bf16, 2, 4)
IE_DEF_SCONV_PLAIN(bf16_c128_o64, __bf16, 4, 4)

SConvFn pick_kernel(bool h, int nch, int ncf, int act, bool pool, bool tail) {
#define IE_PICK_PLAIN(TAG) return (tail || pool) ? (SConvFn) nullptr : (act == IE_ACT_RELU ? sconv3_##TAG##_relu : (act == IE_ACT_LRELU ? sconv3_##TAG##_lrelu : sconv3_##TAG##_none))
#define IE_PICK_ACT(TAG) return tail ? sconv3_##TAG##_relu_tail : pool ? sconv3_##TAG##_relu_pool : (act == IE_ACT_RELU ? sconv3_##TAG##_relu : (act == IE_ACT_LRELU ? sconv3_##TAG##_lrelu : sconv3_##TAG##_none))
    if (h) {
        if (nch == 1 && ncf == 2) IE_PICK_ACT(f16_c32_o32);
        if (nch == 1 && ncf == 4) IE_PICK_ACT(f16_c32_o64);
        if (nch == 2 && ncf == 2) IE_PICK_ACT(f16_c64_o32);
        if (nch == 4) IE_PICK_PLAIN(f16_c128_o64);
        IE_PICK_ACT(f16_c64_o64);
    }
    if (nch == 1 && ncf == 2) IE_PICK_ACT(bf16_c32_o32);
    if (nch == 1 && ncf == 4) IE_PICK_ACT(bf16_c32_o64);
    if (nch == 2 && ncf == 2) IE_PICK_ACT(bf16_c64_o32);
    if (nch == 4) IE_PICK_PLAIN(bf16_c128_o64);
    IE_PICK_ACT(bf16_c64_o64);
#undef IE_PICK_ACT
#undef IE_PICK_PLAIN
}

}  // namespace

// ... and its fused max-pool / fused 1x1 head forms?  (Every shape but the four-chunk one.)
bool ie_sconv3_fused_ok(int nsrc, const int* cs) {
    int cin = 0;
    for (int s = 0; s < nsrc; ++s) cin += cs[s];
    return cin <= 64;
}

// Does the strip kernel cover this call?  (ie_mconv_nhwc asks before it builds its own launch.)
bool ie_sconv3_applicable(int nsrc, const int* cs, int Cout, int kh, int kw, int stride, int pad, int pad_mode, int act, bool residual, bool clamp,
                          bool out_f32, int W) {
    static const int off = getenv("IE_SCONV") ? !atoi(getenv("IE_SCONV")) : 0;      // IE_SCONV=0: A/B switch back to the tiled kernel
    // stride 2 (retinexnet/model.py:60-62) = the stride-1 rows with the even rows / columns stored: 4x the MFMA work at the strip kernel's rate
    // beats the generic kernel, which computes the same positions at a lower one (RetinexNet 1080p: 0.52 -> 0.30 ms)
    static const int s2 = getenv("IE_SCONV_S2") ? atoi(getenv("IE_SCONV_S2")) : 1;
    if (off || kh != 3 || kw != 3 || (stride != 1 && !(stride == 2 && s2)) || pad != 1 || residual || clamp || out_f32) return false;
    if (pad_mode != IE_PAD_ZEROS && pad_mode != IE_PAD_REPLICATE) return false;
    if (act != IE_ACT_NONE && act != IE_ACT_RELU && act != IE_ACT_LRELU) return false;
    if (Cout != 32 && Cout != 64) return false;
    if (W < 64) return false;                                                          // a 128-column strip would be mostly padding
    static const int wide = getenv("IE_SCONV_WIDE") ? atoi(getenv("IE_SCONV_WIDE")) : 1;   // four K chunks (128 input channels), Cout 64 only
    if (nsrc == 1) return cs[0] == 32 || cs[0] == 64 || (wide && cs[0] == 128 && Cout == 64);
    if (nsrc != 2) return false;
    return (cs[0] == 32 && cs[1] == 32) || (wide && cs[0] == 64 && cs[1] == 64 && Cout == 64);
}

// The first DoubleConv (+ 2x2 max-pool) of a U-Net in one launch: conv1 8 -> C (thin packing), conv2 C -> C, C = 32 or 64, both 3x3 / zero padding /
// bias / ReLU.  img8: NHWC 8-channel image; dst: NHWC C channels; pool_dst: its pooled copy.
int ie_sconv3_first_launch(const void* img8, const void* w1, const float* b1, const void* wpk2, const float* b2, void* dst, void* pool_dst, int dtype,
                           int N, int H, int W, int C, hipStream_t st) {
    if (!img8 || !w1 || !wpk2 || !dst || !pool_dst || (C != 32 && C != 64) || H < 2 || W < 64 || N <= 0) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3_first: C 32 / 64, W >= 64, H >= 2");
    if ((double)H * W * 256.0 >= 2147483647.0) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3_first: frame too large for 31-bit offsets");
    SConvParams p;
    memset(&p, 0, sizeof(p));
    const int nch = C / 32, ncf = C / 16;
    p.img = img8; p.w1 = w1; p.b1 = b1; p.wpk = wpk2; p.bias = b2; p.dst = dst; p.pool = pool_dst; p.N = N; p.H = H; p.W = W; p.sub = 1;
    p.th[0] = p.th[1] = H; p.tw[0] = p.tw[1] = W;
    p.strips = ie_cdiv(W, TW);
    {
        const int slots = 256 * (C == 64 ? 1 : 2);
        const long long base = (long long)p.strips * N;
        int best_c = 1;
        double best = -1.0;
        for (int c = 1; c <= H / 16 + 1 && c <= 64; ++c) {
            int rpb = ie_cdiv(H, c);
            rpb += rpb & 1;
            const int cc = ie_cdiv(H, rpb);
            const long long blocks = base * cc;
            const long long rounds = (blocks + slots - 1) / slots;
            const double score = (double)blocks / (double)(rounds * slots) * ((double)rpb / (double)(rpb + 4));
            if (score > best + 1e-9) { best = score; best_c = cc; }
        }
        p.rows_per_block = ie_cdiv(H, best_c);
        p.rows_per_block += p.rows_per_block & 1;
        p.chunks = ie_cdiv(H, p.rows_per_block);
    }
    const long long blocks = (long long)p.strips * p.chunks * N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "ie_sconv3_first: bad grid");
    const size_t lds = (size_t)4 * SRC_ROW_B * nch + 3 * (size_t)TW * (C * 2) + (size_t)6 * 3072;       // NR = 4 (PF = 3), three parked rows, NRI = PF + 3 image rows
    SConvFn fn = C == 32 ? (dtype == IE_F16 ? sconv3_f16_c32_o32_relu_pool_first : sconv3_bf16_c32_o32_relu_pool_first)
                         : (dtype == IE_F16 ? sconv3_f16_c64_o64_relu_pool_first : sconv3_bf16_c64_o64_relu_pool_first);
    IE_GRANT_LDS(fn, lds);
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(ncf * 128), lds, st, p);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

int ie_sconv3_launch(const void* const* srcs, const int* cs, int nsrc, const void* wpk, const float* bias, void* dst, int dtype, int N, int H, int W,
                     int Cout, int act, float slope, int pad_mode, const int* src_h, const int* src_w, void* pool_dst, const ie_sconv_tail* tail, int stride, hipStream_t st) {
    SConvParams p;
    memset(&p, 0, sizeof(p));
    int nch = 0;
    p.src[0] = srcs[0]; p.src[1] = srcs[nsrc - 1];
    p.tps[0] = cs[0] * 2; p.tps[1] = cs[nsrc - 1] * 2;
    p.th[0] = (src_h && src_h[0] > 0) ? src_h[0] : H; p.tw[0] = (src_w && src_w[0] > 0) ? src_w[0] : W;
    p.th[1] = (src_h && src_h[nsrc - 1] > 0) ? src_h[nsrc - 1] : H; p.tw[1] = (src_w && src_w[nsrc - 1] > 0) ? src_w[nsrc - 1] : W;
    for (int s = 0; s < nsrc; ++s)
        for (int c0 = 0; c0 < cs[s]; c0 += 32) { p.tsel[nch] = s; p.coff[nch] = c0 * 2; ++nch; }
    if (pool_dst && (act != IE_ACT_RELU || H < 2 || W < 2)) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3: the fused 2x2 max-pool needs a ReLU conv of at least 2x2 pixels");
    if (stride != 1 && (stride != 2 || pool_dst || tail)) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3: stride 1, or 2 without a fused pool / tail");
    p.sub = stride;
    if (tail) {
        if (pool_dst || act != IE_ACT_RELU || (W & 3) || tail->cout < 1 || tail->cout > 16 || !tail->w || !tail->b || !tail->dst)
            IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3: the fused 1x1 tail needs a ReLU conv, W %% 4 == 0 and at most 16 tail channels");
        if ((double)tail->cout * H * W * 4.0 >= 2147483647.0) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3: tail frame too large for 31-bit offsets");
        p.tail_w = tail->w; p.tail_b = tail->b; p.tail_dst = tail->dst; p.tail_c = tail->cout; p.tail_clamp = tail->clamp; p.tail_lo = tail->lo; p.tail_hi = tail->hi;
    }
    p.wpk = wpk; p.bias = bias; p.dst = dst; p.pool = pool_dst; p.N = N; p.H = H; p.W = W; p.slope = slope; p.replicate = pad_mode == IE_PAD_REPLICATE;
    if ((double)H * W * 256.0 >= 2147483647.0) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3: frame too large for 31-bit offsets");
    p.strips = ie_cdiv(W, TW);
    // Row chunks per strip: the grid should fill whole rounds of the chip's resident workgroup slots (a 2.1-round grid runs 3 rounds),
    // while every chunk pays 3 rows of pipeline fill.  Score = slot fill x useful-row fraction; 8-wave workgroups (Cout 64) sit one
    // per CU, 4-wave ones two.
    {
        const int slots = 256 * (Cout == 64 ? 1 : 2);
        const long long base = (long long)p.strips * N;
        int best_c = 1;
        double best = -1.0;
        for (int c = 1; c <= H / 16 + 1 && c <= 64; ++c) {
            int rpb = ie_cdiv(H, c);
            if (pool_dst) rpb += rpb & 1;            // row pairs stay inside one chunk
            const int cc = ie_cdiv(H, rpb);
            const long long blocks = base * cc;
            const long long rounds = (blocks + slots - 1) / slots;
            const double score = (double)blocks / (double)(rounds * slots) * ((double)rpb / (double)(rpb + 3));
            if (score > best + 1e-9) { best = score; best_c = cc; }
        }
        p.rows_per_block = ie_cdiv(H, best_c);
        if (pool_dst) p.rows_per_block += p.rows_per_block & 1;
        p.chunks = ie_cdiv(H, p.rows_per_block);
    }
    const long long blocks = (long long)p.strips * p.chunks * N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "ie_sconv3: bad grid");
    const int ncf = Cout / 16;
    const size_t lds = (size_t)(nch == 4 ? 3 : ((nch == 2 && ncf == 2) ? IE_SC_NR_C64O32 : 4)) * SRC_ROW_B * nch + (pool_dst ? 3 : 2) * (size_t)TW * (Cout * 2) + (nch == 4 ? 256 : 0);       // + the bias area of the four-chunk kernels
    SConvFn fn = pick_kernel(dtype == IE_F16, nch, ncf, act, pool_dst != nullptr, tail != nullptr);
    if (!fn) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_sconv3: the fused max-pool / head forms cover Cin <= 64 (ask ie_sconv3_fused_ok)");
    static const int s2skip = getenv("IE_SCONV_S2SKIP") ? atoi(getenv("IE_SCONV_S2SKIP")) : 1;      // A/B: 0 = every row computed
    if (stride == 2 && s2skip && nch == 2 && ncf == 4 && act == IE_ACT_RELU) fn = dtype == IE_F16 ? sconv3_f16_c64_o64_relu_s2 : sconv3_bf16_c64_o64_relu_s2;
    IE_GRANT_LDS(fn, lds);
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(ncf * 128), lds, st, p);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
