// Zero-DCE, round-4 schedule: the partial sum P = W7a * x1 + b7 no longer exists.
//
//   K12X : e_conv1 + e_conv2                      image -> x2                       12 + 64      =  76 B/px   (K12: 140)
//   K345 : e_conv3 + e_conv4 + e_conv5            x2 -> x5   (ie_zdce_fused.hip)    64 + 64      = 128 B/px
//   K67X : e_conv1 (again) + e_conv6 + e_conv7    image, x2, x5 -> enhanced         12 + 128 + 12 = 152 B/px  (K67N: 216)
//          + tanh + LE curve
//
// Round 3 ended with K12 write-bound (128 B/px written at 3.2 TB/s) and K67N on the HBM roof (225 B/px at 4.8 TB/s) beside an
// MFMA-bound K345: the 128 B/px P costs (written by K12, read by K67N) bought 36 MFMAs per row that K67N did not have to run.
// e_conv1 is 1 % of the network's arithmetic, so the last kernel recomputes x1 from the image it reads anyway (for the curve)
// and contracts e_conv7 over BOTH of its sources [x1, x6] (zero_dce.py:182): 356 B/px instead of 484, every kernel on the MFMA
// side of the balance point, and e_conv7 accumulates in fp32 over all 64 input channels (P was rounded to 16 bits).
//
// Reference semantics: src/mon/vision/enhance/llie/zero_dce/zero_dce.py:173-191.
//
// Both kernels are two-stage eight-wave strip kernels (ie_zdce_fused.hip's machinery in ie_conv_first32.hip's compact form): a workgroup
// marches down a 128-column strip, weights are MFMA A operands in registers for the whole launch, rows travel through LDS rings,
// every pixel fragment read from LDS feeds the three ky taps into three rotating accumulator sets, one s_barrier per row.
#include "ie_zdce_internal.h"
#include <string.h>
#include <type_traits>

namespace {

constexpr int TW = 128, MARG = 2, TS = 124;    // computed / dropped-per-side / stored columns of a strip (no halo columns are loaded: each layer simply
                                               // does not use the outermost computed column of the one before; e_conv1 has the image's halo)
constexpr int PXB = 64;                        // one pixel of a 32-channel 16-bit tensor
constexpr int ROW_B = TW * PXB;                // 8,192: parked output rows (pixel q at index q)
constexpr int ROWH_B = (TW + 2) * PXB;         // 8,320: rows that feed a convolution: pixel q at index q + 1, so that taps q - 1 .. q + 1 of EVERY tile are base + immediate
                                               // (the two end pixels are never written: they only feed the dropped columns 0 and 127)
constexpr int IPX = 136, IPB = 8;              // converted image row: [136 px][R, G, B, 0] 16-bit; pixel i = image column xc0 - 1 + i (fragments read up to pixel 134)
constexpr int IMG_PART_B = IPX * IPB;          // 1,088 (bf16 operands: a hi and a lo part, so the fp32 pixel survives bf16's 8 mantissa bits)
constexpr int RAW_SLOT_B = 2048;               // three fp32 plane rows of 136 pixels = 1,632 bytes (8 DMA instructions x 256 bytes)
constexpr unsigned OOB = 0x80000000u;

#ifndef IE_X_ST_AUX
#define IE_X_ST_AUX 0     // cache-policy bits of the x2 / x5 row stores (experiment: 2 = nt)
#endif
#ifndef IE_XEXP
#define IE_XEXP 0     // K67X timing experiments (WRONG results), tools/exp_build.sh ie_zdce_x.hip <tag> -DIE_XEXP=bits: 1 no curve fillers, 2 no e_conv1 / row conversion,
#endif                // 4 no hand-off writes, 8 no x2 / x5 DMA, 16 no image DMA / output stores, 32 no e_conv6 MFMAs, 64 no e_conv7 MFMAs, 128 curve without its transcendentals,
                      // 256 transcendentals only, 512 transcendentals replaced by v_fma
#define IE_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef IE_PRIO_TOGGLE
#define IE_PRIO_TOGGLE 0    // experiment: a wave drops its issue priority right behind each of its MFMAs and raises it again at the end of the gap, so that the SIMD's
#endif                      // two waves alternate on the matrix pipe instead of the older one taking every slot it is ready for
#if IE_PRIO_TOGGLE
#define IE_PRIO_LO() do { IE_SB(); __builtin_amdgcn_s_setprio(0); IE_SB(); } while (0)
#define IE_PRIO_HI() do { IE_SB(); __builtin_amdgcn_s_setprio(1); IE_SB(); } while (0)
#else
#define IE_PRIO_LO() do { } while (0)
#define IE_PRIO_HI() do { } while (0)
#endif
typedef __attribute__((address_space(3))) void* lds_ptr;

template <typename T> constexpr int parts_of() { return ie_elem<T>::dtype == IE_BF16 ? 2 : 1; }

// ---- pieces shared by the two kernels ----------------------------------------------------------------------------------------------------
// raw image rows: instruction j of wave w4 covers flat indices (4 j + w4) * 64 + lane of [plane][pixel] (3 x 136 floats)
__device__ __forceinline__ void raw_offsets(int w4, int lane, int xc0, int W, size_t plane, unsigned (&raw_off)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = (4 * j + w4) * 64 + lane, pl = idx / IPX, px = idx - pl * IPX, x = xc0 - 1 + px;
        raw_off[j] = (idx < 3 * IPX && x >= 0 && x < W) ? (unsigned)(((size_t)pl * plane + x) * 4) : OOB;
    }
}

// one fp32 raw row (three planes) -> [px][R, G, B, 0] 16-bit (hi part; bf16: + lo part = x - hi).  Lane ci of the stage's 256 converts pixel ci (< 136).
template <typename T> __device__ __forceinline__ void convert_row(const char* raw_slot, char* img_slot, char* dump, int ci, int lane) {
    const bool ok = ci < IPX;
    const char* rs = raw_slot + (ok ? ci : 0) * 4;
    const float r = *(const float*)rs, g = *(const float*)(rs + IPX * 4), b = *(const float*)(rs + 2 * IPX * 4);
    char* d = ok ? img_slot + ci * IPB : dump + lane * 8;
    *(ie_u32x2*)d = ie_u32x2{ie_pack2<T>(r, g), ie_pack2<T>(b, 0.f)};
    if constexpr (parts_of<T>() == 2) {
        const float rl = r - (float)(T)r, gl = g - (float)(T)g, bl = b - (float)(T)b;
        *(ie_u32x2*)(d + (ok ? IMG_PART_B : 512)) = ie_u32x2{ie_pack2<T>(rl, gl), ie_pack2<T>(bl, 0.f)};
    }
}

// a finished row (ReLU, 16 bits, zero where `keep` says the pixel lies outside the image = the next layer's zero padding) into an LDS row image as 16-byte
// pieces: lanes l / l + 16 exchange halves, so a ds_write_b128 serves 8 consecutive pixels x 16 bytes (every bank once).  Wave = (cout fragment cf, half hf).
template <typename T, bool MASKED = true> __device__ __forceinline__ void put_row(const ie_f32x4 (&acc)[4], char* base, int em_off, unsigned keep) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const ie_u32x2 a = {ie_pack2_relu<T>(acc[2 * jj][0], acc[2 * jj][1]), ie_pack2_relu<T>(acc[2 * jj][2], acc[2 * jj][3])};
        const ie_u32x2 b2 = {ie_pack2_relu<T>(acc[2 * jj + 1][0], acc[2 * jj + 1][1]), ie_pack2_relu<T>(acc[2 * jj + 1][2], acc[2 * jj + 1][3])};
        const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
        const bool k = !MASKED || ((keep >> jj) & 1u);
        *(ie_u32x4*)(base + em_off + jj * 32 * PXB) = ie_u32x4{k ? sx[0] : 0u, k ? sy[0] : 0u, k ? sx[1] : 0u, k ? sy[1] : 0u};
    }
}
// one tile pair (2 jj, 2 jj + 1) of put_row: lets a stream hand a row over in two halves, the first one while its last MFMAs still run
template <typename T, bool MASKED = true> __device__ __forceinline__ void put_pair(const ie_f32x4 (&acc)[4], int jj, char* base, int em_off, unsigned keep) {
    const ie_u32x2 a = {ie_pack2_relu<T>(acc[2 * jj][0], acc[2 * jj][1]), ie_pack2_relu<T>(acc[2 * jj][2], acc[2 * jj][3])};
    const ie_u32x2 b2 = {ie_pack2_relu<T>(acc[2 * jj + 1][0], acc[2 * jj + 1][1]), ie_pack2_relu<T>(acc[2 * jj + 1][2], acc[2 * jj + 1][3])};
    const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
    const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
    const bool k = !MASKED || ((keep >> jj) & 1u);
    *(ie_u32x4*)(base + em_off + jj * 32 * PXB) = ie_u32x4{k ? sx[0] : 0u, k ? sy[0] : 0u, k ? sx[1] : 0u, k ? sy[1] : 0u};
}
// this lane's 16-byte piece of tile pair 0 in a row image whose pixel q sits at index q + halo; pair 1 is 32 pixels on.  Wave = (cout fragment cf, half hf).
__device__ __forceinline__ int put_off(int cf, int hf, int p16, int kq, int halo) {
    const int P = 16 * (4 * hf + (kq & 1)) + p16 + halo;
    return P * PXB + (((2 * cf + (kq >> 1)) ^ ((P >> 1) & 3)) << 4);
}

// e_conv1 of one converted image row into three rotating accumulator sets: K = 8 kx taps x 4 channels per MFMA (taps 3..7 and channel 3 carry zero weights),
// lane (p16, kq) of tile j reads taps kx = 2 kq, 2 kq + 1 = 16 bytes at ring pixel (column + 2 kq)
template <typename T> __device__ __forceinline__ void conv1_row(const char* img_slot, const int (&rdi)[4], const ie_u32x4 (&w1)[3], const ie_f32x4& bias4,
                                                               ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) {
#pragma unroll
    for (int part = 0; part < parts_of<T>(); ++part) {
        ie_u32x4 bx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const ie_u32x2 lo = *(const ie_u32x2*)(img_slot + part * IMG_PART_B + rdi[j]), hi = *(const ie_u32x2*)(img_slot + part * IMG_PART_B + rdi[j] + 8);
            bx[j] = ie_u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            accN[j] = ie_mfma16_raw<T>(w1[0], bx[j], part == 0 ? bias4 : accN[j]);
            accM[j] = ie_mfma16_raw<T>(w1[1], bx[j], accM[j]);
            accO[j] = ie_mfma16_raw<T>(w1[2], bx[j], accO[j]);
        }
    }
}

// Fragment reads of a halo row image (pixel q at index q + 1): lane (p16, kq) of the wave's half hf reads chunk kq of pixel 64 hf + 16 j + p16 + kx - 1 for tile j
// and tap column kx: base[kx] + 1024 j -- three address registers, every tile an immediate (the 16-byte chunks of pixel P are XOR-swizzled by (P >> 1) & 3, which
// does not depend on the tile: conflict-free ds_read_b128 at every tap offset).
struct RowRead { int base[3]; };
__device__ __forceinline__ RowRead row_read(int hf, int p16, int kq) {
    RowRead r;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int P = 64 * hf + p16 + kx;
        r.base[kx] = P * PXB + ((kq ^ ((P >> 1) & 3)) << 4);
    }
    return r;
}

// One source row (32 channels) into three rotating accumulator sets: 12 fragments read AHEAD ahead, 36 MFMAs in a pinned order; fill(s) is placed behind MFMA s
// (a wave's vector work is close to free between its OWN MFMAs while the gap's issue costs stay under the MFMA's 16 cycles -- 8 for the MFMA itself, 4 per vector
// instruction, 8 per transcendental one -- and costs its full length as a phase of its own: tools/micro/coissue.hip, MI355X_MICROARCH.md constants table).
// The fragment read of a group comes right behind the group's first MFMA, so a gap that ends in a transcendental filler is followed by an MFMA, not by a vector
// instruction (hipcc puts an s_nop between the two, another 4 cycles of the issue port).
// compiler-scheduled form of the same row (no pinned order): experiment IE_K67X_UNPIN_A
template <typename T>
__device__ __forceinline__ void mfma_row_free(const char* slot, const RowRead& rr, const ie_u32x4* wv, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4],
                                              bool first, const ie_f32x4& init) {
    const char* bk[3] = {slot + rr.base[0], slot + rr.base[1], slot + rr.base[2]};
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int kx = g >> 2, j = g & 3;
        const ie_u32x4 bx = *(const ie_u32x4*)(bk[kx] + 1024 * j);
        accN[j] = ie_mfma16_raw<T>(wv[0 * 3 + kx], bx, (first && kx == 0) ? init : accN[j]);
        accM[j] = ie_mfma16_raw<T>(wv[1 * 3 + kx], bx, accM[j]);
        accO[j] = ie_mfma16_raw<T>(wv[2 * 3 + kx], bx, accO[j]);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int f = 0; f < 9; ++f) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
}

template <typename T, int AHEAD = 3, typename F>
__device__ __forceinline__ void mfma_row(const char* slot, const RowRead& rr, const ie_u32x4* wv, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4],
                                         bool first, const ie_f32x4& init, F&& fill) {
    ie_u32x4 bx[12];
    const char* bk[3] = {slot + rr.base[0], slot + rr.base[1], slot + rr.base[2]};
    auto rd = [&](int g) __attribute__((always_inline)) { bx[g] = *(const ie_u32x4*)(bk[g >> 2] + 1024 * (g & 3)); };
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) rd(g);
    IE_SB();
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int kx = g >> 2, j = g & 3;
        accN[j] = ie_mfma16_raw<T>(wv[0 * 3 + kx], bx[g], (first && kx == 0) ? init : accN[j]);
        IE_PRIO_LO();
        if (g + AHEAD < 12) rd(g + AHEAD);
        fill(3 * g);
        IE_PRIO_HI();
        IE_SB();
        accM[j] = ie_mfma16_raw<T>(wv[1 * 3 + kx], bx[g], accM[j]);
        IE_PRIO_LO();
        fill(3 * g + 1);
        IE_PRIO_HI();
        IE_SB();
        accO[j] = ie_mfma16_raw<T>(wv[2 * 3 + kx], bx[g], accO[j]);
        IE_PRIO_LO();
        fill(3 * g + 2);
        IE_PRIO_HI();
        IE_SB();
    }
}

// The same row for a wave that owns BOTH 16-row tiles of its layer over a quarter of the strip (2 pixel tiles): every fragment feeds SIX MFMAs, a row's 36 MFMAs need 6
// ds_read_b128 instead of 12 (tools/micro/mfma_pair.hip form 8: -4 % wall time per MFMA at the power cap).  acc index = 2 * tile + j.
#ifndef IE_K67X_AHBOTH
#define IE_K67X_AHBOTH 1
#endif
template <typename T, int AHEAD = IE_K67X_AHBOTH, bool ZINIT = true>
__device__ __forceinline__ void mfma_row_both(const char* smem_base, const char* slot, const RowRead& rr, const ie_u32x4* w0, const ie_u32x4* w1, ie_f32x4 (&accN)[4],
                                              ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], bool first, const ie_f32x4& b0 = ie_f32x4{0.f, 0.f, 0.f, 0.f},
                                              const ie_f32x4& b1 = ie_f32x4{0.f, 0.f, 0.f, 0.f}) {
    // ZINIT (K67X): the new row starts from the inline constant 0 -- no registers -- and the bias joins in the curve; else from the bias pair b0 / b1
    const ie_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const ie_f32x4 init0 = ZINIT ? zero : b0, init1 = ZINIT ? zero : b1;
    ie_u32x4 bx[6];
    // the row's offset stays a scalar the compiler cannot fold into per-row copies of the three lane offsets (six address registers held across the stream spilled)
    unsigned so = (unsigned)(slot - smem_base);
    asm volatile("" : "+s"(so));
    auto rd = [&](int g) __attribute__((always_inline)) { bx[g] = *(const ie_u32x4*)(smem_base + (unsigned)(rr.base[g >> 1] + 1024 * (g & 1)) + so); };
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) rd(g);
    IE_SB();
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int kx = g >> 1, j = g & 1;
        accN[j] = ie_mfma16_raw<T>(w0[0 * 3 + kx], bx[g], (first && kx == 0) ? init0 : accN[j]);
        if (g + AHEAD < 6) rd(g + AHEAD);
        IE_SB();
        accM[j] = ie_mfma16_raw<T>(w0[1 * 3 + kx], bx[g], accM[j]); IE_SB();
        accO[j] = ie_mfma16_raw<T>(w0[2 * 3 + kx], bx[g], accO[j]); IE_SB();
        accN[2 + j] = ie_mfma16_raw<T>(w1[0 * 3 + kx], bx[g], (first && kx == 0) ? init1 : accN[2 + j]); IE_SB();
        accM[2 + j] = ie_mfma16_raw<T>(w1[1 * 3 + kx], bx[g], accM[2 + j]); IE_SB();
        accO[2 + j] = ie_mfma16_raw<T>(w1[2 * 3 + kx], bx[g], accO[2 + j]); IE_SB();
    }
}

// ==========================================================================================================================================
// K12X: e_conv1 + e_conv2 -> x2
// ==========================================================================================================================================
struct K12XParams {
    const float* img; const void* w1; const float* b1; const void* w2; const float* b2; void* dst;
    int H, W, y_begin, y_end, rows_per_block, strips, chunks, n_begin;
};

template <typename T> struct K12XGeom {
    static constexpr int PFR = 2, NRAW = PFR + 1, NRI = 2, NRH = 3, NPARK = 2;
    static constexpr int IMG_ROW_B = parts_of<T>() * IMG_PART_B;
    static constexpr int OFF_RAW = 0, OFF_IMG = OFF_RAW + NRAW * RAW_SLOT_B, OFF_H = OFF_IMG + NRI * IMG_ROW_B, OFF_PARK = OFF_H + NRH * ROWH_B;
    static constexpr int OFF_DUMP = OFF_PARK + NPARK * ROW_B, LDS_B = OFF_DUMP + 1024;        // f16: 6,144 + 2,176 + 24,960 + 16,384 + 1,024 = 50,688: two workgroups per CU
};

template <typename T>
__device__ __forceinline__ void k12x_body(const K12XParams& p) {
    using G = K12XGeom<T>;
    using frag = ie_u32x4;
    constexpr int PFR = G::PFR, NRAW = G::NRAW, NRI = G::NRI, NRH = G::NRH, NPARK = G::NPARK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ie_fp16_saturate_on();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int stg = wave >> 2, w4 = wave & 3, cf = w4 & 1, hf = w4 >> 1;      // stage; 16-cout fragment; half of the strip (tiles 4 hf .. 4 hf + 3)

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = p.n_begin + b;
    const int xc0 = strip * TS - MARG;
    const int H = p.H, W = p.W;
    const int ys = p.y_begin + chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, p.y_end);
    const int n_steps = (ye - ys) + 8;

    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000); };
    const size_t plane = (size_t)H * W;
    const auto rsI = mk((const char*)(p.img + (size_t)n * 3 * plane), 3 * plane * 4);
    const auto rsD = mk((char*)p.dst + (size_t)n * plane * PXB, plane * PXB);

    frag wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (stg == 0) wv[t] = t < 3 ? ((const frag*)p.w1)[(cf * 3 + t) * 64 + lane] : frag{0u, 0u, 0u, 0u};
        else wv[t] = ((const frag*)p.w2)[(cf * 9 + t) * 64 + lane];
    }
    ie_f32x4 bias4;
    {
        const float* bp = stg ? p.b2 : p.b1;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = bp[16 * cf + 4 * kq + i];
    }
    const int em_off = put_off(cf, hf, p16, kq, stg == 0 ? 1 : 0);      // stage A writes the hand-off ring (halo rows), stage B the parked rows
    unsigned em_in = 0;
    {
        const int q = 16 * (4 * hf + (kq & 1)) + p16;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int x = xc0 + q + 32 * jj;
            em_in |= (x >= 0 && x < W) ? (1u << jj) : 0u;
        }
    }
    ie_f32x4 a0[4], a1[4], a2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a0[j] = bias4; a1[j] = bias4; a2[j] = bias4; }

    if (stg == 0) {
        // ---- stage A: image rows (LDS-DMA, converted one step ahead) -> e_conv1 -> hand-off ring; image row of step t: ys - 2 + t ----
        unsigned raw_off[2];
        raw_offsets(w4, lane, xc0, W, plane, raw_off);
        auto issue_raw = [&](int t) __attribute__((always_inline)) {
            const int yi = ys - 2 + t;
            const unsigned rmask = (yi >= 0 && yi < H) ? 0u : OOB;
            const unsigned row_b = rmask ? 0u : (unsigned)yi * (unsigned)W * 4u;
            char* slot = smem + G::OFF_RAW + (t % NRAW) * RAW_SLOT_B;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(slot + (4 * j + w4) * 256), 4, (int)((row_b + raw_off[j]) | rmask), 0, 0, 0);
        };
        const int ci = w4 * 64 + lane;
        auto convert = [&](int t) __attribute__((always_inline)) {
            convert_row<T>(smem + G::OFF_RAW + (t % NRAW) * RAW_SLOT_B, smem + G::OFF_IMG + (t % NRI) * G::IMG_ROW_B, smem + G::OFF_DUMP, ci, lane);
        };
        int rdi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rdi[j] = (16 * (4 * hf + j) + p16 + 2 * kq) * IPB;
        const frag w1[3] = {wv[0], wv[1], wv[2]};

#pragma unroll
        for (int t = 0; t <= PFR; ++t) issue_raw(t);
        ie_wait_vm_imm<2 * PFR>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // prologue barrier (all eight waves): raw row 0 has landed
        convert(0);
        auto step_a = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            ie_wait_vm_imm<2 * (PFR - 1)>();             // raw row t + 1 has landed (behind it: rows up to t + PFR)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            conv1_row<T>(smem + G::OFF_IMG + (t % NRI) * G::IMG_ROW_B, rdi, w1, bias4, accN, accM, accO);
            convert(t + 1);                              // next step's image row (its ring slot was last read at step t - 1)
            issue_raw(t + 1 + PFR);
            const int m = ys - 3 + t;                    // the x1 row finished now; outside the image it is e_conv2's zero padding
            put_row<T>(accO, smem + G::OFF_H + (t % NRH) * ROWH_B, em_off, (m >= 0 && m < H) ? em_in : 0u);
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_a(t, a0, a1, a2);
            if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0);
        }
    } else {
        // ---- stage B: e_conv2; finished rows are parked as whole 64-byte pixel lines and leave one step later, 1 KiB of lines per store instruction ----
        const RowRead rr = row_read(hf, p16, kq);
        unsigned st_col[2];
        int st_lds[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 32 * w4 + 16 * i + (lane >> 2), c = lane & 3, x = xc0 + q;
            st_col[i] = (q >= MARG && q < MARG + TS && x < W) ? (unsigned)(x * PXB + c * 16) : OOB;
            st_lds[i] = q * PXB + ((c ^ ((q >> 1) & 3)) << 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // prologue barrier
        // step t: the x1 row ys - 5 + t (stage A's step t - 2) finishes x2 row ys - 6 + t; accN = the row finished at step t - 1 is parked into slot t % 2;
        // the row parked at step t - 1 (ys - 8 + t) leaves
        auto step_b = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int yo = ys - 8 + t;
            const unsigned rmask = (yo >= ys && yo < ye) ? 0u : OOB;
            const char* islot = smem + G::OFF_PARK + ((t + 1) % NPARK) * ROW_B;        // parked at step t - 1
            put_row<T>(accN, smem + G::OFF_PARK + (t % NPARK) * ROW_B, em_off, 0x3u);
            auto nofill = [&](int) __attribute__((always_inline)) {};
            mfma_row<T, 3>(smem + G::OFF_H + ((t + 1) % NRH) * ROWH_B, rr, wv, accN, accM, accO, true, bias4, nofill);     // (a third fragment ahead costs the second workgroup per CU)
            // the leaving row is read AFTER the MFMA stream (its slot is not written during this step): 8 registers that do not have to live across it
            ie_u32x4 ln[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) ln[i] = *(const ie_u32x4*)(islot + st_lds[i]);
            const unsigned row_b = rmask ? 0u : (unsigned)yo * (unsigned)W * PXB;
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_buffer_store_b128(ln[i], rsD, (int)((row_b + st_col[i]) | rmask), 0, IE_X_ST_AUX);
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_b(t, a0, a1, a2);
            if (t + 1 < n_steps) step_b(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_b(t + 2, a1, a2, a0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void zdce_k12x_conv1_conv2_f16(K12XParams p) { k12x_body<_Float16>(p); }
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void zdce_k12x_conv1_conv2_bf16(K12XParams p) { k12x_body<__bf16>(p); }

// ==========================================================================================================================================
// K67X: e_conv1 (recomputed) + e_conv6 + e_conv7 + tanh + LE curve -> enhanced
// ==========================================================================================================================================
// Stage A (waves 0-3 = 2 cout fragments x 2 strip halves): e_conv1 on the image row (12 MFMAs, as in K12X) and e_conv6 on [x2, x5] (72 MFMAs, rows by LDS-DMA);
//   both finished rows (same row index) go to stage B through 3-slot LDS rings.  Stage A also carries every chore of the step that is not stage B's own
//   arithmetic -- ALL the DMA issue (x2, x5, the raw image row), the image row conversion and the `enhanced` stores: in-kernel stamps of the first build had
//   stage B as the step's critical chain with stage A waiting 1,100 of 4,000 cycles at the barrier (profiles/r04/k67x_experiments.log).
// Stage B (waves 4-7 = the four quarters of the strip; IE_K67X_BOTH): e_conv7 over [x1, x6], BOTH 16-row tiles per wave (72 MFMAs per wave and row; all 36 weight
//   fragments = 144 VGPRs).  e_conv7's 24 output rows are permuted on the host (ie_zdce_plan.hip): row 4 kq + i of tile j is curve map r_{4j+i} of colour kq (kq = 3:
//   zero rows), scaled by 2 / ln 2, so after the MFMAs lane (pixel, kq) holds r_0 .. r_7 of ONE colour of its pixel as log2(e) * 2x: tanh = 1 - 2 / (exp2(acc) + 1) and
//   the eight curve steps (zero_dce.py:183-188) run straight off the accumulators in one lane: fp32 pixel from the raw image ring in, staged row out (it leaves a step
//   later as 16-byte vectors).  A fragment feeds SIX MFMAs (both tiles x ky), so a wave reads 12 fragments per step instead of 24: at the board's power cap that is
//   wall time (tools/micro/mfma_pair.hip form 8): 8.36 -> 8.02 ms per 64 frames same box.  To fit 256 VGPRs: accumulators start from the inline constant 0 and the bias
//   (a 32-float LDS table) joins in the curve; the row offsets stay scalar (six folded address registers held across the streams had spilled); fragments one ahead.
//   The whole curve of the row finished at step t - 1 runs as ONE BURST at the head of step t, before the stage's MFMAs (finding 14); FOUR accumulator sets rotate (the
//   finished row is only read while the set freed a step earlier restarts).
//   IE_K67X_BOTH = 0 keeps the first mapping (2 row tiles x 2 strip halves; the tile-1 waves one step behind the tile-0 waves, y4 through an LDS exchange row).
struct K67XParams {
    const float* img; const void* x2; const void* x5;
    const void* w1; const float* b1; const void* w6; const float* b6; const void* w7; const float* b7;
    const void* w1q;     // e_conv1 as two K = 32 fragments over a 3-row window (IE_K67X_BOTHA)
    float* enhanced; float* adjust; unsigned char* u8;
    int H, W, y_begin, y_end, rows_per_block, strips, chunks, n_begin;
    unsigned long long* stamps;   // diagnostic builds (-DIE_STAMP) only: [block][wave][8] cycle sums
};

#ifdef IE_STAMP
#define IE_T(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define IE_TACC(i) do { unsigned long long now_; IE_T(now_); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define IE_TACC(i) do { } while (0)
#endif
#ifndef IE_K67X_HEAD
#define IE_K67X_HEAD 8      // MFMA gaps at the head of stage B's stream that carry no curve work
#endif
#ifndef IE_K67X_AHA
#define IE_K67X_AHA 3       // fragments read ahead of their MFMAs in stage A's / stage B's streams (each one in flight = 4 VGPRs; a stream alone is bound by the LDS
#endif                      // latency its reads in flight cover: stage A alone ran 22.6 cycles per MFMA at 3, stamps in profiles/r04/k67x_experiments.log)
#ifndef IE_K67X_C1POS
#define IE_K67X_C1POS 1     // where stage A's e_conv1 (gather, 4 MFMAs, hand-off) sits: 0 in front of the e_conv6 streams, 1 between the x2 and the x5 stream
#endif
#ifndef IE_K67X_BURSTPOS
#define IE_K67X_BURSTPOS 0      // where stage B's curve burst sits (both-tiles mapping): 0 head of the step, 1 between its two streams, 2 tail
#endif
#ifndef IE_K67X_EXPB
#define IE_K67X_EXPB 1      // stage B (both tiles): e_conv7's bias enters the tanh as a factor exp2(b) of exp2(acc) instead of a term of the exponent
#endif
#ifndef IE_K67X_BOTHA
#define IE_K67X_BOTHA 0     // experiment (correct results; 1): stage A with both cout fragments of e_conv6 per wave over a quarter of the strip + e_conv1 as ONE MFMA per tile over a
                            // 3-row window (4 instead of 12 MFMAs per wave and row).  Half of stage A's fragment reads and 8 MFMAs fewer, lower draw -- and the kernel 3 % SLOWER
                            // (7.93 against 7.72 ms; the model the same): the older stage's reads were never on the step's critical chain, its longer e_conv1 chain is
                            // (profiles/r04/k67x_experiments.log, 18).  Three of the twelve compiled forms spill 2 - 8 VGPRs in this mapping.
#endif
#ifndef IE_K67X_BOTH
#define IE_K67X_BOTH 1      // stage B's waves own BOTH row tiles of e_conv7 over a quarter of the strip: half the fragment reads, no exchange row between the tiles, the staged row a
                            // step earlier: 8.36 -> 8.02 ms per 64 frames same box (profiles/r04/k67x_experiments.log, 16).  0: the (row tile, strip half) mapping of the first K67X
#endif
#ifndef IE_K67X_AHB
#define IE_K67X_AHB 3
#endif
#ifndef IE_K67X_PACE_A
#define IE_K67X_PACE_A -1
#endif
#ifndef IE_K67X_BURST_FIRST
#define IE_K67X_BURST_FIRST 1      // the curve as one burst at the head of stage B's step (0: as 8-cycle items in the gaps of its MFMA stream).  At the head of a step the older stage
#endif                             // (A) takes every matrix-pipe slot it is ready for, so stage B's MFMAs would wait anyway: -6 % on the kernel (profiles/r04/k67x_experiments.log, 10)
#ifndef IE_K67X_UNPIN_A
#define IE_K67X_UNPIN_A 0
#endif
#ifndef IE_K67X_CONVERT_B
#define IE_K67X_CONVERT_B 0
#endif
#ifndef IE_K67X_IPG
#define IE_K67X_IPG 1       // gap items per MFMA gap
#endif
#ifndef IE_K67X_PRIO
#define IE_K67X_PRIO 0      // s_setprio level of stage B's waves (the younger half of the workgroup loses the vector-issue arbitration at equal priority)
#endif

template <typename TH> struct K67XGeom {
    static constexpr int PF = 3, NRA = PF + 1;           // x2 / x5 rows of DMA issued ahead, ring slots
    static constexpr int PFR = 2, NRAW = 8;              // raw image rows issued ahead; the raw row of step u is read by the conversion (step u - 1) and by the curve (step u + 4)
    static constexpr int NRI = IE_K67X_BOTHA ? 4 : 2, NRH = 3;      // converted image rows: BOTHA reads a 3-row window while the next row is being converted
    static constexpr int IMG_ROW_B = parts_of<TH>() * IMG_PART_B;
    static constexpr int YPLANE_B = 132 * 4, YSLOT_B = 4 * YPLANE_B;      // fp32 [colour (+ a spare plane for lane group kq = 3)][pixel q at float q + 2]
    static constexpr int OFF_X2 = 0, OFF_X5 = OFF_X2 + NRA * ROWH_B, OFF_H6 = OFF_X5 + NRA * ROWH_B, OFF_H1 = OFF_H6 + NRH * ROWH_B;
    static constexpr int OFF_RAW = OFF_H1 + NRH * ROWH_B, OFF_IMG = OFF_RAW + NRAW * RAW_SLOT_B, OFF_Y4 = OFF_IMG + NRI * IMG_ROW_B;
    static constexpr int OFF_OUT = OFF_Y4 + 2 * YSLOT_B, OFF_DUMP = OFF_OUT + 2 * YSLOT_B, OFF_BIAS = OFF_DUMP + 1024, LDS_B = OFF_BIAS + 128;      // (bias table: IE_K67X_BOTH)
    // f16: 66,560 + 49,920 + 16,384 + 2,176 + 4,224 + 4,224 + 1,024 = 144,512 bytes: one workgroup per CU
    // vector-memory operations of a stage-A wave per step, in issue order: [enhanced store (waves 0-2)] raw x 2, x2 x 2, x5 x 2.  At the barrier of step t the raw row
    // t + 1 (issued first in step t - PFR) and the x2 / x5 rows t (issued in step t - PF) must have landed: at most min(4 + (PFR - 1) S, (PF - 1) S) younger
    // operations may be outstanding, S = 6 or 7 (8 on the scalar store path: the count for S = 6 is safe for all)
    static constexpr int WAIT_A = (4 + (PFR - 1) * 6) < (PF - 1) * 6 ? (4 + (PFR - 1) * 6) : (PF - 1) * 6;
};

// MODE (what the kernel is compiled for; same arithmetic, same bits): 0 any mix of enhanced / adjust / u8, any width; 1 `enhanced` fp32 alone, W % 4 == 0 and 16-byte
// aligned (the throughput path: no output-mode branches); 2 enhanced + the 24 curve maps, W % 4 == 0 (what model.forward() returns); 3 the uint8 HWC image alone
// (the uint8 pipeline).
template <typename TA0, typename TA1, typename TH, int MODE>
__device__ __forceinline__ void k67x_body(const K67XParams& p) {
    using G = K67XGeom<TH>;
    using frag = ie_u32x4;
    constexpr int PF = G::PF, NRA = G::NRA, PFR = G::PFR, NRAW = G::NRAW, NRI = G::NRI, NRH = G::NRH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ie_fp16_saturate_on();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
#ifndef IE_PAIR_SAME
#define IE_PAIR_SAME 0    // experiment: 1 = the two waves of a SIMD (w and w + 4) belong to the SAME stage (stage A = waves 0, 1, 4, 5; stage B = waves 2, 3, 6, 7)
#endif
#if IE_PAIR_SAME
    const int stg = (wave >> 1) & 1, w4 = (wave & 1) | ((wave >> 2) << 1), cf = w4 & 1, hf = w4 >> 1;
#else
    const int stg = wave >> 2, w4 = wave & 3, cf = w4 & 1, hf = w4 >> 1;      // stage; stage A: 16-cout fragment / stage B: 16-row tile of e_conv7; half of the strip
#endif

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = p.n_begin + b;
    const int xc0 = strip * TS - MARG;                     // image column of computed column 0 (xc0 + MARG is a multiple of 4)
    const int H = p.H, W = p.W;
    const int ys = p.y_begin + chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, p.y_end);
    const int n_steps = (ye - ys) + (IE_K67X_BOTH ? 7 : 8);
    const size_t plane = (size_t)H * W;
    const size_t frame_b = plane * PXB;
#ifdef IE_STAMP
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_last = 0;
    IE_T(st_last);
#endif

    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000); };
    const RowRead rr = row_read(hf, p16, kq);

#ifdef IE_ONLY_STAGE      // register-pressure diagnostic (tools/kregs.sh ... -DIE_ONLY_STAGE=0|1): compile one stage's code only
    if (stg != IE_ONLY_STAGE) return;
#endif
    if (stg == 0) {
        // ================================================================ stage A ================================================================
        if constexpr (IE_K67X_BOTHA) {
        // ---- stage A with both cout fragments per wave: wave w4 = a QUARTER of the strip (pixels 32 w4 ..), e_conv6 with all 36 weight fragments (a fragment feeds six MFMAs),
        // e_conv1 as ONE K = 32 MFMA per tile and cout fragment over the 27 taps of a 3-row window (k = 8 ky + {(kx, c)} in lane groups 0-2, the three (kx = 2, B) taps in
        // lane group 3): 4 MFMAs per wave and row instead of 12, no rotating sets -- which is what makes room for e_conv6's second fragment.  The chores keep their lane maps.
        const auto rs2 = mk((const char*)p.x2 + (size_t)n * frame_b, frame_b);
        const auto rs5 = mk((const char*)p.x5 + (size_t)n * frame_b, frame_b);
        const auto rsI = mk((const char*)(p.img + (size_t)n * 3 * plane), 3 * plane * 4);
        const auto rsE = mk(p.enhanced ? (const void*)(p.enhanced + (size_t)n * 3 * plane) : (const void*)p.w7, p.enhanced ? 3 * plane * 4 : 0);
        frag wq0[18], wq1[18], w1q[2];
#pragma unroll
        for (int f = 0; f < 18; ++f) { wq0[f] = ((const frag*)p.w6)[f * 64 + lane]; wq1[f] = ((const frag*)p.w6)[(18 + f) * 64 + lane]; }
#pragma unroll
        for (int f = 0; f < 2; ++f) w1q[f] = ((const frag*)p.w1q)[f * 64 + lane];
        ie_f32x4 bias6a, bias6b, bias1a, bias1b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { bias6a[i] = p.b6[4 * kq + i]; bias6b[i] = p.b6[16 + 4 * kq + i]; bias1a[i] = p.b1[4 * kq + i]; bias1b[i] = p.b1[16 + 4 * kq + i]; }
        unsigned body_xoff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = (2 * w4 + j) * 16 + (lane >> 2);
            const int sc = (lane & 3) ^ (((q + 1) >> 1) & 3);
            const int x = xc0 + q;
            body_xoff[j] = (x >= 0 && x < W) ? (unsigned)(x * PXB + sc * 16) : OOB;
        }
        unsigned raw_off[2];
        raw_offsets(w4, lane, xc0, W, plane, raw_off);
        auto issue = [&](int t) __attribute__((always_inline)) {      // raw image row of step t - PF + 1 + PFR first, then the x2 / x5 rows of step t
            {
                const int tr = t - PF + 1 + PFR, yi = ys - 2 + tr;
                const unsigned rmask = (yi >= 0 && yi < H) ? 0u : OOB;
                const unsigned row_b = rmask ? 0u : (unsigned)yi * (unsigned)W * 4u;
                char* slot = smem + G::OFF_RAW + (tr % NRAW) * RAW_SLOT_B;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(slot + (4 * j + w4) * 256), 4, (int)((row_b + raw_off[j]) | rmask), 0, 0, 0);
            }
            const int y = ys - 2 + t;
            const unsigned rmask = (y >= 0 && y < H) ? 0u : OOB;
            const unsigned row_b = rmask ? 0u : (unsigned)y * (unsigned)W * PXB;
            char* s2 = smem + G::OFF_X2 + (t % NRA) * ROWH_B + PXB;
            char* s5 = smem + G::OFF_X5 + (t % NRA) * ROWH_B + PXB;
#pragma unroll
            for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(s2 + (2 * w4 + j) * 1024), 16, (int)((row_b + body_xoff[j]) | rmask), 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs5, (lds_ptr)(s5 + (2 * w4 + j) * 1024), 16, (int)((row_b + body_xoff[j]) | rmask), 0, 0, 0);
        };
        const int ci = w4 * 64 + lane;
        auto convert = [&](int t) __attribute__((always_inline)) {
            convert_row<TH>(smem + G::OFF_RAW + (t % NRAW) * RAW_SLOT_B, smem + G::OFF_IMG + (t % NRI) * G::IMG_ROW_B, smem + G::OFF_DUMP, ci, lane);
        };
        RowRead rq;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = 32 * w4 + p16 + kx;
            rq.base[kx] = P * PXB + ((kq ^ ((P >> 1) & 3)) << 4);
        }
        // hand-off: after the lane-pair exchange this lane holds, per cout fragment cf, the 16-byte chunk 2 cf + (kq >> 1) of ring pixel 32 w4 + 16 (kq & 1) + p16 + 1
        int emq[2];
        unsigned em_in;
        {
            const int P = 32 * w4 + 16 * (kq & 1) + p16 + 1, x = xc0 + P - 1;
#pragma unroll
            for (int cf = 0; cf < 2; ++cf) emq[cf] = P * PXB + (((2 * cf + (kq >> 1)) ^ ((P >> 1) & 3)) << 4);
            em_in = (x >= 0 && x < W) ? 1u : 0u;
        }
        // e_conv1's gathered fragment: read r (0..2) of tile j is the 8-byte pixel (column + r) of window row kq (lane groups 0-2) or the pixel (column + 2) of window row r (group 3)
        const bool is3 = kq == 3;
        const int gpx = (32 * w4 + p16) * IPB;        // tile j: + 16 pixels = 128 bytes
        const bool strip_inside = xc0 >= 0 && xc0 + TW <= W;
        unsigned est_off;
        int est_lds;
        {
            const int x = xc0 + MARG + 4 * lane;
            est_off = (w4 < 3 && lane < 31 && x < W) ? (unsigned)(((size_t)w4 * plane + x) * 4) : OOB;
            est_lds = (w4 < 3 ? w4 : 0) * G::YPLANE_B + 16 + 16 * (lane < 31 ? lane : 0);
        }
        constexpr bool M_ENH = MODE != 3, M_U8 = MODE == 0 || MODE == 3;
        const auto rsU = mk((M_U8 && p.u8) ? (const void*)(p.u8 + (size_t)n * 3 * plane) : (const void*)p.w7, (M_U8 && p.u8) ? 3 * plane : 0);
        const bool vec_store = MODE == 1 || MODE == 2 || (M_ENH && p.enhanced && (W & 3) == 0 && ((size_t)p.enhanced & 15) == 0);
        ie_f32x4 a0[4], a1[4], a2[4];      // e_conv6: index 2 * cout fragment + tile
#pragma unroll
        for (int j = 0; j < 4; ++j) { const ie_f32x4 bb = j < 2 ? bias6a : bias6b; a0[j] = bb; a1[j] = bb; a2[j] = bb; }
#pragma unroll
        for (int t = PF - 1 - PFR; t < PF; ++t) issue(t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // prologue barrier: raw row 0 has landed
        convert(0);
        auto put_q = [&](const ie_f32x4 (&acc)[4], char* dst, bool plain, bool keep) __attribute__((always_inline)) {
#pragma unroll
            for (int cf = 0; cf < 2; ++cf) {
                const ie_u32x2 a = {ie_pack2_relu<TH>(acc[2 * cf][0], acc[2 * cf][1]), ie_pack2_relu<TH>(acc[2 * cf][2], acc[2 * cf][3])};
                const ie_u32x2 b2 = {ie_pack2_relu<TH>(acc[2 * cf + 1][0], acc[2 * cf + 1][1]), ie_pack2_relu<TH>(acc[2 * cf + 1][2], acc[2 * cf + 1][3])};
                const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
                const bool k = plain || keep;
                *(ie_u32x4*)(dst + emq[cf]) = ie_u32x4{k ? sx[0] : 0u, k ? sy[0] : 0u, k ? sx[1] : 0u, k ? sy[1] : 0u};
            }
        };
        auto step_a = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            IE_TACC(3);
            ie_wait_vm_imm<G::WAIT_A>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            IE_TACC(0);
            if constexpr (M_ENH) {   // the row staged at step t - 1 leaves
                const int yo = ys - (IE_K67X_BOTH ? 7 : 8) + t;
                const bool row_ok = (yo >= ys) && (yo < ye);
                const char* os = smem + G::OFF_OUT + ((t + 1) & 1) * G::YSLOT_B;
                if (vec_store) {
                    const ie_u32x4 v = *(const ie_u32x4*)(os + est_lds);
                    const unsigned rmask = row_ok ? 0u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsE, (int)((est_off + (rmask ? 0u : (unsigned)yo * (unsigned)W * 4u)) | rmask), 0, 0);
                } else if (MODE == 0 && p.enhanced) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int q = lane + 64 * k, x = xc0 + q;
                        const float v = *(const float*)(os + (w4 < 3 ? w4 : 0) * G::YPLANE_B + (q + 2) * 4);
                        const bool ok = row_ok && w4 < 3 && q >= MARG && q < TW - MARG && x < W;
                        const unsigned off = ok ? (unsigned)(((size_t)w4 * plane + (size_t)yo * W + x) * 4) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsE, (int)off, 0, 0);
                    }
                }
            }
            if constexpr (M_U8) {
                if (p.u8 && w4 == 3) {
                    const int yo = ys - (IE_K67X_BOTH ? 7 : 8) + t;
                    const bool row_ok = (yo >= ys) && (yo < ye);
                    const char* os = smem + G::OFF_OUT + ((t + 1) & 1) * G::YSLOT_B;
                    const int xs = xc0 + MARG;
                    const int nbytes = row_ok ? min(TS, W - xs) * 3 : 0;
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    if ((W & 3) == 0) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int d = ln + 64 * k;
                            unsigned pk = 0;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int bb = 4 * d + e, pxr = bb / 3, c = bb - 3 * pxr;
                                const float v = *(const float*)(os + c * G::YPLANE_B + ((pxr < TS ? pxr : 0) + MARG + 2) * 4);
                                pk |= (unsigned)ie_quant_u8(v) << (8 * e);
                            }
                            const unsigned off = (4 * d < nbytes) ? (unsigned)(((size_t)yo * W + xs) * 3 + 4 * d) : OOB;
                            __builtin_amdgcn_raw_buffer_store_b32(pk, rsU, (int)off, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            const int bb = ln + 64 * k, pxr = bb / 3, c = bb - 3 * pxr;
                            const float v = *(const float*)(os + c * G::YPLANE_B + ((pxr < TS ? pxr : 0) + MARG + 2) * 4);
                            const unsigned off = (bb < nbytes) ? (unsigned)(((size_t)yo * W + xs) * 3 + bb) : OOB;
                            __builtin_amdgcn_raw_buffer_store_b8(ie_quant_u8(v), rsU, (int)off, 0, 0);
                        }
                    }
                }
            }
            IE_TACC(4);
            const int m = ys - 3 + t;                                          // the x1 / x6 row this step finishes
            const bool row_in = m >= 0 && m < H, plain = strip_inside && row_in, keep = row_in && em_in;
#if IE_K67X_C1POS == 0
            {   // ---- e_conv1: converted rows t - 2, t - 1, t (image rows m - 1, m, m + 1) -> x1 row m, four MFMAs ----
                const int s0 = ((t + NRI - 2) % NRI) * G::IMG_ROW_B, s1 = ((t + NRI - 1) % NRI) * G::IMG_ROW_B, s2 = (t % NRI) * G::IMG_ROW_B;
                const int srow = kq == 0 ? s0 : (kq == 1 ? s1 : s2);
                const int ad0 = G::OFF_IMG + gpx + (is3 ? s0 + 2 * IPB : srow), ad1 = G::OFF_IMG + gpx + (is3 ? s1 + 2 * IPB : srow + IPB),
                          ad2 = G::OFF_IMG + gpx + (is3 ? s2 + 2 * IPB : srow + 2 * IPB);
                ie_f32x4 e[4];
#pragma unroll
                for (int part = 0; part < parts_of<TH>(); ++part) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const ie_u32x2 r0 = *(const ie_u32x2*)(smem + ad0 + part * IMG_PART_B + 128 * j), r1 = *(const ie_u32x2*)(smem + ad1 + part * IMG_PART_B + 128 * j),
                                       r2 = *(const ie_u32x2*)(smem + ad2 + part * IMG_PART_B + 128 * j);
                        ie_u32x4 fr;
                        fr[0] = is3 ? ((r0[1] & 0xffffu) | (r1[1] << 16)) : r0[0];
                        fr[1] = is3 ? (r2[1] & 0xffffu) : ((r0[1] & 0xffffu) | (r1[0] << 16));
                        fr[2] = is3 ? 0u : ((r1[0] >> 16) | (r1[1] << 16));
                        fr[3] = is3 ? 0u : r2[0];
                        e[j] = ie_mfma16_raw<TH>(w1q[0], fr, part == 0 ? bias1a : e[j]);
                        e[2 + j] = ie_mfma16_raw<TH>(w1q[1], fr, part == 0 ? bias1b : e[2 + j]);
                    }
                }
                put_q(e, smem + G::OFF_H1 + (t % NRH) * ROWH_B, plain, keep);
            }
#endif
            mfma_row_both<TA0, IE_K67X_AHBOTH, false>(smem, smem + G::OFF_X2 + (t % NRA) * ROWH_B, rq, wq0, wq1, accN, accM, accO, true, bias6a, bias6b);
            IE_TACC(2);
#if IE_K67X_C1POS == 1
            {   // ---- e_conv1: converted rows t - 2, t - 1, t (image rows m - 1, m, m + 1) -> x1 row m, four MFMAs ----
                const int s0 = ((t + NRI - 2) % NRI) * G::IMG_ROW_B, s1 = ((t + NRI - 1) % NRI) * G::IMG_ROW_B, s2 = (t % NRI) * G::IMG_ROW_B;
                const int srow = kq == 0 ? s0 : (kq == 1 ? s1 : s2);
                const int ad0 = G::OFF_IMG + gpx + (is3 ? s0 + 2 * IPB : srow), ad1 = G::OFF_IMG + gpx + (is3 ? s1 + 2 * IPB : srow + IPB),
                          ad2 = G::OFF_IMG + gpx + (is3 ? s2 + 2 * IPB : srow + 2 * IPB);
                ie_f32x4 e[4];
#pragma unroll
                for (int part = 0; part < parts_of<TH>(); ++part) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const ie_u32x2 r0 = *(const ie_u32x2*)(smem + ad0 + part * IMG_PART_B + 128 * j), r1 = *(const ie_u32x2*)(smem + ad1 + part * IMG_PART_B + 128 * j),
                                       r2 = *(const ie_u32x2*)(smem + ad2 + part * IMG_PART_B + 128 * j);
                        ie_u32x4 fr;
                        fr[0] = is3 ? ((r0[1] & 0xffffu) | (r1[1] << 16)) : r0[0];
                        fr[1] = is3 ? (r2[1] & 0xffffu) : ((r0[1] & 0xffffu) | (r1[0] << 16));
                        fr[2] = is3 ? 0u : ((r1[0] >> 16) | (r1[1] << 16));
                        fr[3] = is3 ? 0u : r2[0];
                        e[j] = ie_mfma16_raw<TH>(w1q[0], fr, part == 0 ? bias1a : e[j]);
                        e[2 + j] = ie_mfma16_raw<TH>(w1q[1], fr, part == 0 ? bias1b : e[2 + j]);
                    }
                }
                put_q(e, smem + G::OFF_H1 + (t % NRH) * ROWH_B, plain, keep);
            }
#endif
            mfma_row_both<TA1, IE_K67X_AHBOTH, false>(smem, smem + G::OFF_X5 + (t % NRA) * ROWH_B, rq, wq0 + 9, wq1 + 9, accN, accM, accO, false, bias6a, bias6b);
            IE_TACC(1);
            issue(t + PF);
            IE_TACC(5);
            convert(t + 1);
            IE_TACC(6);
            put_q(accO, smem + G::OFF_H6 + (t % NRH) * ROWH_B, plain, keep);
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_a(t, a0, a1, a2);
            if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0);
        }
        } else {
        const auto rs2 = mk((const char*)p.x2 + (size_t)n * frame_b, frame_b);
        const auto rs5 = mk((const char*)p.x5 + (size_t)n * frame_b, frame_b);
        const auto rsI = mk((const char*)(p.img + (size_t)n * 3 * plane), 3 * plane * 4);
        const auto rsE = mk(p.enhanced ? (const void*)(p.enhanced + (size_t)n * 3 * plane) : (const void*)p.w7, p.enhanced ? 3 * plane * 4 : 0);
        frag wv[18], w1[3];
#pragma unroll
        for (int f = 0; f < 18; ++f) wv[f] = ((const frag*)p.w6)[(cf * 18 + f) * 64 + lane];
#pragma unroll
        for (int f = 0; f < 3; ++f) w1[f] = ((const frag*)p.w1)[(cf * 3 + f) * 64 + lane];
        ie_f32x4 bias6, bias1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { bias6[i] = p.b6[16 * cf + 4 * kq + i]; bias1[i] = p.b1[16 * cf + 4 * kq + i]; }
        // x2 / x5 rows: piece m = 2 w4 + j (16 pixels x 64 bytes) lands at ring pixels 16 m + 1 ..; lane -> (pixel, LDS chunk position); the source chunk is the
        // position's un-swizzled one
        unsigned body_xoff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = (2 * w4 + j) * 16 + (lane >> 2);
            const int s = (lane & 3) ^ (((q + 1) >> 1) & 3);
            const int x = xc0 + q;
            body_xoff[j] = (x >= 0 && x < W) ? (unsigned)(x * PXB + s * 16) : OOB;
        }
        unsigned raw_off[2];
        raw_offsets(w4, lane, xc0, W, plane, raw_off);
        auto issue = [&](int t) __attribute__((always_inline)) {      // raw image row of step t - PF + 1 + PFR first, then the x2 / x5 rows of step t
            {
                const int tr = t - PF + 1 + PFR, yi = ys - 2 + tr;
                const unsigned rmask = (yi >= 0 && yi < H) ? 0u : OOB;
                const unsigned row_b = rmask ? 0u : (unsigned)yi * (unsigned)W * 4u;
                char* slot = smem + G::OFF_RAW + (tr % NRAW) * RAW_SLOT_B;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(slot + (4 * j + w4) * 256), 4, (int)((row_b + raw_off[j]) | rmask), 0, 0, 0);
            }
            const int y = ys - 2 + t;
            const unsigned rmask = (y >= 0 && y < H) ? 0u : OOB;
            const unsigned row_b = rmask ? 0u : (unsigned)y * (unsigned)W * PXB;
            char* s2 = smem + G::OFF_X2 + (t % NRA) * ROWH_B + PXB;
            char* s5 = smem + G::OFF_X5 + (t % NRA) * ROWH_B + PXB;
#pragma unroll
            for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(s2 + (2 * w4 + j) * 1024), 16, (int)((row_b + body_xoff[j]) | rmask), 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs5, (lds_ptr)(s5 + (2 * w4 + j) * 1024), 16, (int)((row_b + body_xoff[j]) | rmask), 0, 0, 0);
        };
        const int ci = w4 * 64 + lane;
        auto convert = [&](int t) __attribute__((always_inline)) {
            convert_row<TH>(smem + G::OFF_RAW + (t % NRAW) * RAW_SLOT_B, smem + G::OFF_IMG + (t % NRI) * G::IMG_ROW_B, smem + G::OFF_DUMP, ci, lane);
        };
        int rdi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rdi[j] = (16 * (4 * hf + j) + p16 + 2 * kq) * IPB;
        const int em_off = put_off(cf, hf, p16, kq, 1);
        unsigned em_in = 0;
        {
            const int q = 16 * (4 * hf + (kq & 1)) + p16;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int x = xc0 + q + 32 * jj;
                em_in |= (x >= 0 && x < W) ? (1u << jj) : 0u;
            }
        }
        const bool strip_inside = xc0 >= 0 && xc0 + TW <= W;       // every computed column lies inside the image: no column masks
        // enhanced, W % 4 == 0: wave w4 < 3 stores colour w4; lane l < 31: pixels 4 l .. 4 l + 3 of the strip's 124 stored ones
        unsigned est_off;
        int est_lds;
        {
            const int x = xc0 + MARG + 4 * lane;
            est_off = (w4 < 3 && lane < 31 && x < W) ? (unsigned)(((size_t)w4 * plane + x) * 4) : OOB;
            est_lds = (w4 < 3 ? w4 : 0) * G::YPLANE_B + 16 + 16 * (lane < 31 ? lane : 0);
        }
        constexpr bool M_ENH = MODE != 3, M_U8 = MODE == 0 || MODE == 3;
        const auto rsU = mk((M_U8 && p.u8) ? (const void*)(p.u8 + (size_t)n * 3 * plane) : (const void*)p.w7, (M_U8 && p.u8) ? 3 * plane : 0);
        const bool vec_store = MODE == 1 || MODE == 2 || (M_ENH && p.enhanced && (W & 3) == 0 && ((size_t)p.enhanced & 15) == 0);
        ie_f32x4 a0[4], a1[4], a2[4], e0[4], e1[4], e2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a0[j] = bias6; a1[j] = bias6; a2[j] = bias6; e0[j] = bias1; e1[j] = bias1; e2[j] = bias1; }

        // prologue: raw rows 0 .. PFR and x2 / x5 rows 0 .. PF - 1 (issue(t) carries the raw row t - PF + 1 + PFR)
#pragma unroll
        for (int t = PF - 1 - PFR; t < PF; ++t) issue(t);      // (t < 0: a raw row only -- its x2 / x5 rows land in slots that row PF + t overwrites before any read)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // prologue barrier: raw row 0 has landed
        convert(0);
        // step t: image / x2 / x5 row ys - 2 + t; the x1 and x6 rows ys - 3 + t are finished and handed over in slot t % 3; the staged output row ys - 8 + t leaves
        auto step_a = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], ie_f32x4 (&eN)[4], ie_f32x4 (&eM)[4], ie_f32x4 (&eO)[4]) __attribute__((always_inline)) {
            IE_TACC(3);
            ie_wait_vm_imm<G::WAIT_A>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            IE_TACC(0);
            if constexpr (!(IE_XEXP & 16) && M_ENH) {   // the row staged by the tile-1 waves at step t - 1 leaves
                const int yo = ys - (IE_K67X_BOTH ? 7 : 8) + t;
                const bool row_ok = (yo >= ys) && (yo < ye);
                const char* os = smem + G::OFF_OUT + ((t + 1) & 1) * G::YSLOT_B;
                if (vec_store) {
                    const ie_u32x4 v = *(const ie_u32x4*)(os + est_lds);
                    const unsigned rmask = row_ok ? 0u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsE, (int)((est_off + (rmask ? 0u : (unsigned)yo * (unsigned)W * 4u)) | rmask), 0, 0);
                } else if (MODE == 0 && p.enhanced) {      // (any width / alignment: dword stores)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int q = lane + 64 * k, x = xc0 + q;
                        const float v = *(const float*)(os + (w4 < 3 ? w4 : 0) * G::YPLANE_B + (q + 2) * 4);
                        const bool ok = row_ok && w4 < 3 && q >= MARG && q < TW - MARG && x < W;
                        const unsigned off = ok ? (unsigned)(((size_t)w4 * plane + (size_t)yo * W + x) * 4) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsE, (int)off, 0, 0);
                    }
                }
            }
            if constexpr (M_U8 && !(IE_XEXP & 16)) {
                if (p.u8 && w4 == 3) {        // HWC bytes: 372 per strip row: wave 3 (it stores no colour plane) quantises the staged fp32 values (save_image rule)
                    const int yo = ys - (IE_K67X_BOTH ? 7 : 8) + t;
                    const bool row_ok = (yo >= ys) && (yo < ye);
                    const char* os = smem + G::OFF_OUT + ((t + 1) & 1) * G::YSLOT_B;
                    const int xs = xc0 + MARG;
                    const int nbytes = row_ok ? min(TS, W - xs) * 3 : 0;
                    int ln = lane;
                    asm volatile("" : "+v"(ln));      // the byte -> (pixel, colour) arithmetic below is loop-invariant: recomputed per step rather than carried in 8+ VGPRs across the MFMAs
                    if ((W & 3) == 0) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int d = ln + 64 * k;
                            unsigned pk = 0;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int bb = 4 * d + e, pxr = bb / 3, c = bb - 3 * pxr;
                                const float v = *(const float*)(os + c * G::YPLANE_B + ((pxr < TS ? pxr : 0) + MARG + 2) * 4);
                                pk |= (unsigned)ie_quant_u8(v) << (8 * e);
                            }
                            const unsigned off = (4 * d < nbytes) ? (unsigned)(((size_t)yo * W + xs) * 3 + 4 * d) : OOB;
                            __builtin_amdgcn_raw_buffer_store_b32(pk, rsU, (int)off, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            const int bb = ln + 64 * k, pxr = bb / 3, c = bb - 3 * pxr;
                            const float v = *(const float*)(os + c * G::YPLANE_B + ((pxr < TS ? pxr : 0) + MARG + 2) * 4);
                            const unsigned off = (bb < nbytes) ? (unsigned)(((size_t)yo * W + xs) * 3 + bb) : OOB;
                            __builtin_amdgcn_raw_buffer_store_b8(ie_quant_u8(v), rsU, (int)off, 0, 0);
                        }
                    }
                }
            }
            IE_TACC(4);
            if constexpr (!(IE_XEXP & 2)) conv1_row<TH>(smem + G::OFF_IMG + (t % NRI) * G::IMG_ROW_B, rdi, w1, bias1, eN, eM, eO);
#if IE_K67X_PACE_A >= 0
            // experiment: stage A leaves pipe slots to stage B by not being ready for its next MFMA (IE_K67X_PACE_A = the s_nop argument after every MFMA)
            auto nofill = [&](int) __attribute__((always_inline)) { asm volatile("s_nop %0" ::"n"(IE_K67X_PACE_A)); };
#else
            auto nofill = [&](int) __attribute__((always_inline)) {};
#endif
            if constexpr (IE_XEXP & 16384) {
                // timing experiment (wrong results): stage A's e_conv6 as 36 v_mfma_f32_32x32x16 on 12 of its 24 LDS fragments per step (the instruction a lone wave sustains
                // back to back) -- same matrix-pipe cycles as the 72 16x16x32
                ie_f32x16 big[3];
#pragma unroll
                for (int i = 0; i < 16; ++i) { big[0][i] = accN[i >> 2][i & 3]; big[1][i] = accM[i >> 2][i & 3]; big[2][i] = accO[i >> 2][i & 3]; }
                const char* r2 = smem + G::OFF_X2 + (t % NRA) * ROWH_B;
                const char* r5 = smem + G::OFF_X5 + (t % NRA) * ROWH_B;
                ie_u32x4 fx[12];
#pragma unroll
                for (int g = 0; g < 3; ++g) fx[g] = *(const ie_u32x4*)(r2 + rr.base[g % 3] + 1024 * (g & 3));
                IE_SB();
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    if (g + 3 < 12) fx[g + 3] = *(const ie_u32x4*)((g + 3 < 6 ? r2 : r5) + rr.base[(g + 3) % 3] + 1024 * ((g + 3) & 3));
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        big[ky] = ie_elem<TH>::mfma32(__builtin_bit_cast(typename ie_elem<TH>::frag, wv[(g + ky) % 18]), __builtin_bit_cast(typename ie_elem<TH>::frag, fx[g]), big[ky]);
                        IE_SB();
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) { accN[i >> 2][i & 3] = big[0][i]; accM[i >> 2][i & 3] = big[1][i]; accO[i >> 2][i & 3] = big[2][i]; }
            } else if constexpr (IE_K67X_UNPIN_A) {
            mfma_row_free<TA0>(smem + G::OFF_X2 + (t % NRA) * ROWH_B, rr, wv, accN, accM, accO, true, bias6);
            IE_TACC(2);
            mfma_row_free<TA1>(smem + G::OFF_X5 + (t % NRA) * ROWH_B, rr, wv + 9, accN, accM, accO, false, bias6);
            } else if constexpr (!(IE_XEXP & 32)) {
            mfma_row<TA0, IE_K67X_AHA>(smem + G::OFF_X2 + (t % NRA) * ROWH_B, rr, wv, accN, accM, accO, true, bias6, nofill);
            IE_TACC(2);
            mfma_row<TA1, IE_K67X_AHA>(smem + G::OFF_X5 + (t % NRA) * ROWH_B, rr, wv + 9, accN, accM, accO, false, bias6, nofill);
            }
            IE_TACC(1);
            if constexpr (!(IE_XEXP & 8)) issue(t + PF);
            IE_TACC(5);
            if constexpr (!(IE_XEXP & 2) && !IE_K67X_CONVERT_B) convert(t + 1);
            IE_TACC(6);
            const int m = ys - 3 + t;
            if constexpr (!(IE_XEXP & 4)) {
            if (strip_inside && m >= 0 && m < H) {
                put_row<TH, false>(eO, smem + G::OFF_H1 + (t % NRH) * ROWH_B, em_off, 3u);
                put_row<TH, false>(accO, smem + G::OFF_H6 + (t % NRH) * ROWH_B, em_off, 3u);
            } else {
                const unsigned keep = (m >= 0 && m < H) ? em_in : 0u;
                put_row<TH, true>(eO, smem + G::OFF_H1 + (t % NRH) * ROWH_B, em_off, keep);
                put_row<TH, true>(accO, smem + G::OFF_H6 + (t % NRH) * ROWH_B, em_off, keep);
            }
            }
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_a(t, a0, a1, a2, e0, e1, e2);
            if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1, e2, e0, e1);
            if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0, e1, e2, e0);
        }
        }
    } else if constexpr (IE_K67X_BOTH) {
        // ===================================================== stage B, both row tiles per wave =====================================================
        // wave w4 = quarter of the strip (pixels 32 w4 .. 32 w4 + 31 = tiles j = 0, 1); it holds all 36 weight fragments of e_conv7, so a fragment feeds six MFMAs, and the
        // eight curve iterations of a pixel run in ONE lane: no exchange row, no tile-1 lag (the staged row leaves one step earlier: n_steps = rows + 7)
        constexpr bool M_ADJ = MODE == 0 || MODE == 2;
        const auto rsJ = mk((M_ADJ && p.adjust) ? (const void*)(p.adjust + (size_t)n * 24 * plane) : (const void*)p.w7, (M_ADJ && p.adjust) ? 24 * plane * 4 : 0);
        frag wv0[18], wv1[18];
#pragma unroll
        for (int f = 0; f < 18; ++f) { wv0[f] = ((const frag*)p.w7)[f * 64 + lane]; wv1[f] = ((const frag*)p.w7)[(18 + f) * 64 + lane]; }
        // e_conv7's bias: a 32-float table in LDS (OFF_BIAS), read back by (tile, kq) at the head of each curve burst -- not 8 registers held across the streams
        if (w4 == 0 && lane < 32) *(float*)(smem + G::OFF_BIAS + 4 * lane) = IE_K67X_EXPB ? __builtin_amdgcn_exp2f(p.b7[lane]) : p.b7[lane];      // (EXPB: exp2(acc + b) + 1 = fma(exp2(acc), exp2(b), 1): one instruction less per tanh)
        const int bias_off = G::OFF_BIAS + 16 * kq;
        RowRead rq;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = 32 * w4 + p16 + kx;
            rq.base[kx] = P * PXB + ((kq ^ ((P >> 1) & 3)) << 4);
        }
        const int q0 = 32 * w4 + p16;
        const int yout_off = kq * G::YPLANE_B + (q0 + 2) * 4;
        const int yin_off = (kq < 3 ? kq : 2) * (IPX * 4) + (q0 + 1) * 4;
        unsigned aj_ok = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + 16 * j, x = xc0 + q;
            aj_ok |= (kq < 3 && q >= MARG && q < TW - MARG && x < W) ? (1u << j) : 0u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // prologue barrier
        ie_f32x4 a0[4], a1[4], a2[4], a3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const ie_f32x4 b = {0.f, 0.f, 0.f, 0.f}; a0[j] = b; a1[j] = b; a2[j] = b; a3[j] = b; }
        // step t: the x1 / x6 rows ys - 4 + t (stage A's step t - 1) finish e_conv7 row ys - 5 + t; accF = the row finished at step t - 1 (ys - 6 + t): its whole curve runs now
        auto step_b = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], const ie_f32x4 (&accF)[4]) __attribute__((always_inline)) {
            IE_TACC(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            IE_TACC(0);
            const char* yin = smem + G::OFF_RAW + ((t + 4) % NRAW) * RAW_SLOT_B;
            char* yout = smem + G::OFF_OUT + (t & 1) * G::YSLOT_B;
            float v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) v[j] = *(const float*)(yin + yin_off + 64 * j);
            unsigned aj_base = OOB;
            if constexpr (M_ADJ) {
                const int yc = ys - 6 + t;
                if ((MODE == 2 || p.adjust) && yc >= ys && yc < ye) aj_base = (unsigned)(((size_t)kq * plane + (size_t)yc * W + xc0 + q0) * 4);
            }
            auto curve_burst = [&]() __attribute__((always_inline)) {
            const ie_f32x4 bz[2] = {*(const ie_f32x4*)(smem + bias_off), *(const ie_f32x4*)(smem + bias_off + 64)};
            IE_TACC(6);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float e;
                    if constexpr (IE_K67X_EXPB) e = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(accF[2 * (kk >> 2) + j][kk & 3]), bz[kk >> 2][kk & 3], 1.0f));
                    else e = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(accF[2 * (kk >> 2) + j][kk & 3] + bz[kk >> 2][kk & 3]) + 1.0f);
                    e = __builtin_fmaf(e, -2.0f, 1.0f);                                        // r_kk
                    if constexpr (M_ADJ) {
                        const unsigned off = (aj_base != OOB && ((aj_ok >> j) & 1u)) ? aj_base + (unsigned)(64 * j) + (unsigned)((size_t)(3 * kk) * plane * 4) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e), rsJ, (int)off, 0, 0);
                    }
                    const float u = __builtin_fmaf(v[j], v[j], -v[j]);
                    v[j] = __builtin_fmaf(e, u, v[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) *(float*)(yout + yout_off + 64 * j) = v[j];
            };
            if constexpr (IE_K67X_BURSTPOS == 0) curve_burst();
            IE_SB();
            const int hs = ((t + 2) % NRH) * ROWH_B;
            mfma_row_both<TH>(smem, smem + G::OFF_H1 + hs, rq, wv0, wv1, accN, accM, accO, true);
            IE_TACC(5);
            if constexpr (IE_K67X_BURSTPOS == 1) { curve_burst(); IE_SB(); }
            mfma_row_both<TH>(smem, smem + G::OFF_H6 + hs, rq, wv0 + 9, wv1 + 9, accN, accM, accO, false);
            if constexpr (IE_K67X_BURSTPOS == 2) curve_burst();
            IE_TACC(1);
        };
        for (int t = 0; t < n_steps; t += 4) {
            step_b(t, a0, a1, a2, a3);
            if (t + 1 < n_steps) step_b(t + 1, a3, a0, a1, a2);
            if (t + 2 < n_steps) step_b(t + 2, a2, a3, a0, a1);
            if (t + 3 < n_steps) step_b(t + 3, a1, a2, a3, a0);
        }
    } else {
        // ================================================================ stage B ================================================================
        const int cj = cf;                                  // this wave's 16-row tile of e_conv7 = curve iterations 4 cj .. 4 cj + 3; tile 1 runs one step behind
        constexpr bool M_ADJ = MODE == 0 || MODE == 2;
        const auto rsJ = mk((M_ADJ && p.adjust) ? (const void*)(p.adjust + (size_t)n * 24 * plane) : (const void*)p.w7, (M_ADJ && p.adjust) ? 24 * plane * 4 : 0);
        frag wv[18];
#pragma unroll
        for (int f = 0; f < 18; ++f) wv[f] = ((const frag*)p.w7)[(cj * 18 + f) * 64 + lane];
        ie_f32x4 bias7;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias7[i] = p.b7[16 * cj + 4 * kq + i];
        // where this lane's curve value of tile j comes from / goes to: the raw row's plane kq at ring pixel q + 1 (tile 0), the exchange / output rows at float q + 2
        // (tile j: + 64 j bytes, an immediate)
        const int q0 = 64 * hf + p16;
        const int yout_off = kq * G::YPLANE_B + (q0 + 2) * 4;                                    // (kq = 3: the spare plane)
        const int yin_off = cj ? yout_off : (kq < 3 ? kq : 2) * (IPX * 4) + (q0 + 1) * 4;
        // the 24 curve maps: this lane's tile j is valid when its column is a stored one inside the image (bit j)
        unsigned aj_ok = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + 16 * j, x = xc0 + q;
            aj_ok |= (kq < 3 && q >= MARG && q < TW - MARG && x < W) ? (1u << j) : 0u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // prologue barrier (stage A converts the first image row behind it)
#if IE_K67X_PRIO
        __builtin_amdgcn_s_setprio(IE_K67X_PRIO);
#endif

        ie_f32x4 a0[4], a1[4], a2[4], a3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a0[j] = bias7; a1[j] = bias7; a2[j] = bias7; a3[j] = bias7; }

        // step t (tile 0; tile 1 does the same one step later): the x1 / x6 rows ys - 4 + t (stage A's step t - 1) finish e_conv7 row ys - 5 + t; accF = the row
        // finished at step t - 1 (ys - 6 + t): its curve runs now
        auto step_b = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], const ie_f32x4 (&accF)[4]) __attribute__((always_inline)) {
            IE_TACC(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this stage issues no loads; its stores are only waited for at the end)
            __builtin_amdgcn_s_barrier();
            IE_TACC(0);
            if constexpr (IE_K67X_CONVERT_B)       // experiment: the image row conversion at the head of stage B's step instead of in stage A's tail
                convert_row<TH>(smem + G::OFF_RAW + ((t + 1) % NRAW) * RAW_SLOT_B, smem + G::OFF_IMG + ((t + 1) % NRI) * G::IMG_ROW_B, smem + G::OFF_DUMP, w4 * 64 + lane, lane);
            IE_TACC(4);
            // the curve's running value of the row in accF: row ys - 6 + t - cj = raw row of step t - 4 - cj; tile 1 reads what tile 0 left at step t - 1
            const char* yin = cj ? smem + G::OFF_Y4 + ((t + 1) & 1) * G::YSLOT_B : smem + G::OFF_RAW + ((t + 4) % NRAW) * RAW_SLOT_B;
            char* yout = cj ? smem + G::OFF_OUT + (t & 1) * G::YSLOT_B : smem + G::OFF_Y4 + (t & 1) * G::YSLOT_B;
            float v[4], e[4], u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = *(const float*)(yin + yin_off + 64 * j); e[j] = 0.f; u[j] = 0.f; }
            // the 24 curve maps (optional output) leave straight from the lane that computed them: NCHW fp32, plane 3 (4 cj + i) + colour
            unsigned aj_base = OOB;        // byte offset of curve map 12 cj of this lane's colour at tile 0's pixel; tile j: + 64 j, map k: + 3 k planes
            if constexpr (M_ADJ) {
                const int yc = ys - 6 + t - cj;
                if ((MODE == 2 || p.adjust) && yc >= ys && yc < ye) aj_base = (unsigned)(((size_t)(12 * cj + kq) * plane + (size_t)yc * W + xc0 + q0) * 4);
            }
            IE_TACC(6);
            // The 96 tanh / curve operations (per curve iteration k and pixel tile j: E = exp2, A = + 1, R = rcp, F = r = 1 - 2 R, U = v v - v, V = v + r u;
            // tanh(x) = 1 - 2 / (exp(2x) + 1), ie_tanh_fast) as 64 gap items of 8 issue cycles each -- ONE transcendental or TWO plain vector instructions: an
            // MFMA holds the SIMD's vector issue for 8 of its 16 cycles, so that is what a gap hides; two transcendentals in one gap (the first build's 4 fillers per
            // 3 MFMAs) cost twice their issue time (profiles/r04/k67x_experiments.log).  Item i of iteration k: 0-3 E(j = i), 4-5 A pairs, 6-9 R(j = i - 6), 10-11 F
            // pairs, 12-13 U pairs, 14-15 V pairs; gap s of the step's 72 carries item s - IE_K67X_HEAD (the head of a segment is the worst place for vector work).
            auto op = [&](int k, int w, int j) __attribute__((always_inline)) {
                if constexpr (IE_XEXP & 128) { if (w == 0 || w == 2) return; }       // no transcendentals
                if constexpr (IE_XEXP & 256) { if (w != 0 && w != 2) return; }       // transcendentals only
                if constexpr (IE_XEXP & 512) {                                       // transcendentals replaced by plain vector instructions
                    if (w == 0) { e[j] = __builtin_fmaf(accF[j][k], 0.25f, 1.0f); return; }
                    if (w == 2) { e[j] = __builtin_fmaf(e[j], 0.25f, 1.0f); return; }
                }
                if (w == 0) e[j] = __builtin_amdgcn_exp2f(accF[j][k]);
                if (w == 1) e[j] = e[j] + 1.0f;
                if (w == 2) e[j] = __builtin_amdgcn_rcpf(e[j]);
                if (w == 3) {
                    e[j] = __builtin_fmaf(e[j], -2.0f, 1.0f);       // r_{4 cj + k}
                    if constexpr (M_ADJ) {                            // (an absent `adjust` has an empty descriptor range and out-of-range offsets: the store is dropped)
                        const unsigned off = (aj_base != OOB && ((aj_ok >> j) & 1u)) ? aj_base + (unsigned)(64 * j) + (unsigned)((size_t)(3 * k) * plane * 4) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e[j]), rsJ, (int)off, 0, 0);
                    }
                }
                if (w == 4) u[j] = __builtin_fmaf(v[j], v[j], -v[j]);
                if (w == 5) v[j] = __builtin_fmaf(e[j], u[j], v[j]);
            };
            auto item = [&](int it) __attribute__((always_inline)) {
                if (it < 0 || it >= 64) return;
                if constexpr (IE_XEXP & 2048) {       // experiment: 16 plain adds instead of the 96 curve operations (keeps the MFMAs alive)
                    if ((it & 3) == 0) { const int k = it >> 4, j = (it >> 2) & 3; v[j] += accF[j][k]; }
                    return;
                }
                const int k = it >> 4, i = it & 15;
                if (i < 4) op(k, 0, i);
                else if (i < 6) { op(k, 1, 2 * (i - 4)); op(k, 1, 2 * (i - 4) + 1); }
                else if (i < 10) op(k, 2, i - 6);
                else { const int w = 3 + ((i - 10) >> 1), j0 = 2 * ((i - 10) & 1); op(k, w, j0); op(k, w, j0 + 1); }
            };
            if constexpr (IE_K67X_BURST_FIRST) {       // the whole curve as a burst BEFORE the stage's MFMAs: at the head of a step the older stage has the matrix pipe anyway
#pragma unroll
                for (int it = 0; it < 64; ++it) item(it);
                IE_SB();
            }
            auto gap = [&](int s) __attribute__((always_inline)) {
                if constexpr ((IE_XEXP & 1) || (IE_XEXP & 1024) || IE_K67X_BURST_FIRST) return;
#pragma unroll
                for (int r = 0; r < IE_K67X_IPG; ++r) item((s - IE_K67X_HEAD) * IE_K67X_IPG + r);
            };
            // the x1 / x6 rows handed over by stage A at step t - 1 - cj into the three live sets: two 36-MFMA streams
            const int hs = ((t + 2 - cj) % NRH) * ROWH_B;
            auto fill1 = [&](int s) __attribute__((always_inline)) { gap(s); };
            auto fill6 = [&](int s) __attribute__((always_inline)) { gap(s + 36); };
            if constexpr (IE_XEXP & 8192) {
                // timing experiment (wrong results): stage B's arithmetic as 36 v_mfma_f32_32x32x16 on the same number of LDS fragments (12 per step, two thirds of them
                // dummies of the first row's) -- the same matrix-pipe cycles as 72 16x16x32, but a single wave sustains the 32-cycle instruction back to back
                ie_f32x16 big[3];
#pragma unroll
                for (int i = 0; i < 16; ++i) { big[0][i] = accN[i >> 2][i & 3]; big[1][i] = accM[i >> 2][i & 3]; big[2][i] = accO[i >> 2][i & 3]; }
                const char* r1 = smem + G::OFF_H1 + hs;
                const char* r6 = smem + G::OFF_H6 + hs;
                ie_u32x4 fx[12];
#pragma unroll
                for (int g = 0; g < 3; ++g) fx[g] = *(const ie_u32x4*)(r1 + rr.base[g % 3] + 1024 * (g & 3));
                IE_SB();
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    if (g + 3 < 12) fx[g + 3] = *(const ie_u32x4*)((g + 3 < 6 ? r1 : r6) + rr.base[(g + 3) % 3] + 1024 * ((g + 3) & 3));
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        big[ky] = ie_elem<TH>::mfma32(__builtin_bit_cast(typename ie_elem<TH>::frag, wv[(g + ky) % 18]), __builtin_bit_cast(typename ie_elem<TH>::frag, fx[g]), big[ky]);
                        IE_SB();
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) { accN[i >> 2][i & 3] = big[0][i]; accM[i >> 2][i & 3] = big[1][i]; accO[i >> 2][i & 3] = big[2][i]; }
            } else if constexpr (!(IE_XEXP & 64)) {
            mfma_row<TH, (M_ADJ ? IE_K67X_AHB - 1 : IE_K67X_AHB)>(smem + G::OFF_H1 + hs, rr, wv, accN, accM, accO, true, bias7, fill1);
            IE_TACC(5);
            mfma_row<TH, (M_ADJ ? IE_K67X_AHB - 1 : IE_K67X_AHB)>(smem + G::OFF_H6 + hs, rr, wv + 9, accN, accM, accO, false, bias7, fill6);
            } else {
#pragma unroll
                for (int s = 0; s < 72; ++s) gap(s);
            }
            if constexpr (IE_XEXP & 1024) {       // experiment: the curve as a burst behind the streams
#pragma unroll
                for (int it = 0; it < 64; ++it) item(it);
            }
            IE_TACC(1);
            // the running values go on: to the tile-1 waves (exchange row) or into the staged output row
#pragma unroll
            for (int j = 0; j < 4; ++j) *(float*)(yout + yout_off + 64 * j) = v[j];
        };
        for (int t = 0; t < n_steps; t += 4) {      // (N, M, O, F) -> (F, N, M, O)
            step_b(t, a0, a1, a2, a3);
            if (t + 1 < n_steps) step_b(t + 1, a3, a0, a1, a2);
            if (t + 2 < n_steps) step_b(t + 2, a2, a3, a0, a1);
            if (t + 3 < n_steps) step_b(t + 3, a1, a2, a3, a0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef IE_STAMP
    // per wave: [0] wait + barrier, [1] MFMA phase, [2] the rest of the step, n_steps, fine buckets [4..6]: A stores / DMA issue / conversion, B u8 stores / - / curve value reads
    IE_TACC(3);
    if (p.stamps && lane == 0) {
        unsigned long long* d = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        d[0] = st_acc[0]; d[1] = st_acc[1]; d[2] = st_acc[3]; d[3] = n_steps; d[4] = st_acc[4]; d[5] = st_acc[5]; d[6] = st_acc[6]; d[7] = st_acc[2];
    }
#endif
}

#define IE_DEF_K67X(NAME, TA0, TA1, TH, MODE) \
    __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void NAME(K67XParams p) { k67x_body<TA0, TA1, TH, MODE>(p); }
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_f16, _Float16, _Float16, _Float16, 0)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_bf16, __bf16, __bf16, __bf16, 0)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_mix, _Float16, __bf16, _Float16, 0)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_e_f16, _Float16, _Float16, _Float16, 1)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_e_bf16, __bf16, __bf16, __bf16, 1)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_e_mix, _Float16, __bf16, _Float16, 1)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_ea_f16, _Float16, _Float16, _Float16, 2)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_ea_bf16, __bf16, __bf16, __bf16, 2)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_ea_mix, _Float16, __bf16, _Float16, 2)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_u8_f16, _Float16, _Float16, _Float16, 3)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_u8_bf16, __bf16, __bf16, __bf16, 3)
IE_DEF_K67X(zdce_k67x_conv1_conv6_conv7_curve_u8_mix, _Float16, __bf16, _Float16, 3)
#undef IE_DEF_K67X


// ==========================================================================================================================================
// K345X: e_conv3 + e_conv4 + e_conv5 (x2 -> x5; x3 and x4 live in LDS rings only), every wave owning BOTH 16-cout fragments of its layer
// ==========================================================================================================================================
// The strip form at the board's power cap is paid per MFMA AND per fragment read (tools/micro/mfma_pair.hip form 8): a wave that holds both cout fragments of a layer
// reuses a fragment six times instead of three.  Roles (wave w and w + 4 share a SIMD; 72 MFMAs per wave and row everywhere):
//   waves 0, 1  e_conv3 over a HALF of the strip (4 pixel tiles), 18 weight fragments (72 VGPRs), 3 x 8 accumulators; x2 rows by LDS-DMA -> x3 ring
//   waves 2, 3  e_conv4 the same way on the x3 row e_conv3 finished a step earlier -> x4 ring
//   waves 4..7  e_conv5 over a QUARTER of the strip (2 pixel tiles) on [x3, x4] rows of one index, 36 weight fragments (144 VGPRs), 3 x 4 accumulators -> x5 to HBM
// Each layer drops the outermost computed column of the one before (no halo columns are loaded): 128 computed, 122 stored columns per strip.
// Step t: x2 row ys - 3 + t arrives; e_conv3 finishes x3 row ys - 4 + t (ring slot t % 4), e_conv4 x4 row ys - 6 + t (slot t % 3) from the x3 row of step t - 1,
// e_conv5 x5 row ys - 8 + t from the x3 row of step t - 3 and the x4 row of step t - 1.  Rows / columns outside the image are written as zeros (the next layer's padding).
struct K345XParams {
    const void* x2; void* x5;
    const void* w3; const float* b3; const void* w4; const float* b4; const void* w5; const float* b5;
    int H, W, y_begin, y_end, rows_per_block, strips, chunks, n_begin;
};
constexpr int MARG3 = 3, TS3 = TW - 2 * MARG3;      // 122 stored columns per strip
struct K345XGeom {
    static constexpr int PF = 3, NRA = PF + 1, NR3 = 4, NR4 = 3;
    static constexpr int OFF_X2 = 0, OFF_X3 = OFF_X2 + NRA * ROWH_B, OFF_X4 = OFF_X3 + NR3 * ROWH_B, LDS_B = OFF_X4 + NR4 * ROWH_B;      // 11 x 8,320 = 91,520 bytes
};

#ifndef IE_K345X_ASPLIT
#define IE_K345X_ASPLIT 1     // stage A: 1 = split by LAYER with both cout fragments per wave (half the fragment reads, two rows of hand-off writes at the tail; shipped), 0 = (cout fragment,
                              // strip half), both layers per wave, the x3 row handed over between the two streams -- with K345X's stage B that is the slowest of the three (series 19)
#endif
#ifndef IE_K345X_EARLY
#define IE_K345X_EARLY 0
#endif
#ifndef IE_K345X_SWAP
#define IE_K345X_SWAP 0
#endif
#ifndef IE_K345X_EXP
#define IE_K345X_EXP 0      // timing experiments (WRONG results): 1 no hand-off writes of stage A, 2 no x5 stores, 4 no x2 DMA
#endif
// one source row of a 32-channel tensor over HALF a strip for both cout fragments: 12 fragments, 72 MFMAs; n / m / o = the three rotating rows, index = tile
#ifndef IE_K345X_AHA
#define IE_K345X_AHA 1
#endif
#ifndef IE_K345X_AHB
#define IE_K345X_AHB 1
#endif
template <typename T, int AHEAD = IE_K345X_AHA, typename F>
__device__ __forceinline__ void mfma_row_both4(const char* smem_base, const char* slot, const RowRead& rr, const ie_u32x4* w0, const ie_u32x4* w1,
                                               ie_f32x4 (&n0)[4], ie_f32x4 (&m0)[4], ie_f32x4 (&o0)[4], ie_f32x4 (&n1)[4], ie_f32x4 (&m1)[4], ie_f32x4 (&o1)[4],
                                               const ie_f32x4& init0, const ie_f32x4& init1, F&& early) {
    ie_u32x4 bx[12];
    unsigned so = (unsigned)(slot - smem_base);
    asm volatile("" : "+s"(so));
    auto rd = [&](int g) __attribute__((always_inline)) { bx[g] = *(const ie_u32x4*)(smem_base + (unsigned)(rr.base[g >> 2] + 1024 * (g & 3)) + so); };
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) rd(g);
    IE_SB();
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int kx = g >> 2, j = g & 3;
        n0[j] = ie_mfma16_raw<T>(w0[0 * 3 + kx], bx[g], kx == 0 ? init0 : n0[j]);
        if (g + AHEAD < 12) rd(g + AHEAD);
        IE_SB();
        m0[j] = ie_mfma16_raw<T>(w0[1 * 3 + kx], bx[g], m0[j]); IE_SB();
        o0[j] = ie_mfma16_raw<T>(w0[2 * 3 + kx], bx[g], o0[j]); IE_SB();
        n1[j] = ie_mfma16_raw<T>(w1[0 * 3 + kx], bx[g], kx == 0 ? init1 : n1[j]); IE_SB();
        m1[j] = ie_mfma16_raw<T>(w1[1 * 3 + kx], bx[g], m1[j]); IE_SB();
        o1[j] = ie_mfma16_raw<T>(w1[2 * 3 + kx], bx[g], o1[j]); IE_SB();
        if (IE_K345X_EARLY && g == 9) { early(); IE_SB(); }       // experiment: tiles 0, 1 of the oldest row are final behind fragment 9 -- their half of the hand-off under the last 12 MFMAs: slower (7.33 against 7.22 ms)
    }
}

// T2: element type of x2 (and of e_conv3's weights / MFMAs); T3: of x3, x4, x5 (e_conv4 / e_conv5)
template <typename T2, typename T3>
__device__ __forceinline__ void k345x_body(const K345XParams& p) {
    using G = K345XGeom;
    using frag = ie_u32x4;
    constexpr int PF = G::PF, NRA = G::NRA, NR3 = G::NR3, NR4 = G::NR4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ie_fp16_saturate_on();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int stg = (wave >> 2) ^ IE_K345X_SWAP, w4 = wave & 3;      // (IE_K345X_SWAP: e_conv5 on the older waves 0-3)

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = p.n_begin + b;
    const int xc0 = strip * TS3 - MARG3;
    const int H = p.H, W = p.W;
    const int ys = p.y_begin + chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, p.y_end);
    const int n_steps = (ye - ys) + 8;
    const size_t frame_b = (size_t)H * W * PXB;
    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000); };
#ifdef IE_ONLY_STAGE
    if (stg != IE_ONLY_STAGE) return;
#endif

    if (stg == 0) {
        // ============================================== stage A: waves 0, 1 e_conv3; waves 2, 3 e_conv4 ==============================================
        const int hf = w4 & 1;
        // (the role is a template constant of the whole stage: as a run-time select inside the step the two instantiations of the stream kept both sets of temporaries alive and spilled 78 VGPRs)
        auto stage_a = [&](auto role_c) __attribute__((always_inline)) {
        constexpr int role = decltype(role_c)::value;
        const auto rs2 = mk((const char*)p.x2 + (size_t)n * frame_b, frame_b);
        frag wc0[9], wc1[9];
        {
            const frag* wp = (const frag*)(role ? p.w4 : p.w3);
#pragma unroll
            for (int f = 0; f < 9; ++f) { wc0[f] = wp[f * 64 + lane]; wc1[f] = wp[(9 + f) * 64 + lane]; }
        }
        ie_f32x4 bias0, bias1;
        {
            const float* bp = role ? p.b4 : p.b3;
#pragma unroll
            for (int i = 0; i < 4; ++i) { bias0[i] = bp[4 * kq + i]; bias1[i] = bp[16 + 4 * kq + i]; }
        }
        const RowRead rr = row_read(hf, p16, kq);
        // x2 rows: piece m = 2 w4 + j (16 pixels x 64 bytes) lands at ring pixels 16 m + 1 ..; the source chunk is the position's un-swizzled one
        unsigned body_xoff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = (2 * w4 + j) * 16 + (lane >> 2);
            const int sc = (lane & 3) ^ (((q + 1) >> 1) & 3);
            const int x = xc0 + q;
            body_xoff[j] = (x >= 0 && x < W) ? (unsigned)(x * PXB + sc * 16) : OOB;
        }
        auto issue = [&](int t) __attribute__((always_inline)) {
            const int y = ys - 3 + t;
            const unsigned rmask = (y >= 0 && y < H) ? 0u : OOB;
            const unsigned row_b = rmask ? 0u : (unsigned)y * (unsigned)W * PXB;
            char* s2 = smem + G::OFF_X2 + (t % NRA) * ROWH_B + PXB;
#pragma unroll
            for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(s2 + (2 * w4 + j) * 1024), 16, (int)((row_b + body_xoff[j]) | rmask), 0, 0, 0);
        };
        const int em_off0 = put_off(0, hf, p16, kq, 1), em_off1 = put_off(1, hf, p16, kq, 1);
        unsigned em_in = 0;
        {
            const int q = 16 * (4 * hf + (kq & 1)) + p16;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int x = xc0 + q + 32 * jj;
                em_in |= (x >= 0 && x < W) ? (1u << jj) : 0u;
            }
        }
        const bool strip_inside = xc0 >= 0 && xc0 + TW <= W;
        ie_f32x4 a0[4], a1[4], a2[4], c0[4], c1[4], c2[4];      // a*: cout fragment 0, c*: cout fragment 1
#pragma unroll
        for (int j = 0; j < 4; ++j) { a0[j] = bias0; a1[j] = bias0; a2[j] = bias0; c0[j] = bias1; c1[j] = bias1; c2[j] = bias1; }
#pragma unroll
        for (int t = 0; t < PF; ++t) issue(t);
        auto step_a = [&](int t, ie_f32x4 (&n0)[4], ie_f32x4 (&m0)[4], ie_f32x4 (&o0)[4], ie_f32x4 (&n1)[4], ie_f32x4 (&m1)[4], ie_f32x4 (&o1)[4]) __attribute__((always_inline)) {
            ie_wait_vm_imm<(PF - 1) * 2>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int m = role == 0 ? ys - 4 + t : ys - 6 + t;       // the row this step finishes
            char* dst = role == 0 ? smem + G::OFF_X3 + (t % NR3) * ROWH_B : smem + G::OFF_X4 + (t % NR4) * ROWH_B;
            const bool plain = strip_inside && m >= 0 && m < H;
            const unsigned keep = (m >= 0 && m < H) ? em_in : 0u;
            auto hand = [&](int jj) __attribute__((always_inline)) {
                if (plain) { put_pair<T3, false>(o0, jj, dst, em_off0, 3u); put_pair<T3, false>(o1, jj, dst, em_off1, 3u); }
                else { put_pair<T3, true>(o0, jj, dst, em_off0, keep); put_pair<T3, true>(o1, jj, dst, em_off1, keep); }
            };
            auto early = [&]() __attribute__((always_inline)) { hand(0); };
            if constexpr (role == 0) mfma_row_both4<T2>(smem, smem + G::OFF_X2 + (t % NRA) * ROWH_B, rr, wc0, wc1, n0, m0, o0, n1, m1, o1, bias0, bias1, early);
            else mfma_row_both4<T3>(smem, smem + G::OFF_X3 + ((t + NR3 - 1) % NR3) * ROWH_B, rr, wc0, wc1, n0, m0, o0, n1, m1, o1, bias0, bias1, early);
            if (!(IE_K345X_EXP & 1)) {
            if (!IE_K345X_EARLY) hand(0);
            hand(1);
            }
            if (!(IE_K345X_EXP & 4)) issue(t + PF);       // (behind the hand-off writes: they are what the barrier waits for)
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_a(t, a0, a1, a2, c0, c1, c2);
            if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1, c2, c0, c1);
            if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0, c1, c2, c0);
        }
        };
        if constexpr (IE_K345X_ASPLIT) {
            if ((w4 >> 1) == 0) stage_a(std::integral_constant<int, 0>{}); else stage_a(std::integral_constant<int, 1>{});
        } else {
            // ---- stage A in the (cout fragment, strip half) mapping: every wave runs e_conv3 AND e_conv4 for its fragment (36 + 36 MFMAs, 12 + 12 fragment reads); the x3 row is
            // handed over BETWEEN the two streams, under stage B's MFMAs, and only the x4 row at the tail (K345X's layer split ended every stage-A step with two rows of writes) ----
            const int cf = w4 & 1, hfa = w4 >> 1;
            const auto rs2 = mk((const char*)p.x2 + (size_t)n * frame_b, frame_b);
            frag wv[18];
#pragma unroll
            for (int f = 0; f < 9; ++f) { wv[f] = ((const frag*)p.w3)[(cf * 9 + f) * 64 + lane]; wv[9 + f] = ((const frag*)p.w4)[(cf * 9 + f) * 64 + lane]; }
            ie_f32x4 bias3, bias4;
#pragma unroll
            for (int i = 0; i < 4; ++i) { bias3[i] = p.b3[16 * cf + 4 * kq + i]; bias4[i] = p.b4[16 * cf + 4 * kq + i]; }
            const RowRead rr = row_read(hfa, p16, kq);
            unsigned body_xoff[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = (2 * w4 + j) * 16 + (lane >> 2);
                const int sc = (lane & 3) ^ (((q + 1) >> 1) & 3);
                const int x = xc0 + q;
                body_xoff[j] = (x >= 0 && x < W) ? (unsigned)(x * PXB + sc * 16) : OOB;
            }
            auto issue = [&](int t) __attribute__((always_inline)) {
                const int y = ys - 3 + t;
                const unsigned rmask = (y >= 0 && y < H) ? 0u : OOB;
                const unsigned row_b = rmask ? 0u : (unsigned)y * (unsigned)W * PXB;
                char* s2 = smem + G::OFF_X2 + (t % NRA) * ROWH_B + PXB;
#pragma unroll
                for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(s2 + (2 * w4 + j) * 1024), 16, (int)((row_b + body_xoff[j]) | rmask), 0, 0, 0);
            };
            const int em_off = put_off(cf, hfa, p16, kq, 1);
            unsigned em_in = 0;
            {
                const int q = 16 * (4 * hfa + (kq & 1)) + p16;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int x = xc0 + q + 32 * jj;
                    em_in |= (x >= 0 && x < W) ? (1u << jj) : 0u;
                }
            }
            const bool strip_inside = xc0 >= 0 && xc0 + TW <= W;
            ie_f32x4 a0[4], a1[4], a2[4], c0[4], c1[4], c2[4];      // a*: e_conv3, c*: e_conv4
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0[j] = bias3; a1[j] = bias3; a2[j] = bias3; c0[j] = bias4; c1[j] = bias4; c2[j] = bias4; }
#pragma unroll
            for (int t = 0; t < PF; ++t) issue(t);
            auto nofill = [&](int) __attribute__((always_inline)) {};
            auto put1 = [&](const ie_f32x4 (&acc)[4], char* dst, int m) __attribute__((always_inline)) {
                if (strip_inside && m >= 0 && m < H) put_row<T3, false>(acc, dst, em_off, 3u);
                else put_row<T3, true>(acc, dst, em_off, (m >= 0 && m < H) ? em_in : 0u);
            };
            auto step_a = [&](int t, ie_f32x4 (&n3)[4], ie_f32x4 (&m3)[4], ie_f32x4 (&o3)[4], ie_f32x4 (&n4)[4], ie_f32x4 (&m4)[4], ie_f32x4 (&o4)[4]) __attribute__((always_inline)) {
                ie_wait_vm_imm<(PF - 1) * 2>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                mfma_row<T2, 3>(smem + G::OFF_X2 + (t % NRA) * ROWH_B, rr, wv, n3, m3, o3, true, bias3, nofill);
                put1(o3, smem + G::OFF_X3 + (t % NR3) * ROWH_B, ys - 4 + t);
                mfma_row<T3, 3>(smem + G::OFF_X3 + ((t + NR3 - 1) % NR3) * ROWH_B, rr, wv + 9, n4, m4, o4, true, bias4, nofill);
                put1(o4, smem + G::OFF_X4 + (t % NR4) * ROWH_B, ys - 6 + t);
                issue(t + PF);
            };
            for (int t = 0; t < n_steps; t += 3) {
                step_a(t, a0, a1, a2, c0, c1, c2);
                if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1, c2, c0, c1);
                if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0, c1, c2, c0);
            }
        }
    } else {
        // ===================================================== stage B: e_conv5, a quarter of the strip per wave =====================================================
        const auto rs5 = mk((char*)p.x5 + (size_t)n * frame_b, frame_b);
        frag wv0[18], wv1[18];
#pragma unroll
        for (int f = 0; f < 18; ++f) { wv0[f] = ((const frag*)p.w5)[f * 64 + lane]; wv1[f] = ((const frag*)p.w5)[(18 + f) * 64 + lane]; }
        ie_f32x4 bias0, bias1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { bias0[i] = p.b5[4 * kq + i]; bias1[i] = p.b5[16 + 4 * kq + i]; }
        RowRead rq;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = 32 * w4 + p16 + kx;
            rq.base[kx] = P * PXB + ((kq ^ ((P >> 1) & 3)) << 4);
        }
        // after the lane-pair exchange this lane holds, per cout fragment cf, the 16-byte chunk 2 cf + (kq >> 1) of pixel 32 w4 + 16 (kq & 1) + p16
        unsigned st_off[2];
        {
            const int q = 32 * w4 + 16 * (kq & 1) + p16, x = xc0 + q;
            const bool ok = q >= MARG3 && q < TW - MARG3 && x < W;
#pragma unroll
            for (int cf = 0; cf < 2; ++cf) st_off[cf] = ok ? (unsigned)(x * PXB + (2 * cf + (kq >> 1)) * 16) : OOB;
        }
        ie_f32x4 a0[4], a1[4], a2[4];       // index 2 * cout fragment + tile
#pragma unroll
        for (int j = 0; j < 4; ++j) { const ie_f32x4 bb = j < 2 ? bias0 : bias1; a0[j] = bb; a1[j] = bb; a2[j] = bb; }
        // the finished row leaves at the HEAD of the next step (the younger stage does what is not an MFMA while the older one streams, finding 14): accN of step t is
        // the set that was accO at step t - 1 -- its row ys - 9 + t is packed and stored before the first MFMA restarts it
        auto store_row = [&](int yo, const ie_f32x4 (&acc)[4]) __attribute__((always_inline)) {
            const unsigned rmask = (yo >= ys && yo < ye) ? 0u : OOB;
            const unsigned row_b = rmask ? 0u : (unsigned)yo * (unsigned)W * PXB;
#pragma unroll
            for (int cf = 0; cf < 2; ++cf) {
                const ie_u32x2 a = {ie_pack2_relu<T3>(acc[2 * cf][0], acc[2 * cf][1]), ie_pack2_relu<T3>(acc[2 * cf][2], acc[2 * cf][3])};
                const ie_u32x2 b2 = {ie_pack2_relu<T3>(acc[2 * cf + 1][0], acc[2 * cf + 1][1]), ie_pack2_relu<T3>(acc[2 * cf + 1][2], acc[2 * cf + 1][3])};
                const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
                __builtin_amdgcn_raw_buffer_store_b128(ie_u32x4{sx[0], sy[0], sx[1], sy[1]}, rs5, (int)((row_b + st_off[cf]) | rmask), 0, IE_X_ST_AUX);
            }
        };
        auto step_b = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!(IE_K345X_EXP & 2)) store_row(ys - 9 + t, accN);
            IE_SB();
            mfma_row_both<T3, IE_K345X_AHB, false>(smem, smem + G::OFF_X3 + ((t + NR3 - 3) % NR3) * ROWH_B, rq, wv0, wv1, accN, accM, accO, true, bias0, bias1);
            mfma_row_both<T3, IE_K345X_AHB, false>(smem, smem + G::OFF_X4 + ((t + NR4 - 1) % NR4) * ROWH_B, rq, wv0 + 9, wv1 + 9, accN, accM, accO, false, bias0, bias1);
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int t = 0;
        for (; t < n_steps; t += 3) {
            step_b(t, a0, a1, a2);
            if (t + 1 < n_steps) step_b(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_b(t + 2, a1, a2, a0);
        }
        // the last row (finished at step n_steps - 1 = accO of that step): which set it is depends on n_steps % 3
        const int last = n_steps - 1;
        if (last % 3 == 0) store_row(ys - 8 + last, a2); else if (last % 3 == 1) store_row(ys - 8 + last, a1); else store_row(ys - 8 + last, a0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#define IE_DEF_K345X(NAME, T2, T3) \
    __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void NAME(K345XParams p) { k345x_body<T2, T3>(p); }
IE_DEF_K345X(zdce_k345x_conv3_conv4_conv5_f16, _Float16, _Float16)
IE_DEF_K345X(zdce_k345x_conv3_conv4_conv5_bf16, __bf16, __bf16)
IE_DEF_K345X(zdce_k345x_conv3_conv4_conv5_mix, _Float16, __bf16)

}  // namespace

int ie_zdce_launch_x(const ZdceXLaunch& a, hipStream_t st) {
    if ((size_t)a.H * a.W * 32 * 2 >= 0x7fffffffull) IE_FAIL(IE_ERR_UNSUPPORTED, "zdce x: frame too large for 31-bit offsets");
    const int strips = ie_cdiv(a.W, TS);
    if (a.kind == 12) {
        if (a.dt_h != IE_F16 && a.dt_h != IE_BF16) IE_FAIL(IE_ERR_UNSUPPORTED, "zdce k12x: dtype %d", a.dt_h);
        K12XParams p;
        memset(&p, 0, sizeof(p));
        p.img = a.img; p.w1 = a.w1; p.b1 = a.b1; p.w2 = a.w2; p.b2 = a.b2; p.dst = a.x2;
        p.H = a.H; p.W = a.W; p.y_begin = a.y_begin; p.y_end = a.y_end; p.n_begin = a.n_begin; p.strips = strips;
        // 50 KiB of LDS and 128 VGPRs: two workgroups per CU = 512 slots; as few row chunks as fill them (every chunk re-runs 8 apron rows)
        const int rows = a.y_end - a.y_begin;
        long long chunks = 512 / ((long long)strips * a.n_count);
        if (chunks < 1) chunks = 1;
        if (chunks > (rows + 7) / 8) chunks = (rows + 7) / 8;
        p.rows_per_block = ie_cdiv(rows, (int)chunks);
        p.chunks = ie_cdiv(rows, p.rows_per_block);
        const long long blocks = (long long)strips * p.chunks * a.n_count;
        if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "zdce k12x: bad grid %lld", blocks);
        typedef void (*Fn)(K12XParams);
        const Fn fn = a.dt_h == IE_F16 ? zdce_k12x_conv1_conv2_f16 : zdce_k12x_conv1_conv2_bf16;
        const int lds = a.dt_h == IE_F16 ? K12XGeom<_Float16>::LDS_B : K12XGeom<__bf16>::LDS_B;
        void* args[] = {(void*)&p};
        IE_HIP_CHECK(hipLaunchKernel((const void*)fn, dim3((unsigned)blocks), dim3(512), args, lds, st));
        return IE_OK;
    }
    if (a.kind == 345) {
        K345XParams p;
        memset(&p, 0, sizeof(p));
        p.x2 = a.x2; p.x5 = (void*)a.x5; p.w3 = a.w3; p.b3 = a.b3; p.w4 = a.w4; p.b4 = a.b4; p.w5 = a.w5; p.b5 = a.b5;
        p.H = a.H; p.W = a.W; p.y_begin = a.y_begin; p.y_end = a.y_end; p.n_begin = a.n_begin; p.strips = ie_cdiv(a.W, TS3);
        p.rows_per_block = a.rows_per_block;
        p.chunks = ie_cdiv(a.y_end - a.y_begin, a.rows_per_block);
        const long long blocks = (long long)p.strips * p.chunks * a.n_count;
        if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "zdce k345x: bad grid %lld", blocks);
        typedef void (*Fn)(K345XParams);
        const Fn fn = (a.dt_a0 == IE_F16 && a.dt_3 == IE_F16) ? zdce_k345x_conv3_conv4_conv5_f16
                      : (a.dt_a0 == IE_BF16 && a.dt_3 == IE_BF16) ? zdce_k345x_conv3_conv4_conv5_bf16
                      : (a.dt_a0 == IE_F16 && a.dt_3 == IE_BF16) ? zdce_k345x_conv3_conv4_conv5_mix : (Fn) nullptr;
        if (!fn) IE_FAIL(IE_ERR_UNSUPPORTED, "zdce k345x: dtypes x2=%d x3=%d", a.dt_a0, a.dt_3);
        IE_GRANT_LDS(fn, K345XGeom::LDS_B);
        void* args[] = {(void*)&p};
        IE_HIP_CHECK(hipLaunchKernel((const void*)fn, dim3((unsigned)blocks), dim3(512), args, K345XGeom::LDS_B, st));
        return IE_OK;
    }
    if (a.kind != 67) IE_FAIL(IE_ERR_INVALID, "zdce x: kind %d", a.kind);
    if (a.adjust && (size_t)a.H * a.W * 24 * 4 >= 0x7fffffffull) IE_FAIL(IE_ERR_UNSUPPORTED, "zdce k67x: frame too large for 31-bit offsets into `adjust`");
    K67XParams p;
    memset(&p, 0, sizeof(p));
    p.img = a.img; p.x2 = a.x2; p.x5 = a.x5; p.w1 = a.w1; p.b1 = a.b1; p.w6 = a.w6; p.b6 = a.b6; p.w7 = a.w7; p.b7 = a.b7; p.w1q = a.w1q;
    p.enhanced = a.enhanced; p.adjust = a.adjust; p.u8 = a.u8;
    p.H = a.H; p.W = a.W; p.y_begin = a.y_begin; p.y_end = a.y_end; p.n_begin = a.n_begin; p.strips = strips;
    p.rows_per_block = a.rows_per_block;
    p.stamps = (unsigned long long*)a.stamps;
    p.chunks = ie_cdiv(a.y_end - a.y_begin, a.rows_per_block);
    const long long blocks = (long long)strips * p.chunks * a.n_count;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "zdce k67x: bad grid %lld", blocks);
    typedef void (*Fn)(K67XParams);
    const bool al = (a.W & 3) == 0 && ((size_t)a.enhanced & 15) == 0;
    const int mode = (a.enhanced && !a.adjust && !a.u8 && al) ? 1 : (a.enhanced && a.adjust && !a.u8 && al) ? 2 : (a.u8 && !a.enhanced && !a.adjust) ? 3 : 0;
    const bool h_all = a.dt_h == IE_F16 && a.dt_a0 == IE_F16 && a.dt_a1 == IE_F16;
    const bool b_all = a.dt_h == IE_BF16 && a.dt_a0 == IE_BF16 && a.dt_a1 == IE_BF16;
    const bool mix = a.dt_h == IE_F16 && a.dt_a0 == IE_F16 && a.dt_a1 == IE_BF16;
    static const Fn fns[3][4] = {
        {zdce_k67x_conv1_conv6_conv7_curve_f16, zdce_k67x_conv1_conv6_conv7_curve_e_f16, zdce_k67x_conv1_conv6_conv7_curve_ea_f16, zdce_k67x_conv1_conv6_conv7_curve_u8_f16},
        {zdce_k67x_conv1_conv6_conv7_curve_bf16, zdce_k67x_conv1_conv6_conv7_curve_e_bf16, zdce_k67x_conv1_conv6_conv7_curve_ea_bf16, zdce_k67x_conv1_conv6_conv7_curve_u8_bf16},
        {zdce_k67x_conv1_conv6_conv7_curve_mix, zdce_k67x_conv1_conv6_conv7_curve_e_mix, zdce_k67x_conv1_conv6_conv7_curve_ea_mix, zdce_k67x_conv1_conv6_conv7_curve_u8_mix}};
    const Fn fn = h_all ? fns[0][mode] : (b_all ? fns[1][mode] : (mix ? fns[2][mode] : nullptr));
    if (!fn) IE_FAIL(IE_ERR_UNSUPPORTED, "zdce k67x: dtypes x2=%d x5=%d h=%d", a.dt_a0, a.dt_a1, a.dt_h);
    const int lds = a.dt_h == IE_F16 ? K67XGeom<_Float16>::LDS_B : K67XGeom<__bf16>::LDS_B;
    IE_GRANT_LDS(fn, lds);
    void* args[] = {(void*)&p};
    IE_HIP_CHECK(hipLaunchKernel((const void*)fn, dim3((unsigned)blocks), dim3(512), args, lds, st));
    return IE_OK;
}
