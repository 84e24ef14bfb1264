// Two-stage fused strip kernels for Zero-DCE (gfx950): the intermediate of the pair never goes
// to HBM (or goes there once, write-only, when a later layer needs it).
//
//   K12: e_conv1 (image -> x1, LDS only) + e_conv2 (x1 -> x2) + the x1 half of e_conv7 (x1 -> P)   12+64+64 = 140 B/px
//        (e_conv7([x1,x6]) = W7a*x1 + W7b*x6: P = W7a*x1 + bias7 is computed where x1 lives and replaces x1 in HBM,
//         which halves stage B of K67 -- the busiest kernel).  Stage A = e_conv1 + e_conv7a (44 MFMAs per wave and row),
//         stage B = e_conv2 (36): balanced like K345, each wave's non-MFMA work hides under its SIMD partner's MFMAs.
//   K45: e_conv4 (x3 -> x4, LDS only) + e_conv5 ([x3,x4] -> x5)       64+64    = 128 B/px (was 320)
//   K345: e_conv3 + e_conv4 on stage A (x2 -> x3 -> x4, both LDS only; 72 MFMAs per wave and row) + e_conv5 on stage B
//        (72 MFMAs): x3 never reaches HBM and the two stages are balanced.  64+64 = 128 B/px for three layers (conv3 + K45: 256).
//        Two hand-offs cost two halo columns a side: 124 of the 128 computed columns are stored.
//   K67: e_conv6 ([x2,x5] -> x6, LDS only) + e_conv7 (P + W7b*x6) + curve  192+12+12 = 216 B/px (was 344)
//
// Reference semantics: src/mon/vision/enhance/llie/zero_dce/zero_dce.py:173-191.
//
// One workgroup = 8 waves.  Waves 0-3 run stage A, waves 4-7 stage B, both on the same 128 computed
// columns of a strip and in lock step down the rows (one s_barrier per row): at step t stage A
// finishes row r of its output and writes it (ReLU, 16-bit, zero outside the image = the next
// layer's zero padding) into a 2-slot LDS ring in exactly the swizzled [pixel][32ch] image the
// consumer's ds_read_b128 fragments expect; stage B consumes that row at step t+1.  A-waves and
// B-waves share each SIMD, so A's VALU/LDS work (im2col for e_conv1, ring writes) overlaps B's MFMAs.
// Stage B needs a 1-column halo of A's output, so of the 128 computed columns the centre 126 are
// stored and strips advance by 126 columns.
//
// Per-stage machinery is the one of zdce_conv3x3_mfma (ie_zdce_kernels.hip): weights as the MFMA A
// operand in registers, HBM rows DMA'd once into LDS rings (buffer_load ... lds) behind counted
// vmcnt waits, every pixel fragment feeding three MFMAs (ky = 0,1,2) into rotating accumulators.
#include "ie_zdce_internal.h"
#include <string.h>
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int TW = 128;          // computed columns per workgroup
constexpr int TS = 126;          // stored columns per workgroup (strip pitch)
constexpr int ROW_PX = TW + 2;   // pixels per ring row (body + 2 halo)
constexpr unsigned OOB = 0x80000000u;
#ifndef IE_PF
#define IE_PF 3
#endif
constexpr int PF_DEFAULT = IE_PF; // rows of DMA issued ahead (K67 is LDS-limited to 3)

enum { K12 = 12, K45 = 45, K67 = 67, K345 = 345, K67N = 670, K67F = 671, K67U = 672, K67A = 673 };   // K67A: enhanced + the 24 curve maps, fp32, W % 4 == 0 (what model.forward() returns)
//   // K67U: K67N compiled for the uint8 HWC output alone (the u8 pipeline, ie_zdce_forward_u8)
//   // K67N: K67 with the curve in stage B's registers (below); K67F: K67N
                                                                              // compiled for `enhanced` fp32 as the only output and W % 4 == 0 (no output-mode branches)

#ifndef IE_EXP
#define IE_EXP 0      // timing experiments (wrong results): 1 no image DMA / row conversion, 2 e_conv1 without LDS reads, 4 no hand-off write,
#endif                // 8 MFMA streams without LDS reads, 16 no K12 stores, 32 no e_conv1 at all, 64 raw-accumulator stores (no epilogue VALU), 128 K12 stores forced out of range; K67: 1024 no curve, 2048 curve without tanh, 4096 no image DMA, 8192 no P DMA
#ifndef IE_HOTWAIT
#define IE_HOTWAIT 7      // bit 0: K12, bit 1: K345, bit 2: the others -- steady-state vmcnt waits tested first (ie_wait_vm_hot), A-B switch
#endif
#ifndef IE_K67N_STREAM_A
#define IE_K67N_STREAM_A 0   // 1: K67N stage A as ONE pair of MFMA streams per step: the hand-off write of the row finished a step earlier and the
#endif                       // step's four LDS-DMA pieces ride as fillers behind fixed MFMAs (0: MFMAs, then DMA issue, then the hand-off write)
#ifndef IE_K67N_PA
#define IE_K67N_PA 0      // 1: the P row's eight LDS-DMA pieces are issued by stage A (two per wave) instead of stage B (measured slower: 6.86 against 6.65 ms)
#endif
#if IE_K67N_STREAM_A && IE_K67N_PA
#error "IE_K67N_STREAM_A issues only stage A's own rows: build it with -DIE_K67N_PA=0"
#endif
#ifndef IE_K67N_PRIO
#define IE_K67N_PRIO 0
#endif
#ifndef IE_K12_RPB
#define IE_K12_RPB 1
#endif
#ifndef IE_K12_STREAM
#define IE_K12_STREAM 1
#endif
#ifndef IE_K67_CURVE2
#define IE_K67_CURVE2 0   // 1: K67's curve on three stage-B waves, two columns per thread in packed fp32 arithmetic, the fourth wave issuing all of
                          // stage B's DMA.  Half the vector instructions, but measured 9.6 against 7.6 ms: three waves then carry a twice as long
                          // serial chain (transcendental latency) while five wait at the barrier -- the curve wants MORE lanes, not fewer.
#endif
#ifndef IE_EMIT_B128
#define IE_EMIT_B128 1    // hand-off rows leave the accumulators as 16-byte pieces (lanes l / l+16 exchange halves, ds_write_b128: 8 pixels x
#endif                    // 16 bytes per lane group = every bank once) instead of 8-byte pieces (ds_write_b64: pixels q and q+8 share a bank pair)
#ifndef IE_K345_OUT
#define IE_K345_OUT 0     // 1: K345 parks its x5 rows in LDS and stores whole lines like K12 (measured slower: 7.42 vs 7.0 ms -- the kernel is
#endif                    // MFMA-dense and its writes are half of K12's, so the extra LDS traffic costs more than the write path gives back)
// Ring geometry shared by the kernel body and the launcher (LDS size).
template <int KIND> struct FusedGeom {
    // rows per barrier.  K12's per-row fixed chain (barrier, e_conv1's LDS-read -> MFMA -> MFMA chain, row conversion, hand-off write)
    // is as long as its MFMA time, and its rings are small: it runs TWO rows between barriers (the stages then talk through rows
    // that are one barrier period old, so every hand-off lags RPB rows and the rings hold 2 RPB slots).
    static constexpr int RPB = (KIND == K12) ? IE_K12_RPB : 1;
    static constexpr bool IS67N = (KIND == K67N || KIND == K67F || KIND == K67U || KIND == K67A);
    static constexpr bool IS67 = (KIND == K67 || IS67N);
    static constexpr int PF = IS67 ? 3 : (KIND == K12 && RPB > 1 ? 5 : PF_DEFAULT);   // rows of DMA issued ahead
    static constexpr int ROW_B = ROW_PX * 64;
    static constexpr int NSA = IS67 ? 2 : ((KIND == K45 || KIND == K345) ? 1 : 0);     // stage A HBM sources (K12: image)
    static constexpr int NRA = (KIND == K45) ? PF + 3 : PF + 1;   // slots of the A rings (K45: B re-reads x3 two rows later)
    static constexpr int NRB = PF + 1;                            // K67: P ring of stage B
    static constexpr int NR3 = 4;                                 // K345: x3 ring (written at step t, read by e_conv4 at t+1 and e_conv5 at t+3)
    static constexpr int NRH = 2 * RPB;                           // A -> B hand-off ring
    static constexpr int LAG_A = (KIND == K12) ? RPB : 0;         // K12: stage A runs behind the image row conversion
    static constexpr int LAG_B = RPB - 1;                         // extra rows stage B runs behind stage A
    // K12 streamed form: every epilogue (hand-off write, stores) is deferred by one step and rides between the MFMAs of the next
    static constexpr int DEF = (KIND == K12 && IE_K12_STREAM) ? 1 : 0;
    static constexpr int DEFA = (IS67N && IE_K67N_STREAM_A) ? 1 : 0;   // K67N: stage A's hand-off write is deferred by one step
    static constexpr int LAG_H = LAG_B + DEF + DEFA;              // rows between stage A computing a hand-off row and its consumers reading it, minus one
    // fp32 image ring slots (K67: kept until the deferred curve; K67N: DMA'd PF steps ahead of the step whose curve reads it).
    // (Tried for K67N: the fp32 pixel riding in the unused bytes 48..59 of K12's P record instead of a ring of its own -- one 64-byte line per
    // pixel, no image read in the last kernel.  K12 + 0.45 ms per 64 frames for the six LDS reads and eight selects per wave and row in its
    // stage A stream against 0 measurable for these DMAs here: not kept.)
    static constexpr int NRI = (KIND == K12) ? PF + RPB : (IS67N ? PF + 1 : PF + 3);
    static constexpr int NRC = 2 + LAG_A + RPB;                   // K12: converted image rows (3 being read + RPB being written)
    static constexpr int IMG_PLANE67N = 144 * 4;                   // K67N: plane pitch 144 floats -> colours 0 / 1 of a pixel on different banks
    static constexpr int IMG_SLOT_B = IS67N ? 3 * IMG_PLANE67N : 2048;   // 3 planes x 132 px fp32 (K12) / 3 x 128 (K67)
    static constexpr int CV_PLANE_B = 1088;                       // K12: converted image row, [136 px][RGB0] 16-bit
    static constexpr int CV_SLOT_B = 2 * CV_PLANE_B;              //      hi plane + lo plane
    static constexpr int STAGE_SLOT_B = 24 * TW * 4;              // K67: tanh(r) staging [24][128] fp32, double buffered
    static constexpr int A_RING_B = NSA * NRA * ROW_B;
    static constexpr int B_RING_B = IS67 ? NRB * ROW_B : 0;
    static constexpr int R3_B = (KIND == K345) ? NR3 * ROW_B : 0;
    static constexpr int RA_B = NRH * ROW_B;
    static constexpr int IMG_B = (KIND == K45 || KIND == K345) ? 0 : NRI * IMG_SLOT_B;
    static constexpr int CV_B = (KIND == K12) ? NRC * CV_SLOT_B : 0;
    static constexpr int STAGE_B = (KIND == K67) ? 2 * STAGE_SLOT_B : 0;
    static constexpr int DUMP_B = (KIND == K12) ? 1024 : 0;       // K12: where the idle lanes of the branch-free row conversion write
    // K12 streamed form: finished x2 / P rows are parked in LDS as whole pixel lines ([px][32 ch], the hand-off image) and stored one
    // step later, 1 KiB of whole 64-byte lines per store instruction.  Stored straight from the accumulators a wave covers only 32
    // bytes of each pixel (its 16 output channels), and half-line writes cost the write path twice (profiles/r02/k12_experiments_2.log).
    static constexpr int OUT = (DEF || (KIND == K345 && IE_K345_OUT)) ? 1 : 0;
    static constexpr int OUT_B = OUT ? (KIND == K12 ? 2 : 1) * 2 * ROW_B : 0;     // (two tensors x) two slots
    // K67N: the curve's fp32 results of a row, [3 colours][132] (pixel q at float q + 2, so the 124 stored pixels start 16-byte aligned), two slots
    static constexpr int OUTF_PLANE_B = 132 * 4, OUTF_SLOT_B = 4 * OUTF_PLANE_B;      // (a fourth plane: where the idle lane group kq = 3 writes, branch-free)
    static constexpr int OUTF_B = IS67N ? 2 * OUTF_SLOT_B : 0;
    static constexpr int LDS_B = A_RING_B + B_RING_B + R3_B + RA_B + IMG_B + CV_B + STAGE_B + DUMP_B + OUT_B + OUTF_B;
};

struct FusedParams {
    const void* a_src0;  // stage A HBM source 0 (NHWC T)            K45: x3, K67: x2
    const void* a_src1;  // stage A HBM source 1                     K67: x5
    const void* b_src;   // stage B HBM source with its own ring     K67: P (accumulator seed of e_conv7)
    const float* img;    // NCHW fp32 image                          K12 (A input), K67 (curve)
    const void* a_w; const float* a_bias;
    const void* a2_w; const float* a2_bias;   // K345: e_conv4 (stage A's second layer)
    const void* b_w; const float* b_bias;
    const float* b2_bias; // K12: e_conv7's bias (folded into P)
    void* a_dst;         // K12: P = W7a*x1 + bias7 (NHWC T, 32 ch, 24 used)
    void* b_dst;         // K12: x2, K45: x5
    float* enhanced; float* adjust; unsigned char* u8;   // K67
    int N, H, W;
    int y_begin, y_end, rows_per_block, strips, chunks, n_begin;
    unsigned long long* stamps;   // diagnostic builds (-DIE_STAMP) only: [block][wave][8] cycle sums
};

__device__ __forceinline__ void wait_vm(int n) {
    // s_waitcnt takes an immediate; n is wave-uniform.  n < 0: this wave has nothing to wait for.
#define IE_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
    IE_W(1) IE_W(2) IE_W(3) IE_W(4) IE_W(5) IE_W(6) IE_W(7) IE_W(8) IE_W(9) IE_W(10) IE_W(11) IE_W(12) IE_W(13) IE_W(14)
    IE_W(15) IE_W(16) IE_W(17) IE_W(18) IE_W(19) IE_W(20) IE_W(21) IE_W(22) IE_W(23) IE_W(24) IE_W(25) IE_W(26) IE_W(27) IE_W(28) IE_W(29) IE_W(30)
    default: if (n >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef IE_W
}

}  // namespace

#ifdef IE_STAMP
#define IE_T(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IE_T(var) do { } while (0)
#endif

template <typename T> struct ie_tag { using type = T; };

// Element types (all 16-bit): TA0 / TA1 = stage A's HBM sources and their weights (K45: x3; K67: x2, x5);
// TH = the LDS hand-off (x1 / x4 / x6), stage B's weights on it and K67's P seed (K12: also e_conv1's operands);
// TO = stage B's HBM outputs (K12: x2 and P, K45: x5).  They differ only in the "bf16" schedule of the plan,
// which keeps the precision-critical tensors x1, x2, x6 and P in fp16.
template <typename TA0, typename TA1, typename TH, typename TO, int KIND, typename T3 = TH>
__device__ __forceinline__ void zdce_fused2_body(const FusedParams& p) {
    using frag = ie_u32x4;
    using G = FusedGeom<KIND>;
    constexpr int PF = G::PF, RPB = G::RPB;
    constexpr int PXB = 32 * 2;                             // 64 B per pixel per tensor
    constexpr int ROW_B = G::ROW_B;                         // one ring row of one tensor
    constexpr int NSA = G::NSA;
    constexpr bool IS67 = G::IS67, IS67N = G::IS67N, FAST67 = (KIND == K67F || KIND == K67A), U8ONLY = (KIND == K67U), LEAN67 = (KIND == K67F) || U8ONLY, NOU8 = LEAN67 || KIND == K67A;
    (void)FAST67; (void)U8ONLY; (void)LEAN67; (void)NOU8;
    constexpr int NFB = (KIND == K45 || KIND == K12 || KIND == K345 || IS67N) ? 18 : 9;         // stage B weight fragments (K12: e_conv2 + e_conv7a; K67N: both 16-row tiles of e_conv7b)
    constexpr int NW = IS67 ? 18 : NFB;                     // weight fragments held per wave (K12 stage A: 2, K67 / K345 stage A: 18)
    // K67 and K345 load NO halo columns for stage A's HBM rows (one 8-lane LDS-DMA instruction per tensor and row on the
    // stage's busiest wave): the outermost computed column of stage A is then wrong too and one more column a side is dropped.
    // At 1920 columns the strip count stays 16 (124- and 122-column pitches).
    constexpr bool HALO = (KIND == K12 || KIND == K45);
    constexpr int MARG = ((KIND == K345) ? 2 : 1) + (HALO ? 0 : 1);  // computed columns lost per side (one per LDS hand-off, one without halo)
    constexpr int TSK = TW - 2 * MARG;                               // stored columns = strip pitch
    // ---- LDS carve (FusedGeom) ------------------------------------------------------------------
    constexpr int NRA = G::NRA, NRB = G::NRB, NR3 = G::NR3, NRH = G::NRH, LAG_A = G::LAG_A, LAG_B = G::LAG_B, LAG_H = G::LAG_H, NRI = G::NRI, NRC = G::NRC;
    (void)LAG_B;
    constexpr int IMG_SLOT_B = G::IMG_SLOT_B, CV_PLANE_B = G::CV_PLANE_B, CV_SLOT_B = G::CV_SLOT_B, STAGE_SLOT_B = G::STAGE_SLOT_B;
    constexpr int OFF_BRING = G::A_RING_B, OFF_R3 = OFF_BRING + G::B_RING_B, OFF_RA = OFF_R3 + G::R3_B, OFF_IMG = OFF_RA + G::RA_B,
                  OFF_CV = OFF_IMG + G::IMG_B, OFF_STAGE = OFF_CV + G::CV_B, OFF_DUMP = OFF_STAGE + G::STAGE_B, OFF_OUT = OFF_DUMP + G::DUMP_B;
    constexpr int OFF_OUTF = OFF_OUT + G::OUT_B;
    (void)OFF_DUMP; (void)OFF_OUT; (void)OFF_OUTF;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    ie_fp16_saturate_on();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef IE_PAIR_SAME
#define IE_PAIR_SAME 0    // experiment: 1 = the two waves of a SIMD (w and w + 4) belong to the SAME stage (stage A = waves 0, 1, 4, 5; stage B = waves 2, 3, 6, 7)
#endif
#if IE_PAIR_SAME
    const int stg = (wave >> 1) & 1;
    const int sw = (wave & 1) | ((wave >> 2) << 1);
#else
    const int stg = wave >> 2;          // 0 = stage A, 1 = stage B
    const int sw = wave & 3;            // wave within the stage
#endif
    const int p16 = lane & 15, kq = lane >> 4;
    const int ph = sw & 1, ch = sw >> 1;

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = p.n_begin + b;
    const int xc0 = strip * TSK - MARG;                     // image column of computed column 0
    const int ys = p.y_begin + chunk * p.rows_per_block;
    const int ye = min(ys + p.rows_per_block, p.y_end);
    const int n_steps = (ye - ys) + 6 + LAG_A + LAG_H + G::OUT + (IS67 ? 1 : 0) + (KIND == K345 ? 4 : 0);
    constexpr int A_ROW0 = (KIND == K345) ? 4 : 2;          // stage A's HBM input row at step t is ys - A_ROW0 + t
    const int H = p.H, W = p.W;
    const size_t plane = (size_t)H * W;
    const size_t frame_b = plane * PXB;

    // ---- buffer descriptors -----------------------------------------------------------------------
    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    };
    const auto rsA0 = mk((const char*)(NSA >= 1 ? p.a_src0 : p.b_w) + (NSA >= 1 ? (size_t)n * frame_b : 0), frame_b);
    const auto rsA1 = mk((const char*)(NSA >= 2 ? p.a_src1 : p.b_w) + (NSA >= 2 ? (size_t)n * frame_b : 0), frame_b);
    const auto rsB = mk((const char*)(IS67 ? p.b_src : p.b_w) + (IS67 ? (size_t)n * frame_b : 0), frame_b);
    const auto rsI = mk((const char*)((KIND != K45 && KIND != K345) ? (const void*)(p.img + (size_t)n * 3 * plane) : p.b_w), 3 * plane * 4);
    const auto rsDa = mk((char*)(KIND == K12 ? p.a_dst : (void*)p.b_w) + (KIND == K12 ? (size_t)n * frame_b : 0), frame_b);
    const auto rsDb = mk((char*)(!IS67 ? p.b_dst : (void*)p.b_w) + (!IS67 ? (size_t)n * frame_b : 0), frame_b);
    // K67N: the three outputs through buffer descriptors (an out-of-range offset drops a masked lane's store; absent outputs get an empty range)
    const auto rsE = mk((IS67N && p.enhanced) ? (const void*)(p.enhanced + (size_t)n * 3 * plane) : p.b_w, (IS67N && p.enhanced) ? 3 * plane * 4 : 0);
    const auto rsJ = mk((IS67N && p.adjust) ? (const void*)(p.adjust + (size_t)n * 24 * plane) : p.b_w, (IS67N && p.adjust) ? 24 * plane * 4 : 0);
    const auto rsU = mk((IS67N && p.u8) ? (const void*)(p.u8 + (size_t)n * 3 * plane) : p.b_w, (IS67N && p.u8) ? 3 * plane : 0);

    // ---- weights of this wave's stage into registers -------------------------------------------
    frag wv[NW];
    {
        const int nfr = stg ? NFB : (KIND == K12 ? 2 : 9 * NSA);   // fragments per cout half
        const frag* wp = (const frag*)(stg ? p.b_w : p.a_w) + ((IS67N && stg) ? 0 : (size_t)ch * nfr * 64);     // K67N stage B: both tiles, [tile][tap]
#pragma unroll
        for (int f = 0; f < NW; ++f) wv[f] = wp[(f < nfr ? f : 0) * 64 + lane];
        if constexpr (KIND == K345) {       // stage A: e_conv3 in wv[0..8], e_conv4 in wv[9..17]
            if (stg == 0) {
                const frag* wp2 = (const frag*)p.a2_w + (size_t)ch * 9 * 64;
#pragma unroll
                for (int f = 0; f < 9; ++f) wv[9 + f] = wp2[f * 64 + lane];
            }
        }
        if constexpr (KIND == K12) {        // stage A: e_conv1 in wv[0..1], the x1 half of e_conv7 in wv[9..17] (packed behind e_conv2)
            if (stg == 0) {
                const frag* wp2 = (const frag*)p.b_w + (size_t)ch * NFB * 64;
#pragma unroll
                for (int f = 9; f < 18; ++f) wv[f] = wp2[f * 64 + lane];
            }
        }
    }
    ie_f32x4 bias4, bias4p;
    {
        const float* bp = stg ? p.b_bias : p.a_bias;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = (IS67 && stg) ? 0.f : bp[16 * ch + 4 * kq + i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bias4p[i] = (KIND == K12) ? p.b2_bias[16 * ch + 4 * kq + i] : ((KIND == K345 && stg == 0) ? p.a2_bias[16 * ch + 4 * kq + i] : 0.f);
    }

    // ---- DMA geometry ------------------------------------------------------------------------------
    unsigned body_xoff[2];   // byte offset inside an image row, or OOB (adding a row offset < 2^31 keeps it OOB)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (2 * sw + j) * 16 + (lane >> 2);
        const int s = (lane & 3) ^ ((q >> 1) & 3);
        const int x = xc0 + q;
        body_xoff[j] = (x >= 0 && x < W) ? (unsigned)(x * PXB + s * 16) : OOB;
    }
    unsigned halo_xoff;
    {
        const int x = (lane < 4) ? (xc0 - 1) : (xc0 + TW);
        halo_xoff = (lane < 8 && x >= 0 && x < W) ? (unsigned)(x * PXB + (lane & 3) * 16) : OOB;
    }
    // one ring row of one tensor: 2 body DMAs per wave + the halo DMA (wave 0 of the stage).
    // (a macro, not a lambda: the buffer-resource type cannot be a host-side function parameter)
#define IE_DMA_ROW(RS, SLOT, Y)                                                                                        \
    do {                                                                                                               \
        char* slot_ = (SLOT);                                                                                          \
        const int y_ = (Y);                                                                                            \
        const unsigned rmask_ = ((y_ >= 0) && (y_ < H)) ? 0u : OOB;                                                    \
        const unsigned row_b_ = rmask_ ? 0u : (unsigned)y_ * (unsigned)W * PXB;                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                \
            const unsigned off_ = (row_b_ + body_xoff[j]) | rmask_;                                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(slot_ + (2 * sw + j) * 1024), \
                                                     16, (int)off_, 0, 0, 0);                                          \
        }                                                                                                              \
        if (HALO && sw == 0) {                                                                                         \
            const unsigned off_ = (row_b_ + halo_xoff) | rmask_;                                                       \
            if (lane < 8)                                                                                              \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(slot_ + TW * PXB), 16, \
                                                         (int)off_, 0, 0, 0);                                          \
        }                                                                                                              \
    } while (0)
    // Any body piece m (16 pixels, 1 KiB) from any wave: the swizzle term does not depend on m ((16m + l) >> 1 & 3 == (l >> 1) & 3),
    // so piece m's per-lane source offset is piece 0's + 1024 m, valid while its column is inside the image.
    const int pc_x = xc0 + (lane >> 2);
    const int pc_off = pc_x * PXB + (((lane & 3) ^ ((lane >> 3) & 3)) << 4);
#define IE_DMA_PIECE(RS, SLOT, M, ROWB, RMASK)                                                                          \
    do {                                                                                                               \
        const int m_ = (M);                                                                                            \
        const bool ok_ = (unsigned)(pc_x + 16 * m_) < (unsigned)W;                                                     \
        const unsigned off_ = (ok_ ? (ROWB) + (unsigned)(pc_off + 1024 * m_) : OOB) | (RMASK);                         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)((SLOT) + m_ * 1024), 16, \
                                                 (int)off_, 0, 0, 0);                                                  \
    } while (0)
    // K12 image ring: element e = 64*(2*sw+j) + lane over [3 planes][132 px]; px <-> x = xc0 - 1 + px
    unsigned img12_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = 64 * (2 * sw + j) + lane;
        const int pl = e / 132, px = e - 132 * pl;
        const int x = xc0 - 1 + px;
        img12_off[j] = (pl < 3 && px < 130 && x >= 0 && x < W) ? (unsigned)(pl * plane * 4 + (size_t)x * 4) : OOB;
    }
    // K67 curve image ring: wave sw < 3 loads plane sw, columns xc0 + 64j + lane
    unsigned img67_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = xc0 + 64 * j + lane;
        img67_off[j] = (x >= 0 && x < W) ? (unsigned)(x * 4) : OOB;
    }

    // issue the DMA for step `t` (this wave's stage only)
    auto issue = [&](int t) __attribute__((always_inline)) {
        if (stg == 0) {
            if constexpr (NSA >= 1) IE_DMA_ROW(rsA0, smem + (t % NRA) * ROW_B, ys - A_ROW0 + t);
            if constexpr (NSA >= 2) IE_DMA_ROW(rsA1, smem + NRA * ROW_B + (t % NRA) * ROW_B, ys - A_ROW0 + t);
            if constexpr (IS67N && IE_K67N_PA) {      // the P row that seeds stage B's output row started at step t
                const int y = ys - 3 - G::DEFA + t;
                const unsigned rmask = ((y >= 0) && (y < H)) ? 0u : OOB;
                const unsigned row_b = rmask ? 0u : (unsigned)y * (unsigned)W * PXB;
                char* slot = smem + OFF_BRING + (t % NRB) * ROW_B;
#pragma unroll
                for (int j = 0; j < 2; ++j) IE_DMA_PIECE(rsB, slot, 2 * sw + j, row_b, rmask);
            }
            if constexpr (KIND == K12 && !(IE_EXP & 1)) {
                const int y = ys - 2 + t;
                const unsigned rmask = ((y >= 0) && (y < H)) ? 0u : OOB;
                const unsigned row_b = rmask ? 0u : (unsigned)y * (unsigned)W * 4u;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned off = (img12_off[j] + row_b) | rmask;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rsI, (__attribute__((address_space(3))) void*)(smem + OFF_IMG + (t % NRI) * IMG_SLOT_B + (2 * sw + j) * 256), 4,
                        (int)off, 0, 0, 0);
                }
            }
        } else {
            if constexpr (IS67N) {
                if constexpr (!IE_K67N_PA) {
                // the P row that seeds the output row started at step t: wave sw reads (and therefore loads) only its own 32 pixels = pieces 2 sw, 2 sw + 1
                const int y = ys - 3 - G::DEFA + t;
                const unsigned rmask = ((y >= 0) && (y < H)) ? 0u : OOB;
                const unsigned row_b = rmask ? 0u : (unsigned)y * (unsigned)W * PXB;
                char* slot = smem + OFF_BRING + (t % NRB) * ROW_B;
#pragma unroll
                for (int j = 0; j < 2; ++j) IE_DMA_PIECE(rsB, slot, 2 * sw + j, row_b, rmask);
                }
                if (sw < 3) {     // plane sw of the fp32 image row whose curve runs at step t (y = ys-6+t), read there by all four waves
                    const int yi = ys - 6 - G::DEFA + t;
                    const unsigned imask = ((yi >= ys) && (yi < ye)) ? 0u : OOB;
                    const unsigned base = imask ? 0u : ((unsigned)sw * (unsigned)plane + (unsigned)yi * (unsigned)W) * 4u;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            rsI, (__attribute__((address_space(3))) void*)(smem + OFF_IMG + (t % NRI) * IMG_SLOT_B + sw * G::IMG_PLANE67N + j * 256),
                            4, (int)((base + img67_off[j]) | imask), 0, 0, 0);
                }
            }
            if constexpr (KIND == K67) {
                // Waves 4 and 5 carry the curve of colour channel 2 and are the slowest of the step (in-kernel stamps); waves 6 and 7
                // idle a third of it, so THEY issue all of stage B's DMA: the P row that seeds the output row started at step t
                // (8 body pieces; P is read element-wise, it has no halo) and the three fp32 image planes of the row whose curve runs at step t+2.
                if (IE_K67_CURVE2 ? sw == 3 : sw >= 2) {
                    {
                        const int y = ys - 3 + t;
                        const unsigned rmask = ((y >= 0) && (y < H)) ? 0u : OOB;
                        const unsigned row_b = rmask ? 0u : (unsigned)y * (unsigned)W * PXB;
                        char* slot = smem + OFF_BRING + (t % NRB) * ROW_B;
                        if constexpr (!(IE_EXP & 8192)) {
                            if constexpr (IE_K67_CURVE2) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) IE_DMA_PIECE(rsB, slot, j, row_b, rmask);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) IE_DMA_PIECE(rsB, slot, (sw - 2) * 4 + j, row_b, rmask);
                            }
                        }
                    }
                    const int yo = ys - 5 + t;      // row finished at step t, consumed at step t+2
                    const unsigned rmask = ((yo >= ys) && (yo < ye)) ? 0u : OOB;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        if constexpr (IE_EXP & 4096) continue;
                        if (!IE_K67_CURVE2 && (pl == 0) != (sw == 2)) continue;          // (two-wave form) wave 6: plane 0; wave 7: planes 1, 2
                        const unsigned base = rmask ? 0u : ((unsigned)pl * (unsigned)plane + (unsigned)yo * (unsigned)W) * 4u;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const unsigned off = (base + img67_off[j]) | rmask;
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                rsI, (__attribute__((address_space(3))) void*)(smem + OFF_IMG + (t % NRI) * IMG_SLOT_B + pl * (TW * 4) + j * 256),
                                4, (int)off, 0, 0, 0);
                        }
                    }
                }
            }
        }
    };
    // DMA instructions per step of this wave -> counted wait that leaves PF-1 younger rows in flight
    // Waves that issue no DMA do not wait at all.  gfx950 counts loads and stores in the one in-order vmcnt: K12's stage A issues
    // exactly two (possibly out-of-range = dropped) P stores per step after its two loads, which its wait adds in (IE_STEP_HEAD's
    // argument); K67's curve stores only make the wait stricter than needed.
    int L = 0, S = 0;
    if (stg == 0) { L = (KIND == K12) ? 2 : (2 * NSA + ((HALO && sw == 0) ? NSA : 0)) + ((IS67N && IE_K67N_PA) ? 2 : 0); }   // K345: NSA = 1
    else if (KIND == K67) L = IE_K67_CURVE2 ? (sw == 3 ? 14 : 0) : (sw == 2 ? 6 : (sw == 3 ? 8 : 0));
    else if (IS67N) L = (sw < 3 ? 2 : 0) + (IE_K67N_PA ? 0 : 2);
    const int wait_base = L ? (PF - 1) * L : -1;
    (void)S;

    // ---- LDS read offsets of the B operand (as in zdce_conv3x3_mfma) --------------------------------
    auto lds_px = [&](int q) __attribute__((always_inline)) { return q * PXB + ((kq ^ ((q >> 1) & 3)) << 4); };
    int rd_base[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) rd_base[kx] = lds_px(64 * ph + 16 + p16 + kx - 1) - 1024;
    int rd_first, rd_last;
    {
        const int xf = 64 * ph + p16 - 1;
        rd_first = lds_px(xf < 0 ? (HALO ? TW : 0) : xf);             // without halo DMA: any loaded pixel (the column is dropped)
        const int xl = 64 * ph + 48 + p16 + 1;
        rd_last = lds_px(xl >= TW ? (HALO ? TW + 1 : TW - 1) : xl);
    }

    // K67N stage B: wave sw owns columns 32 sw .. 32 sw + 31 (two 16-pixel tiles nt) and BOTH 16-row tiles of e_conv7
    int rdn[2][3] = {{0, 0, 0}, {0, 0, 0}};   // pixel fragment of tile nt at tap column kx (the two out-of-strip columns are clamped: dropped columns)
    int sd_off[2] = {0, 0};                   // this lane's 16 bytes of seeds in a P ring row: chunk kq of its pixel
    int im_off[2] = {0, 0};                   // ... and its colour of the fp32 pixel in an image ring row
    int of_off[2] = {0, 0};                   // where its curve result goes in the fp32 output row
    unsigned est_off = OOB;                   // enhanced, W % 4 == 0: byte offset of this lane's 16-byte store inside a row of its colour plane, or OOB
    int est_lds = 0;                          // ... and where those four pixels sit in the staged output row
    if constexpr (IS67N) {
        {   // wave sw < 3 stores colour sw; lane l < 31: pixels 4 l .. 4 l + 3 of the strip's 124 stored ones (first stored column = a multiple of 4)
            const int x = xc0 + MARG + 4 * lane;
            est_off = (sw < 3 && lane < 31 && x < W) ? (unsigned)(((size_t)sw * plane + x) * 4) : OOB;
            est_lds = (sw < 3 ? sw : 0) * G::OUTF_PLANE_B + 16 + 16 * (lane < 31 ? lane : 0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int q = 32 * sw + 16 * nt + p16;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int qq = q + kx - 1;
                qq = qq < 0 ? 0 : (qq > TW - 1 ? TW - 1 : qq);
                rdn[nt][kx] = lds_px(qq);
            }
            sd_off[nt] = q * PXB + ((kq ^ ((q >> 1) & 3)) << 4);
            im_off[nt] = (kq < 3 ? kq : 2) * G::IMG_PLANE67N + q * 4;
            of_off[nt] = kq * G::OUTF_PLANE_B + (q + 2) * 4;        // (kq = 3: the spare plane)
        }
    }

    // one source row into the three live accumulator sets.  The set started by this row (accN, ky = 0) takes its initial value
    // (`init`: the bias) as the C operand of its first MFMA when `first` is set -- no reset moves, and whatever still reads the
    // previous contents of accN (the epilogue of the row finished one step earlier) only has to come before THAT MFMA.
    // nv (a std::integral_constant): vector instructions the scheduler is asked to place after every MFMA.  Independent VALU work
    // of the same basic block (the deferred epilogue, address arithmetic) then rides in the MFMAs' issue shadow -- two per MFMA are
    // nearly free in a wave's own stream, whereas a VALU-only phase beside the SIMD partner's MFMA-only phase costs almost its full
    // length (tools/micro/coissue.hip, profiles/r02/coissue_microbench.log).
    auto src_mfma = [&](auto tg, const char* row, int wbase, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], bool first,
                        const ie_f32x4& init, auto nv) __attribute__((always_inline)) {
        using TT = typename decltype(tg)::type;
        constexpr int NV = decltype(nv)::value;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int off = (kx == 0 && nt == 0) ? rd_first : (kx == 2 && nt == 3) ? rd_last : rd_base[kx] + 1024 * nt;
                const frag bx = (IE_EXP & 8) ? wv[wbase + (kx + nt) % 9] : *(const frag*)(row + off);
                accN[nt] = ie_mfma16_raw<TT>(wv[wbase + 0 * 3 + kx], bx, (first && kx == 0) ? init : accN[nt]);
                accM[nt] = ie_mfma16_raw<TT>(wv[wbase + 1 * 3 + kx], bx, accM[nt]);
                accO[nt] = ie_mfma16_raw<TT>(wv[wbase + 2 * 3 + kx], bx, accO[nt]);
            }
#ifndef IE_LDS_DEPTH
#define IE_LDS_DEPTH 3
#endif
        __builtin_amdgcn_sched_group_barrier(0x100, IE_LDS_DEPTH, 0);
        if constexpr (NV == 0) {
#pragma unroll
            for (int f = 0; f < 12 - IE_LDS_DEPTH; ++f) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * IE_LDS_DEPTH, 0);
        } else {
#pragma unroll
            for (int f = 0; f < 12; ++f) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                }
                if (f < 12 - IE_LDS_DEPTH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };
    constexpr std::integral_constant<int, 0> NV0{};
    constexpr std::integral_constant<int, 2> NV2{};
    (void)NV0; (void)NV2;

    // write a finished stage-A row (ReLU, 16-bit, zero outside the image) into the hand-off ring slot
    auto a_emit_to = [&](auto tg, const ie_f32x4 (&acc)[4], char* slot, int ra, bool relu = true) __attribute__((always_inline)) {
        using TE = typename decltype(tg)::type;
        const bool row_in = (ra >= 0) && (ra < H);
        if constexpr (IE_EMIT_B128) {
#pragma unroll
            for (int nt = 0; nt < 4; nt += 2) {
                ie_u32x2 pk[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    pk[h][0] = relu ? ie_pack2_relu<TE>(acc[nt + h][0], acc[nt + h][1]) : ie_pack2<TE>(acc[nt + h][0], acc[nt + h][1]);
                    pk[h][1] = relu ? ie_pack2_relu<TE>(acc[nt + h][2], acc[nt + h][3]) : ie_pack2<TE>(acc[nt + h][2], acc[nt + h][3]);
                }
                // even-kq lanes end with couts 4kq .. 4kq+7 of tile nt, odd-kq lanes with couts 4(kq-1) .. +7 of tile nt+1
                const auto sx = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                const int q = 64 * ph + 16 * (nt + (kq & 1)) + p16;
                const int x = xc0 + q;
                const bool in_img = row_in && x >= 0 && x < W;
                const ie_u32x4 v = {in_img ? sx[0] : 0u, in_img ? sy[0] : 0u, in_img ? sx[1] : 0u, in_img ? sy[1] : 0u};
                const int c = 2 * ch + (kq >> 1);      // 16-byte chunk holding couts 16ch + 8 (kq>>1) .. +7
                *(ie_u32x4*)(slot + q * PXB + ((c ^ ((q >> 1) & 3)) << 4)) = v;
            }
            return;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int q = 64 * ph + 16 * nt + p16;
            const int x = xc0 + q;
            const bool in_img = row_in && x >= 0 && x < W;
            ie_u32x2 pk;
            pk[0] = in_img ? (relu ? ie_pack2_relu<TE>(acc[nt][0], acc[nt][1]) : ie_pack2<TE>(acc[nt][0], acc[nt][1])) : 0u;
            pk[1] = in_img ? (relu ? ie_pack2_relu<TE>(acc[nt][2], acc[nt][3]) : ie_pack2<TE>(acc[nt][2], acc[nt][3])) : 0u;
            const int c = 2 * ch + (kq >> 1);      // 16-byte chunk holding couts 16ch+4kq .. +3
            *(ie_u32x2*)(slot + q * PXB + ((c ^ ((q >> 1) & 3)) << 4) + (kq & 1) * 8) = pk;
        }
    };
    auto a_emit = [&](const ie_f32x4 (&acc)[4], int t) __attribute__((always_inline)) {
        a_emit_to(ie_tag<TH>{}, acc, smem + OFF_RA + (t % NRH) * ROW_B, ys - 3 - LAG_A + t);
    };

    // loop-invariant lane geometry of the deferred epilogues
    unsigned st_col[2];     // byte offset of this lane's 16-byte store of piece 2 sw + j inside an output row, or OOB
    int st_lds[2];          // ... and where that chunk sits in the parked row
    int em_off[4];          // hand-off write of tile nt: byte offset inside the ring slot
    int em16_off[2];        // ... of tile pair j as 16-byte pieces (after the lane exchange: even kq -> tile 2j, odd kq -> tile 2j+1)
    unsigned em16_in = 0;   // bit j: this lane's pixel of pair j lies inside the image
    unsigned em_in = 0;     // bit nt: tile nt's column lies inside the image
    if constexpr (KIND == K12 || KIND == K345 || IS67N) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            {   // piece m = 2 sw + j of the parked row: 16 pixels x 64 bytes, lane = (pixel, 16-byte chunk)
                const int q = 16 * (2 * sw + j) + (lane >> 2), xo = xc0 + q, c = lane & 3;
                st_col[j] = (q >= MARG && q < TW - MARG && xo < W) ? (unsigned)(xo * PXB + c * 16) : OOB;
                st_lds[j] = q * PXB + ((c ^ ((q >> 1) & 3)) << 4);
            }
            if constexpr (!G::OUT) {
            const int q = 64 * ph + 16 * (2 * j + (kq & 1)) + p16, xo = xc0 + q;
            st_col[j] = (q >= MARG && q < TW - MARG && xo < W) ? (unsigned)(xo * PXB + (16 * ch + 4 * (kq & 2)) * 2) : OOB;
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int q = 64 * ph + 16 * nt + p16, x = xc0 + q, c = 2 * ch + (kq >> 1);
            em_off[nt] = q * PXB + ((c ^ ((q >> 1) & 3)) << 4) + (kq & 1) * 8;
            em_in |= (x >= 0 && x < W) ? (1u << nt) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 64 * ph + 16 * (2 * j + (kq & 1)) + p16, x = xc0 + q, c = 2 * ch + (kq >> 1);
            em16_off[j] = q * PXB + ((c ^ ((q >> 1) & 3)) << 4);
            em16_in |= (x >= 0 && x < W) ? (1u << j) : 0u;
        }
    }
#define IE_PACK_TILE(TT, A, RELU)                                                                                          \
    ie_u32x2{(RELU) ? ie_pack2_relu<TT>((A)[0], (A)[1]) : ie_pack2<TT>((A)[0], (A)[1]),                                    \
             (RELU) ? ie_pack2_relu<TT>((A)[2], (A)[3]) : ie_pack2<TT>((A)[2], (A)[3])}
// tile pair J of a packed row into an LDS row image as 16-byte pieces (conflict-free ds_write_b128); ZERO: blank pixels outside the image
#define IE_PARK_PAIR(SLOT, P0, P1, J, KEEP)                                                                               \
    do {                                                                                                                  \
        auto sx_ = __builtin_amdgcn_permlane16_swap((P0)[0], (P1)[0], false, false);                                      \
        auto sy_ = __builtin_amdgcn_permlane16_swap((P0)[1], (P1)[1], false, false);                                      \
        const bool k_ = (KEEP);                                                                                           \
        *(ie_u32x4*)((SLOT) + em16_off[J]) = ie_u32x4{k_ ? sx_[0] : 0u, k_ ? sy_[0] : 0u, k_ ? sx_[1] : 0u, k_ ? sy_[1] : 0u}; \
    } while (0)
#define IE_STORE_PAIR(RS, P0, P1, J, ROWB, RMASK)                                                                          \
    do {                                                                                                                  \
        auto sx_ = __builtin_amdgcn_permlane16_swap((P0)[0], (P1)[0], false, false);                                      \
        auto sy_ = __builtin_amdgcn_permlane16_swap((P0)[1], (P1)[1], false, false);                                      \
        ie_u32x4 v_ = {sx_[0], sy_[0], sx_[1], sy_[1]};                                                                   \
        __builtin_amdgcn_raw_buffer_store_b128(v_, RS, (int)(((ROWB) + st_col[J]) | (RMASK)), 0, 0);                       \
    } while (0)

    // stage-B epilogue to HBM (centre 126 columns): ReLU'd activations (x2, x5) or the raw P map (K12)
#define IE_B_STORE(ACC, YO, RS, RELU)                                                                                     \
    do {                                                                                                                  \
        if constexpr (IE_EXP & 64) { /* keep the MFMAs alive with no VALU: raw accumulators to an out-of-range offset */   \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                            \
                ie_u32x4 v_ = {__float_as_uint(ACC[nt][0]), __float_as_uint(ACC[nt][1]), __float_as_uint(ACC[nt][2]), __float_as_uint(ACC[nt][3])}; \
                __builtin_amdgcn_raw_buffer_store_b128(v_, RS, (int)OOB, 0, 0);                                           \
            }                                                                                                             \
            break;                                                                                                        \
        }                                                                                                                 \
        const int yo_ = (YO);                                                                                             \
        const bool row_ok_ = (yo_ >= ys) && (yo_ < ye);                                                                   \
        ie_u32x2 pk_[4];                                                                                                  \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                                \
            pk_[nt][0] = (RELU) ? ie_pack2_relu<TO>(ACC[nt][0], ACC[nt][1]) : ie_pack2<TO>(ACC[nt][0], ACC[nt][1]);       \
            pk_[nt][1] = (RELU) ? ie_pack2_relu<TO>(ACC[nt][2], ACC[nt][3]) : ie_pack2<TO>(ACC[nt][2], ACC[nt][3]);       \
        }                                                                                                                 \
        const unsigned row_b_ = (unsigned)yo_ * (unsigned)W * PXB;                                                        \
        _Pragma("unroll") for (int nt = 0; nt < 4; nt += 2) {                                                             \
            auto sx_ = __builtin_amdgcn_permlane16_swap(pk_[nt][0], pk_[nt + 1][0], false, false);                        \
            auto sy_ = __builtin_amdgcn_permlane16_swap(pk_[nt][1], pk_[nt + 1][1], false, false);                        \
            ie_u32x4 v_ = {sx_[0], sy_[0], sx_[1], sy_[1]};                                                               \
            const int q_ = 64 * ph + 16 * (nt + (kq & 1)) + p16;                                                          \
            const int xo_ = xc0 + q_;                                                                                     \
            const bool ok_ = row_ok_ && q_ >= MARG && q_ < TW - MARG && xo_ < W;                                          \
            const unsigned off_ = ok_ ? row_b_ + (unsigned)xo_ * PXB + (unsigned)((16 * ch + 4 * (kq & 2)) * 2) : OOB;         \
            __builtin_amdgcn_raw_buffer_store_b128(v_, RS, (int)off_, 0, 0);                                              \
        }                                                                                                                 \
    } while (0)

    // K67 stage B: seed the accumulators of the output row started at step t with its P row (from the DMA ring)
    auto seed_from_p = [&](int t, ie_f32x4 (&acc)[4]) __attribute__((always_inline)) {
        const char* slot = smem + OFF_BRING + (t % NRB) * ROW_B;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int q = 64 * ph + 16 * nt + p16;
            const int c = 2 * ch + (kq >> 1);
            typedef __attribute__((ext_vector_type(4))) TH t4;
            const t4 v = *(const t4*)(slot + q * PXB + ((c ^ ((q >> 1) & 3)) << 4) + (kq & 1) * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nt][i] = (float)v[i];
        }
    };

    // K67 curve epilogue.  Stage B parks the raw e_conv7 outputs (bias included) of the row it finished
    // in the previous step in LDS as [24][128] fp32 (double buffered); one step later ALL 8 waves run
    // tanh + the 8-step LE curve, thread = (column, colour channel), on the fp32 input pixel from the
    // image ring, and write enhanced / u8 / adjust with coalesced stores.  No extra barrier: the
    // per-step barrier orders write -> read -> rewrite.
    auto stage_write = [&](const ie_f32x4 (&acc)[4], int slot) __attribute__((always_inline)) {
        // row k is stored rotated by 16*((k>>2)&1) columns: the 32 lanes of a ds_write_b32 group
        // (2 kq values x 16 columns) then hit 32 distinct banks
        float* sp = (float*)(smem + OFF_STAGE + slot * STAGE_SLOT_B);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 16 * ch + 4 * kq + i;
                if (k < 24) sp[k * TW + ((64 * ph + 16 * nt + p16 + 16 * (kq & 1)) & (TW - 1))] = acc[nt][i];
            }
    };
    // 384 of the 512 threads: thread = (column q, colour channel c); s_fin: step at which the row was finished
    auto curve_consume = [&](int s_fin) __attribute__((always_inline)) {
        const int yo = ys - 5 + s_fin;
        const int q = tid & 127, c = tid >> 7;       // c = 0,1 on stage-A waves, 2 on stage-B waves 4,5; 3: idle
        const int xo = xc0 + q;
        if (c < 3 && q >= MARG && q < TW - MARG && xo < W && yo >= ys && yo < ye) {
            const float* sp = (const float*)(smem + OFF_STAGE + (s_fin & 1) * STAGE_SLOT_B);
            const float* ip = (const float*)(smem + OFF_IMG + (s_fin % NRI) * IMG_SLOT_B);
            const size_t pix = (size_t)yo * W + xo;
            float v = ip[c * TW + q];
            float r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 3 * i + c;
                r[i] = (IE_EXP & 2048) ? sp[k * TW + ((q + 16 * ((k >> 2) & 1)) & (TW - 1))] : ie_tanh_fast(sp[k * TW + ((q + 16 * ((k >> 2) & 1)) & (TW - 1))]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v = ie_curve_step_fma(v, r[i]);
            if (p.enhanced) p.enhanced[((size_t)n * 3 + c) * plane + pix] = v;
            if (p.u8) p.u8[((size_t)n * plane + pix) * 3 + c] = ie_quant_u8(v);
            if (p.adjust) {
                float* ap = p.adjust + (size_t)n * 24 * plane + pix;
#pragma unroll
                for (int i = 0; i < 8; ++i) ap[(size_t)(3 * i + c) * plane] = r[i];
            }
        }
    };

    // The same on 192 threads: stage-B waves 0..2 take colour channel c = sw each, thread = columns lane and lane + 64, arithmetic in
    // packed fp32 (v_pk_mul / v_pk_add / v_pk_fma: half the vector instructions for everything but the two transcendentals per tanh).
    // The fourth stage-B wave issues all of the stage's DMA instead, stage A (72 MFMAs per wave and row) carries no curve work.
    auto curve_consume2 = [&](int s_fin) __attribute__((always_inline)) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int yo = ys - 5 + s_fin;
        const int c = sw, q0 = lane, q1 = lane + 64;
        const bool row_ok = (yo >= ys) && (yo < ye);
        const bool ok0 = row_ok && q0 >= MARG && xc0 + q0 < W, ok1 = row_ok && q1 < TW - MARG && xc0 + q1 < W;
        const float* sp = (const float*)(smem + OFF_STAGE + (s_fin & 1) * STAGE_SLOT_B);
        const float* ip = (const float*)(smem + OFF_IMG + (s_fin % NRI) * IMG_SLOT_B);
        f2 v = {ip[c * TW + q0], ip[c * TW + q1]};
        f2 r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 3 * i + c, rot = 16 * ((k >> 2) & 1);
            const f2 a = {sp[k * TW + ((q0 + rot) & (TW - 1))], sp[k * TW + ((q1 + rot) & (TW - 1))]};
            if constexpr (IE_EXP & 2048) { r[i] = a; continue; }
            const f2 a2 = a * 2.885390081777927f;                                             // exp(2x) = exp2(x * 2 / ln 2)
            const f2 e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
            const f2 d = e + 1.0f;
            const f2 rc = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            r[i] = __builtin_elementwise_fma(f2{-2.0f, -2.0f}, rc, f2{1.0f, 1.0f});          // 1 - 2 / (exp(2x) + 1), like ie_tanh_fast
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v = __builtin_elementwise_fma(r[i], __builtin_elementwise_fma(v, v, -v), v);   // ie_curve_step_fma on both columns
        const size_t pix = (size_t)yo * W + xc0;
        if (p.enhanced) {
            float* ep = p.enhanced + ((size_t)n * 3 + c) * plane + pix;
            if (ok0) ep[q0] = v[0];
            if (ok1) ep[q1] = v[1];
        }
        if (p.u8) {
            unsigned char* up = p.u8 + ((size_t)n * plane + pix) * 3 + c;
            if (ok0) up[(size_t)q0 * 3] = ie_quant_u8(v[0]);
            if (ok1) up[(size_t)q1 * 3] = ie_quant_u8(v[1]);
        }
        if (p.adjust) {
            float* ap = p.adjust + (size_t)n * 24 * plane + pix;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (ok0) ap[(size_t)(3 * i + c) * plane + q0] = r[i][0];
                if (ok1) ap[(size_t)(3 * i + c) * plane + q1] = r[i][1];
            }
        }
    };

    // K12 stage A: e_conv1 on the MFMA.  Every landed fp32 image row is converted once into a 16-bit
    // pixel-interleaved row [px][R,G,B,0] (hi part, and for bf16 the lo = x - hi part, so the fp32
    // pixel survives bf16's 8 mantissa bits).  K layout of the two K=32 MFMAs per 16-pixel tile:
    //   MFMA m, lane group kq:  ky = 2m + (kq>>1),  kx = 2(kq&1) + (j>>2),  c = j&3     (ky,kx = 3: zero weights)
    // so a lane's 8 k-values are two adjacent interleaved pixels = 16 contiguous bytes of the row.
    auto convert_row = [&](int t) __attribute__((always_inline)) {
        // branch-free (one basic block with stage B's MFMAs): lanes past the row read a clamped pixel and write into a per-lane dump
        const int px = sw * 64 + lane;      // row R(t) = ys-2+t landed in fp32 slot t % NRI
        {                                   // px 130/131 are read by the zero-weight pad tap: keep them finite (0)
            const float* ip = (const float*)(smem + OFF_IMG + (t % NRI) * IMG_SLOT_B) + (px < 131 ? px : 131);
            const bool real = px < 130;
            const float r0 = ip[0], g0 = ip[132], b0 = ip[264];
            const float r = real ? r0 : 0.f, g = real ? g0 : 0.f, bl = real ? b0 : 0.f;
            const TH hr = (TH)r, hg = (TH)g, hb = (TH)bl;
            ie_u32x2 hi, lo;
            hi[0] = ie_pack2<TH>(r, g);
            hi[1] = ie_pack2<TH>(bl, 0.f);
            lo[0] = ie_pack2<TH>(r - (float)hr, g - (float)hg);
            lo[1] = ie_pack2<TH>(bl - (float)hb, 0.f);
            const bool keep = px < 132;
            char* cv = smem + (keep ? OFF_CV + (t % NRC) * CV_SLOT_B + px * 8 : OFF_DUMP + lane * 8);
            *(ie_u32x2*)cv = hi;
            *(ie_u32x2*)(cv + (keep ? CV_PLANE_B : 512)) = lo;
        }
    };
    auto conv1_row = [&](int t, ie_f32x4 (&acc)[4]) __attribute__((always_inline)) {
        // output row ra = ys-3-LAG_A+t from converted rows R(t-2-LAG_A+ky), ky = 0..2 (all converted before this barrier period);
        // this lane reads row ky = 2m + (kq>>1): slot (t - 2 - LAG_A + ky) % NRC
        const int pxo = (64 * ph + p16 + 2 * (kq & 1)) * 8;
        const int s0 = OFF_CV + ((t + NRC - 2 - LAG_A + (kq >> 1)) % NRC) * CV_SLOT_B + pxo;     // m = 0: ky = kq>>1
        const int s1 = OFF_CV + ((t + NRC - LAG_A) % NRC) * CV_SLOT_B + pxo;                     // m = 1: ky = 2 (kq < 2)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = bias4;
#pragma unroll
        for (int part = 0; part < (ie_elem<TH>::dtype == IE_BF16 ? 2 : 1); ++part) {
            ie_u32x4 b0[4], b1[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {       // all loads first: the four tiles are independent MFMA chains
                const char* r0 = smem + s0 + part * CV_PLANE_B + 128 * nt;
                const char* r1 = smem + s1 + part * CV_PLANE_B + 128 * nt;
                if constexpr (IE_EXP & 2) { b0[nt] = wv[0]; b1[nt] = wv[1]; continue; }
                const ie_u32x2 b0a = *(const ie_u32x2*)r0, b0b = *(const ie_u32x2*)(r0 + 8);
                const ie_u32x2 b1a = *(const ie_u32x2*)r1, b1b = *(const ie_u32x2*)(r1 + 8);
                b0[nt] = ie_u32x4{b0a[0], b0a[1], b0b[0], b0b[1]};
                b1[nt] = ie_u32x4{b1a[0], b1a[1], b1b[0], b1b[1]};
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = ie_mfma16_raw<TH>(wv[0], b0[nt], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = ie_mfma16_raw<TH>(wv[1], b1[nt], acc[nt]);
        }
    };

    // ---- K12, streamed form ------------------------------------------------------------------------
    // tools/micro/coissue.hip: a wave's vector instructions are nearly free when they sit between its OWN MFMAs (two per MFMA),
    // and cost almost their full issue time as a VALU-only phase beside the SIMD partner's MFMA-only phase.  So both stages run ONE
    // MFMA stream per step and everything else of the step is cut into small fillers placed after fixed MFMAs of that stream
    // (sched_barrier pins the order).  Fillers must not depend on the stream's results: every epilogue is deferred by one step
    // (the row finished at step t-1 still sits in the accumulator set that step t starts afresh; its tile nt is packed before
    // the first MFMA that overwrites tile nt), which costs the hand-off one more row of lag (G::DEF).
#define IE_SB() __builtin_amdgcn_sched_barrier(0)
    auto mfma_stream = [&](auto tg, const char* row, int wbase, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], bool first,
                           const ie_f32x4& init, auto&& fill) __attribute__((always_inline)) {
        using TT = typename decltype(tg)::type;
        frag bx[12];
        auto rd = [&](int g) __attribute__((always_inline)) {
            const int kx = g >> 2, nt = g & 3;
            const int off = (kx == 0 && nt == 0) ? rd_first : (kx == 2 && nt == 3) ? rd_last : rd_base[kx] + 1024 * nt;
            bx[g] = (IE_EXP & 8) ? wv[wbase + (kx + nt) % 9] : *(const frag*)(row + off);
        };
        rd(0); rd(1); rd(2);
        IE_SB();
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int kx = g >> 2, nt = g & 3;
            if (g + 3 < 12) rd(g + 3);
            accN[nt] = ie_mfma16_raw<TT>(wv[wbase + 0 * 3 + kx], bx[g], (first && kx == 0) ? init : accN[nt]);
            fill(3 * g);
            IE_SB();
            accM[nt] = ie_mfma16_raw<TT>(wv[wbase + 1 * 3 + kx], bx[g], accM[nt]);
            fill(3 * g + 1);
            IE_SB();
            accO[nt] = ie_mfma16_raw<TT>(wv[wbase + 2 * 3 + kx], bx[g], accO[nt]);
            fill(3 * g + 2);
            IE_SB();
        }
    };
#ifdef IE_STAMP
    unsigned long long st_wait = 0, st_issue = 0, st_work = 0, st_e[4] = {0, 0, 0, 0}, st_last = 0;
    // finer split of a step: IE_TX(i) adds the cycles since the previous IE_TX / IE_TX0 to bucket i
#define IE_TX0() do { IE_T(st_last); } while (0)
#define IE_TX(i) do { unsigned long long now_; IE_T(now_); st_e[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define IE_TX0() do { } while (0)
#define IE_TX(i) do { } while (0)
#endif
    // ---- one step -----------------------------------------------------------------------------------
    // Stage A: MFMAs first (the pipe is busy from the first cycle after the barrier), then its DMA
    // issue, then the hand-off write.  Stage B: the epilogue of the row it finished in the PREVIOUS step
    // first (that accumulator set is this step's accN, reset afterwards), then its MFMAs.  The two stages
    // share each SIMD, so B's epilogue VALU runs under A's MFMAs and A's tail under B's MFMAs.
    // Each stage runs its OWN loop (the wave-uniform `stg` branch is taken once, outside): the register allocator then sees two
    // independent live ranges -- stage A's second accumulator family (K345) or stage B's (K12) is not carried through the other
    // stage's loop -- and both loops execute exactly one s_barrier per step.
    // HOT0 / HOT1: the steady-state values of WAITN (compile-time; tested first, the 30-way switch is the prologue's path)
#define IE_STEP_HEAD(WAITN, HOT0, HOT1)                                                                            \
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;                                                     \
        (void)ts0; (void)ts1; (void)ts2; (void)ts3;                                                                \
        IE_T(ts0);                                                                                                 \
        if (RPB == 1 || (t % RPB) == 0) {                                                                          \
        if constexpr (IE_HOTWAIT & (KIND == K12 ? 1 : (KIND == K345 ? 2 : 4)))                                       \
            ie_wait_vm_hot<(HOT0), (HOT1)>((WAITN), [](int n_) __attribute__((always_inline)) { wait_vm(n_); });    \
        else wait_vm(WAITN);                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's LDS writes of step t-1 (hand-off, staging) */ \
        __builtin_amdgcn_s_barrier();                                                                              \
        }                                                                                                          \
        IE_T(ts1)
#ifdef IE_STAMP
#define IE_STEP_TAIL() do { IE_T(ts3); st_wait += ts1 - ts0; st_issue += ts2 - ts1; st_work += ts3 - ts2; } while (0)
#else
#define IE_STEP_TAIL() do { } while (0)
#endif
    // Stage A: MFMAs first (the pipe is busy from the first cycle after the barrier), then its DMA issue, then the hand-off write.
    auto step_a = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], ie_f32x4 (&pN)[4], ie_f32x4 (&pM)[4],
                      ie_f32x4 (&pO)[4]) __attribute__((always_inline)) {
        // K12: at the barrier before rows t .. t+RPB-1 the image DMA of row t+RPB-1 must have landed.  Ops younger than it: the
        // 2 (PF-RPB) loads of the rows behind it plus, for each of the min(t, PF-RPB+1) row steps since it was issued, two P stores.
        IE_STEP_HEAD(KIND == K12 ? 2 * (PF - RPB) + 2 * (t < PF - RPB + 1 ? t : PF - RPB + 1) : wait_base,
                     (KIND == K12 ? 2 * (PF - RPB) + 2 * (PF - RPB + 1) : (PF - 1) * (2 * NSA + ((IS67N && IE_K67N_PA) ? 2 : 0))), ((HALO && KIND != K12) ? (PF - 1) * 3 * NSA : -1));
        if constexpr (KIND == K12) {
            // e_conv1 on the converted image rows -> x1 row ys-4+t (accO); the x1 half of e_conv7 on the x1 row emitted at step t-1
            // (ys-5+t) -> P row ys-6+t complete in pO.  With e_conv7a here and e_conv2 on stage B the two waves of a SIMD carry 44 and
            // 36 MFMAs per row instead of 8 and 72, so each one's LDS / VALU / store work hides under the other's MFMAs.
            IE_TX0();
            if constexpr (!(IE_EXP & 32)) conv1_row(t, accO);
            IE_TX(0);
            src_mfma(ie_tag<TH>{}, smem + OFF_RA + ((t + RPB) % NRH) * ROW_B, 9, pN, pM, pO, true, bias4p, NV2);    // x1 row emitted at step t-RPB
            IE_TX(1);
            IE_T(ts2);
            issue(t + PF);
            IE_TX(2);
            if constexpr (!(IE_EXP & 4)) a_emit(accO, t);          // unconditional: rows emitted before the first valid one are never part of a stored row
            IE_TX(3);
            if constexpr (!(IE_EXP & 16)) IE_B_STORE(pO, ys - 5 - LAG_A - LAG_B + t, rsDa, false);         // always issued (rows outside the chunk go out of range): the vmcnt bookkeeping counts on it
        } else if constexpr (KIND == K345) {
            // e_conv3 on x2 row ys-4+t -> x3 row ys-5+t (accO); e_conv4 on the x3 row emitted at step t-1 -> x4 row ys-7+t (pO)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { accN[nt] = bias4; pN[nt] = bias4p; }
            src_mfma(ie_tag<TA0>{}, smem + (t % NRA) * ROW_B, 0, accN, accM, accO, false, bias4, NV0);
            src_mfma(ie_tag<T3>{}, smem + OFF_R3 + ((t + NR3 - 1) % NR3) * ROW_B, 9, pN, pM, pO, false, bias4, NV0);
            IE_T(ts2);
            issue(t + PF);
            if (t >= 2) a_emit_to(ie_tag<T3>{}, accO, smem + OFF_R3 + (t % NR3) * ROW_B, ys - 5 + t);
            if (t >= 5) a_emit_to(ie_tag<TH>{}, pO, smem + OFF_RA + (t & 1) * ROW_B, ys - 7 + t);
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) accN[nt] = bias4;
            src_mfma(ie_tag<TA0>{}, smem + (t % NRA) * ROW_B, 0, accN, accM, accO, false, bias4, NV0);
            if constexpr (NSA == 2) src_mfma(ie_tag<TA1>{}, smem + NRA * ROW_B + (t % NRA) * ROW_B, 9, accN, accM, accO, false, bias4, NV0);
            IE_T(ts2);
            issue(t + PF);
            if (t >= 2) a_emit(accO, t);
            if constexpr (KIND == K67 && !(IE_EXP & 1024) && !IE_K67_CURVE2) { if (t >= 7) curve_consume(t - 2); }
        }
        IE_STEP_TAIL();
    };
    // Stage B: the epilogue of the row it finished in the PREVIOUS step first (that accumulator set is this step's accN, reset
    // afterwards), then its MFMAs.  The two stages share each SIMD, so B's epilogue VALU runs under A's MFMAs and A's tail under B's.
    auto step_b = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], ie_f32x4 (&pN)[4], ie_f32x4 (&pM)[4],
                      ie_f32x4 (&pO)[4]) __attribute__((always_inline)) {
        IE_STEP_HEAD(wait_base, (KIND == K67 ? (PF - 1) * 6 : -1), (KIND == K67 ? (PF - 1) * 8 : -1));
        issue(t + PF);
        IE_T(ts2);
        // accN (and pN) still hold the output row finished at step t-1: ys-5-LAG_A+(t-1)
        IE_TX0();
        if constexpr (KIND == K12 && !(IE_EXP & 1)) convert_row(t);
        IE_TX(0);
        if constexpr (KIND == K12) {
            if constexpr (!(IE_EXP & 16)) IE_B_STORE(accN, ys - 6 - LAG_A - LAG_B + t, rsDb, true);      // branch-free (rows outside the chunk go out of range): one basic block with the MFMAs
            IE_TX(1);
        } else if constexpr (KIND == K45) {
            if (t >= 6) IE_B_STORE(accN, ys - 6 + t, rsDb, true);
        } else if constexpr (KIND == K345) {
            if constexpr (!G::OUT) { if (t >= 10) IE_B_STORE(accN, ys - 10 + t, rsDb, true); }
        } else {
            if constexpr (!(IE_EXP & 1024)) {
                if constexpr (IE_K67_CURVE2) { if (t >= 7 && sw < 3) curve_consume2(t - 2); }
                else { if (t >= 7) curve_consume(t - 2); }
            }
            if (t >= 6) stage_write(accN, (t - 1) & 1);
        }
        const char* ra_row = smem + OFF_RA + ((t + RPB) % NRH) * ROW_B;     // written by stage A at step t-RPB
        if constexpr (KIND == K12) {
            src_mfma(ie_tag<TH>{}, ra_row, 0, accN, accM, accO, true, bias4, NV2);     // x1 row ys-5-LAG_B+t
        } else if constexpr (KIND == K45) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) accN[nt] = bias4;
            src_mfma(ie_tag<TA0>{}, smem + ((t + NRA - 2) % NRA) * ROW_B, 0, accN, accM, accO, false, bias4, NV0);   // x3 row ys-4+t
            src_mfma(ie_tag<TH>{}, ra_row, 9, accN, accM, accO, false, bias4, NV0);                                   // x4 row ys-4+t
        } else if constexpr (KIND == K345 && !G::OUT) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) accN[nt] = bias4;
            src_mfma(ie_tag<T3>{}, smem + OFF_R3 + ((t + NR3 - 3) % NR3) * ROW_B, 0, accN, accM, accO, false, bias4, NV0);   // x3 row ys-8+t (emitted at step t-3)
            src_mfma(ie_tag<TH>{}, ra_row, 9, accN, accM, accO, false, bias4, NV0);                                           // x4 row ys-8+t (emitted at step t-1)
        } else if constexpr (KIND == K345) {
            // accN = the x5 row finished at step t-1: packed tile by tile ahead of the MFMAs that restart the set and parked in LDS as
            // whole pixel lines; the row parked at step t-1 leaves as 1 KiB of whole lines per store instruction.
            const int yo = ys - 11 + t;
            const unsigned rmask = ((yo >= ys) && (yo < ye)) ? 0u : OOB;
            const unsigned row_b = rmask ? 0u : (unsigned)yo * (unsigned)W * PXB;
            char* oslot = smem + OFF_OUT + (t & 1) * ROW_B;
            const char* islot = smem + OFF_OUT + ((t + 1) & 1) * ROW_B;
            ie_u32x2 pk[4];
            const ie_u32x4 l0 = *(const ie_u32x4*)(islot + st_lds[0]), l1 = *(const ie_u32x4*)(islot + st_lds[1]);
            pk[0] = IE_PACK_TILE(TO, accN[0], true);
            auto fill = [&](int s) __attribute__((always_inline)) {
                if (s == 0) { pk[1] = IE_PACK_TILE(TO, accN[1], true); *(ie_u32x2*)(oslot + em_off[0]) = pk[0]; }
                if (s == 1) *(ie_u32x2*)(oslot + em_off[1]) = pk[1];
                if (s == 3) { pk[2] = IE_PACK_TILE(TO, accN[2], true); *(ie_u32x2*)(oslot + em_off[2]) = pk[2]; }
                if (s == 6) { pk[3] = IE_PACK_TILE(TO, accN[3], true); *(ie_u32x2*)(oslot + em_off[3]) = pk[3]; }
                if (s == 10) __builtin_amdgcn_raw_buffer_store_b128(l0, rsDb, (int)((row_b + st_col[0]) | rmask), 0, 0);
                if (s == 12) __builtin_amdgcn_raw_buffer_store_b128(l1, rsDb, (int)((row_b + st_col[1]) | rmask), 0, 0);
            };
            auto nofill = [&](int) __attribute__((always_inline)) {};
            mfma_stream(ie_tag<T3>{}, smem + OFF_R3 + ((t + NR3 - 3) % NR3) * ROW_B, 0, accN, accM, accO, true, bias4, fill);   // x3 row ys-8+t (emitted at step t-3)
            mfma_stream(ie_tag<TH>{}, ra_row, 9, accN, accM, accO, false, bias4, nofill);                                       // x4 row ys-8+t (emitted at step t-1)
        } else {
            seed_from_p(t, accN);                                                      // P row ys-3+t = W7a*x1 + bias7
            src_mfma(ie_tag<TH>{}, ra_row, 0, accN, accM, accO, false, bias4, NV0);                    // x6 row ys-4+t
        }
        IE_STEP_TAIL();
    };


    // ---- K67N stage A, streamed: e_conv6 on [x2, x5] as two 36-MFMA streams.  accN still holds the x6 row finished at step t-1 (y = ys-4+t):
    // tile nt is packed (ReLU, 16-bit) just before the MFMA that restarts accumulator tile nt with the bias as its C operand, the two
    // 16-byte hand-off writes per lane, the four LDS-DMA pieces of row t+PF and nothing else ride behind fixed MFMAs.
    auto step_a67n = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
        IE_STEP_HEAD(wait_base, (PF - 1) * (2 * NSA + (IE_K67N_PA ? 2 : 0)), -1);
        IE_T(ts2);
        const int ra = ys - 4 + t;                             // the x6 row in accN
        const bool row_in = (ra >= 0) && (ra < H);
        char* eslot = smem + OFF_RA + (t % NRH) * ROW_B;       // read by stage B at step t+1
        const int yd = ys - A_ROW0 + t + PF;                   // the x2 / x5 row DMA'd now
        const unsigned dmask = ((yd >= 0) && (yd < H)) ? 0u : OOB;
        const unsigned drow = dmask ? 0u : (unsigned)yd * (unsigned)W * PXB;
        char* d0 = smem + ((t + PF) % NRA) * ROW_B;
        char* d1 = d0 + NRA * ROW_B;
        ie_u32x2 pk[4];
        pk[0] = IE_PACK_TILE(TH, accN[0], true);
        auto dma = [&](int src, int j) __attribute__((always_inline)) {
            const unsigned off = (drow + body_xoff[j]) | dmask;
            if (src == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA0, (__attribute__((address_space(3))) void*)(d0 + (2 * sw + j) * 1024), 16, (int)off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, (__attribute__((address_space(3))) void*)(d1 + (2 * sw + j) * 1024), 16, (int)off, 0, 0, 0);
        };
        auto fill = [&](int s) __attribute__((always_inline)) {
            if (s == 0) pk[1] = IE_PACK_TILE(TH, accN[1], true);
            if (s == 1) IE_PARK_PAIR(eslot, pk[0], pk[1], 0, row_in && (em16_in & 1u));
            if (s == 3) pk[2] = IE_PACK_TILE(TH, accN[2], true);
            if (s == 6) pk[3] = IE_PACK_TILE(TH, accN[3], true);
            if (s == 7) IE_PARK_PAIR(eslot, pk[2], pk[3], 1, row_in && ((em16_in >> 1) & 1u));
            if (s == 12) dma(0, 0);
            if (s == 18) dma(0, 1);
            if (s == 24) dma(1, 0);
            if (s == 30) dma(1, 1);
        };
        auto nofill = [&](int) __attribute__((always_inline)) {};
        mfma_stream(ie_tag<TA0>{}, smem + (t % NRA) * ROW_B, 0, accN, accM, accO, true, bias4, fill);
        mfma_stream(ie_tag<TA1>{}, smem + NRA * ROW_B + (t % NRA) * ROW_B, 9, accN, accM, accO, false, bias4, nofill);
        IE_STEP_TAIL();
    };

    // ---- K67N stage B: the x6 half of e_conv7 (seeded by P) + tanh + LE curve, all in registers -------------------------------
    // e_conv7's 24 output rows are permuted on the host (ie_zdce_plan.hip, chan_k67): row 4 kq + i of 16-row tile j is curve map
    // r_{4j+i} of colour kq (kq = 3: zero rows), and scaled by 2 / ln 2 (bias and the x1 half behind P too), so the accumulators hold
    // log2(e) * 2x: exp(2x) is one v_exp.  A wave owns both tiles on 32 columns: after the MFMAs lane (p16, kq) holds ALL EIGHT r_i of
    // colour kq of its pixel, in iteration order (tile 0 regs 0..3, tile 1 regs 0..3): tanh and the eight curve steps
    // (zero_dce.py:183-188) run straight off the accumulators -- no fp32 park in LDS, no second pass over it, no curve work on stage A.
    // acc[2 j + nt] = tile j, pixel tile nt.  The row's results are staged in LDS as fp32 [colour][pixel] and leave one step later as
    // 16-byte vectors (one 496-byte run per colour and strip row).  P is read in the same permuted order (K12 writes position
    // 8 kq + 4 j + i of a pixel's 64-byte line): one ds_read_b128 seeds both tiles.
    // The step is ONE MFMA stream (a wave's vector work is nearly free between its own MFMAs, two issue slots per 16x16x32 MFMA, and
    // costs its full length as a phase of its own; the SIMD's vector issue port -- 8 cycles per MFMA, 4 per vector, 8 per
    // transcendental instruction, both waves -- is what bounds this kernel, not the matrix pipe): the 96 tanh / curve operations of the
    // row finished at step t-1 ride as fillers pinned behind fixed MFMAs.  FOUR accumulator sets rotate: the finished row (accF) is
    // only read while the set freed a step earlier (accN) takes the seeds -- no copy of either.
    // Timeline: row y = ys-3+t is seeded at step t, finished at step t+2, its curve runs at step t+3 (accF), stored at t+4.
    auto step_b67n = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4], const ie_f32x4 (&accF)[4]) __attribute__((always_inline)) {
        IE_STEP_HEAD(wait_base, (PF - 1) * (IE_K67N_PA ? 2 : 4), (PF - 1) * 2);     // waves 4-6: two image pieces per row (+ two P pieces each unless stage A loads P); stores in between only make it stricter
        IE_T(ts2);
        typedef __attribute__((ext_vector_type(8))) TH th8;
        const char* ps = smem + OFF_BRING + (t % NRB) * ROW_B;
        const char* ims = smem + OFF_IMG + (t % NRI) * IMG_SLOT_B;
        const th8 sv0 = *(const th8*)(ps + sd_off[0]), sv1 = *(const th8*)(ps + sd_off[1]);     // seeds of the output row started now (P row ys-3+t)
        float v0 = *(const float*)(ims + im_off[0]), v1 = *(const float*)(ims + im_off[1]);     // fp32 pixel of the row finished at step t-1 (y = ys-6+t)
        IE_TX0();
        {   // (1) the row staged at step t-1 leaves
            const int yo = ys - 7 - G::DEFA + t;
            const bool row_ok = (yo >= ys) && (yo < ye);
            const char* os = smem + OFF_OUTF + ((t + 1) & 1) * G::OUTF_SLOT_B;
            const int xs = xc0 + MARG;                                  // first stored column of the strip (a multiple of 4)
            if constexpr (U8ONLY) {        // HWC bytes: 372 per strip row (W % 4 == 0: dword stores), wave 7 quantises them from the staged fp32 values
                if (sw == 3) {
                    const int nbytes = row_ok ? min(TSK, W - xs) * 3 : 0;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int d = lane + 64 * k;
                        unsigned pk = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int b = 4 * d + e, pxr = b / 3, c = b - 3 * pxr;
                            const float v = *(const float*)(os + c * G::OUTF_PLANE_B + ((pxr < TSK ? pxr : 0) + MARG + 2) * 4);
                            pk |= (unsigned)ie_quant_u8(v) << (8 * e);
                        }
                        const unsigned off = (4 * d < nbytes) ? (unsigned)(((size_t)yo * W + xs) * 3 + 4 * d) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(pk, rsU, (int)off, 0, 0);
                    }
                }
            } else if (FAST67 || (p.enhanced && (W & 3) == 0)) {
                const ie_u32x4 v = *(const ie_u32x4*)(os + est_lds);
                const unsigned rmask = row_ok ? 0u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(v, rsE, (int)((est_off + (rmask ? 0u : (unsigned)yo * (unsigned)W * 4u)) | rmask), 0, 0);
            } else if (p.enhanced) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int q = lane + 64 * k, x = xc0 + q;
                    const float v = *(const float*)(os + (sw < 3 ? sw : 0) * G::OUTF_PLANE_B + (q + 2) * 4);
                    const bool ok = row_ok && sw < 3 && q >= MARG && q < TW - MARG && x < W;
                    const unsigned off = ok ? (unsigned)(((size_t)sw * plane + (size_t)yo * W + x) * 4) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsE, (int)off, 0, 0);
                }
            }
            if constexpr (!NOU8) {
            if (p.u8 && sw == 3) {        // HWC bytes: 372 per strip row, wave 7 quantises them from the staged fp32 values
                const int nbytes = row_ok ? min(TSK, W - xs) * 3 : 0;
                if ((W & 3) == 0) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int d = lane + 64 * k;
                        unsigned pk = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int b = 4 * d + e, pxr = b / 3, c = b - 3 * pxr;
                            const float v = *(const float*)(os + c * G::OUTF_PLANE_B + ((pxr < TSK ? pxr : 0) + MARG + 2) * 4);
                            pk |= (unsigned)ie_quant_u8(v) << (8 * e);
                        }
                        const unsigned off = (4 * d < nbytes) ? (unsigned)(((size_t)yo * W + xs) * 3 + 4 * d) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(pk, rsU, (int)off, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int b = lane + 64 * k, pxr = b / 3, c = b - 3 * pxr;
                        const float v = *(const float*)(os + c * G::OUTF_PLANE_B + ((pxr < TSK ? pxr : 0) + MARG + 2) * 4);
                        const unsigned off = (b < nbytes) ? (unsigned)(((size_t)yo * W + xs) * 3 + b) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b8(ie_quant_u8(v), rsU, (int)off, 0, 0);
                    }
                }
            }
            }
        }
        IE_TX(0);
        issue(t + PF);       // (2)
        IE_TX(1);
        // (3) seeds into the free set
#pragma unroll
        for (int i = 0; i < 4; ++i) { accN[0][i] = (float)sv0[i]; accN[2][i] = (float)sv0[4 + i]; accN[1][i] = (float)sv1[i]; accN[3][i] = (float)sv1[4 + i]; }
        // (4) x6 row ys-4+t (handed over by stage A at step t-1) into the three live sets of both tiles, 36 MFMAs; filler o of 96:
        //     k = o / 12 (curve iteration), op = (o % 12) / 2, pixel tile = o % 2;  tanh(x) = 1 - 2 / (exp(2x) + 1) (ie_tanh_fast)
        float e0 = 0.f, e1 = 0.f, u0 = 0.f, u1 = 0.f;
        // the 24 curve maps (optional output, general variant only) leave straight from the lane that computed them: NCHW fp32, plane 3 i + colour
        unsigned aj_off[2] = {OOB, OOB};
        if constexpr (!LEAN67) {
            if (p.adjust) {
                const int yc = ys - 6 - G::DEFA + t;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int q = 32 * sw + 16 * nt + p16, x = xc0 + q;
                    const bool ok = (yc >= ys) && (yc < ye) && kq < 3 && q >= MARG && q < TW - MARG && x < W;
                    aj_off[nt] = ok ? (unsigned)(((size_t)kq * plane + (size_t)yc * W + x) * 4) : OOB;
                }
            }
        }
        auto op = [&](int o) __attribute__((always_inline)) {
            const int k = o / 12, w = (o % 12) >> 1, nt = o & 1;
            float& e = nt ? e1 : e0; float& u = nt ? u1 : u0; float& v = nt ? v1 : v0;
            if (w == 0) e = __builtin_amdgcn_exp2f(accF[2 * (k >> 2) + nt][k & 3]);
            if (w == 1) e = e + 1.0f;
            if (w == 2) e = __builtin_amdgcn_rcpf(e);
            if (w == 3) {
                e = __builtin_fmaf(e, -2.0f, 1.0f);       // r_k
                if constexpr (!LEAN67) {                  // (an absent `adjust` has an empty descriptor range and out-of-range offsets: the store is dropped)
                    const unsigned off = aj_off[nt] == OOB ? OOB : aj_off[nt] + (unsigned)((size_t)(3 * k) * plane * 4);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e), rsJ, (int)off, 0, 0);
                }
            }
            if (w == 4) u = __builtin_fmaf(v, v, -v);
            if (w == 5) v = __builtin_fmaf(e, u, v);
        };
        const char* ra_row = smem + OFF_RA + ((t + RPB) % NRH) * ROW_B;
        frag bx[6];
        auto rd = [&](int g) __attribute__((always_inline)) { bx[g] = *(const frag*)(ra_row + rdn[g & 1][g >> 1]); };
        rd(0); rd(1); rd(2);
        IE_SB();
        IE_TX(2);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int kx = g >> 1, nt = g & 1;
            if (g + 3 < 6) rd(g + 3);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int s0 = 6 * g + 3 * j;                  // MFMA index of this triplet's first
                accN[2 * j + nt] = ie_mfma16_raw<TH>(wv[9 * j + 0 * 3 + kx], bx[g], accN[2 * j + nt]);
#pragma unroll
                for (int o = (s0 * 8) / 3; o < ((s0 + 1) * 8) / 3; ++o) op(o);
                IE_SB();
                accM[2 * j + nt] = ie_mfma16_raw<TH>(wv[9 * j + 1 * 3 + kx], bx[g], accM[2 * j + nt]);
#pragma unroll
                for (int o = ((s0 + 1) * 8) / 3; o < ((s0 + 2) * 8) / 3; ++o) op(o);
                IE_SB();
                accO[2 * j + nt] = ie_mfma16_raw<TH>(wv[9 * j + 2 * 3 + kx], bx[g], accO[2 * j + nt]);
#pragma unroll
                for (int o = ((s0 + 2) * 8) / 3; o < ((s0 + 3) * 8) / 3; ++o) op(o);
                IE_SB();
            }
        }
        IE_TX(3);
        {   // (5) the curve's results into the fp32 output row; the 24 curve maps (optional) straight to HBM: NCHW fp32, plane 3 i + colour
            char* ow = smem + OFF_OUTF + (t & 1) * G::OUTF_SLOT_B;
            *(float*)(ow + of_off[0]) = v0;
            *(float*)(ow + of_off[1]) = v1;
        }
        IE_STEP_TAIL();
    };

    // Stage B: e_conv2.  accN = the x2 row finished at step t-1.  Fillers: its pack + store, and the conversion of image row R(t).
    auto step_b12 = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
        IE_STEP_HEAD(wait_base, -1, -1);      // (stage B of K12 issues no loads: wait_base < 0)
        IE_T(ts2);
        const int yo = ys - 6 - LAG_A - LAG_H - G::OUT + t;    // the row parked at step t-1 (finished at step t-2)
        const unsigned rmask = ((yo >= ys) && (yo < ye) && !((IE_EXP & 128) && p.N > 0)) ? 0u : OOB;   // IE_EXP 128: every store out of range (same instructions, no HBM writes)
        const unsigned row_b = rmask ? 0u : (unsigned)yo * (unsigned)W * PXB;
        ie_u32x2 pk[4];
        ie_u32x4 ln[2] = {};
        char* oslot = smem + OFF_OUT + (t & 1) * ROW_B;                   // x2 rows: slots 0, 1
        const char* islot = smem + OFF_OUT + ((t + 1) & 1) * ROW_B;
        const int px = sw * 64 + lane;
        const float* ip = (const float*)(smem + OFF_IMG + (t % NRI) * IMG_SLOT_B) + (px < 131 ? px : 131);
        const bool keep = px < 132;
        char* cv = smem + (keep ? OFF_CV + (t % NRC) * CV_SLOT_B + px * 8 : OFF_DUMP + lane * 8);
        float r0 = 0.f, g0 = 0.f, b0 = 0.f;
        ie_u32x2 hi = {0u, 0u}, lo = {0u, 0u};
        pk[0] = IE_PACK_TILE(TO, accN[0], true);
        ln[0] = *(const ie_u32x4*)(islot + st_lds[0]);
        ln[1] = *(const ie_u32x4*)(islot + st_lds[1]);
        auto fill = [&](int s) __attribute__((always_inline)) {
            if (s == 0) pk[1] = IE_PACK_TILE(TO, accN[1], true);
            if (s == 1) IE_PARK_PAIR(oslot, pk[0], pk[1], 0, true);
            if (s == 3) pk[2] = IE_PACK_TILE(TO, accN[2], true);
            if (s == 6) pk[3] = IE_PACK_TILE(TO, accN[3], true);
            if (s == 7) IE_PARK_PAIR(oslot, pk[2], pk[3], 1, true);
            if constexpr (!(IE_EXP & 16)) {
                if (s == 10) __builtin_amdgcn_raw_buffer_store_b128(ln[0], rsDb, (int)((row_b + st_col[0]) | rmask), 0, 0);
                if (s == 12) __builtin_amdgcn_raw_buffer_store_b128(ln[1], rsDb, (int)((row_b + st_col[1]) | rmask), 0, 0);
            }
            if constexpr (!(IE_EXP & 1)) {
                if (s == 9) { r0 = ip[0]; g0 = ip[132]; b0 = ip[264]; }
                if (s == 20) { const bool real = px < 130; r0 = real ? r0 : 0.f; g0 = real ? g0 : 0.f; b0 = real ? b0 : 0.f; }
                if (s == 21) { hi[0] = ie_pack2<TH>(r0, g0); hi[1] = ie_pack2<TH>(b0, 0.f); }
                if (s == 22) { r0 -= (float)(TH)r0; g0 -= (float)(TH)g0; }
                if (s == 23) { b0 -= (float)(TH)b0; lo[0] = ie_pack2<TH>(r0, g0); }
                if (s == 24) { lo[1] = ie_pack2<TH>(b0, 0.f); *(ie_u32x2*)cv = hi; }
                if (s == 25) *(ie_u32x2*)(cv + (keep ? CV_PLANE_B : 512)) = lo;
            }
        };
        mfma_stream(ie_tag<TH>{}, smem + OFF_RA + ((t + RPB) % NRH) * ROW_B, 0, accN, accM, accO, true, bias4, fill);
        IE_STEP_TAIL();
    };
    // Stage A: the x1 half of e_conv7 (36 MFMAs) then e_conv1 (8, or 16 with bf16 operands).  eN = the x1 row e_conv1 finished at
    // step t-1, pN = the P row finished at step t-1.  Fillers: P pack + store, the hand-off write of eN, the image DMA of row t+PF,
    // the LDS reads of e_conv1's operands.
    auto step_a12 = [&](int t, ie_f32x4 (&eN)[4], ie_f32x4 (&eO)[4], ie_f32x4 (&pN)[4], ie_f32x4 (&pM)[4], ie_f32x4 (&pO)[4]) __attribute__((always_inline)) {
        // at the barrier the image DMA of row t+RPB-1 must have landed.  Per step this wave issues two P stores, then two loads:
        // ops younger than that DMA = 2 (PF-RPB) prologue loads + 2 per elapsed step, 4 (PF-RPB) in the steady state
        IE_STEP_HEAD(2 * (PF - RPB) + 2 * (t < PF - RPB ? t : PF - RPB), 4 * (PF - RPB), -1);
        IE_T(ts2);
        const int yo = ys - 6 - LAG_A - LAG_H - G::OUT + t;    // the P row parked at step t-1 (finished at step t-2)
        const unsigned rmask = ((yo >= ys) && (yo < ye) && !((IE_EXP & 128) && p.N > 0)) ? 0u : OOB;   // IE_EXP 128: every store out of range (same instructions, no HBM writes)
        const unsigned row_b = rmask ? 0u : (unsigned)yo * (unsigned)W * PXB;
        ie_u32x4 ln[2] = {};
        char* oslot = smem + OFF_OUT + (2 + (t & 1)) * ROW_B;             // P rows: slots 2, 3
        const char* islot = smem + OFF_OUT + (2 + ((t + 1) & 1)) * ROW_B;
        const int ra = ys - 3 - LAG_A - 1 + t;                      // x1 row computed at step t-1
        const bool row_in = (ra >= 0) && (ra < H);
        char* eslot = smem + OFF_RA + (t % NRH) * ROW_B;
        ie_u32x2 pk[4], ek = {0u, 0u}, ek1 = {0u, 0u};
        constexpr int PARTS = (ie_elem<TH>::dtype == IE_BF16) ? 2 : 1;
        ie_u32x4 c0[PARTS][4], c1[PARTS][4];
        const int pxo = (64 * ph + p16 + 2 * (kq & 1)) * 8;
        const int s0 = OFF_CV + ((t + NRC - 2 - LAG_A + (kq >> 1)) % NRC) * CV_SLOT_B + pxo;     // m = 0: ky = kq>>1
        const int s1 = OFF_CV + ((t + NRC - LAG_A) % NRC) * CV_SLOT_B + pxo;                     // m = 1: ky = 2 (kq < 2)
        auto c1_read = [&](int part, int nt) __attribute__((always_inline)) {
            if constexpr (IE_EXP & 2) { c0[part][nt] = wv[0]; c1[part][nt] = wv[1]; return; }
            const char* r0 = smem + s0 + part * CV_PLANE_B + 128 * nt;
            const char* r1 = smem + s1 + part * CV_PLANE_B + 128 * nt;
            const ie_u32x2 a0_ = *(const ie_u32x2*)r0, a1_ = *(const ie_u32x2*)(r0 + 8);
            const ie_u32x2 b0_ = *(const ie_u32x2*)r1, b1_ = *(const ie_u32x2*)(r1 + 8);
            c0[part][nt] = ie_u32x4{a0_[0], a0_[1], a1_[0], a1_[1]};
            c1[part][nt] = ie_u32x4{b0_[0], b0_[1], b1_[0], b1_[1]};
        };
        pk[0] = IE_PACK_TILE(TO, pN[0], false);
        ln[0] = *(const ie_u32x4*)(islot + st_lds[0]);
        ln[1] = *(const ie_u32x4*)(islot + st_lds[1]);
        auto fill = [&](int s) __attribute__((always_inline)) {
            if (s == 0) pk[1] = IE_PACK_TILE(TO, pN[1], false);
            if (s == 1) IE_PARK_PAIR(oslot, pk[0], pk[1], 0, true);
            if (s == 3) pk[2] = IE_PACK_TILE(TO, pN[2], false);
            if (s == 5) pk[3] = IE_PACK_TILE(TO, pN[3], false);
            if (s == 6) IE_PARK_PAIR(oslot, pk[2], pk[3], 1, true);
            if constexpr (!(IE_EXP & 16)) {
                if (s == 7) __builtin_amdgcn_raw_buffer_store_b128(ln[0], rsDa, (int)((row_b + st_col[0]) | rmask), 0, 0);
                if (s == 8) __builtin_amdgcn_raw_buffer_store_b128(ln[1], rsDa, (int)((row_b + st_col[1]) | rmask), 0, 0);
            }
            if constexpr (!(IE_EXP & 4)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (s == 9 + 5 * j) ek = IE_PACK_TILE(TH, eN[2 * j], true);
                    if (s == 10 + 5 * j) ek1 = IE_PACK_TILE(TH, eN[2 * j + 1], true);
                    if (s == 11 + 5 * j) IE_PARK_PAIR(eslot, ek, ek1, j, row_in && ((em16_in >> j) & 1u));
                }
            }
            if (s == 21) issue(t + PF);
            if constexpr (!(IE_EXP & 32)) {
#pragma unroll
                for (int part = 0; part < PARTS; ++part)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        if (s == 24 + 4 * part + nt) c1_read(part, nt);
            }
        };
        mfma_stream(ie_tag<TH>{}, smem + OFF_RA + ((t + RPB) % NRH) * ROW_B, 9, pN, pM, pO, true, bias4p, fill);
        if constexpr (!(IE_EXP & 32)) {
#pragma unroll
            for (int part = 0; part < PARTS; ++part) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) eO[nt] = ie_mfma16_raw<TH>(wv[0], c0[part][nt], part == 0 ? bias4 : eO[nt]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) eO[nt] = ie_mfma16_raw<TH>(wv[1], c1[part][nt], eO[nt]);
            }
        }
        IE_STEP_TAIL();
    };

#pragma unroll
    for (int t = 0; t < PF; ++t) issue(t);

    ie_f32x4 a0[4], a1[4], a2[4];
    ie_f32x4 q0[4], q1[4], q2[4];     // second accumulator family: K12 stage B (the e_conv7a / P map), K345 stage A (e_conv4)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { a0[nt] = bias4; a1[nt] = bias4; a2[nt] = bias4; q0[nt] = bias4p; q1[nt] = bias4p; q2[nt] = bias4p; }
    if constexpr (KIND == K12 && G::DEF) {
        if (stg == 0) {
            for (int t = 0; t < n_steps; t += 3) {
                step_a12(t, a0, a2, q0, q1, q2);
                if (t + 1 < n_steps) step_a12(t + 1, a2, a1, q2, q0, q1);
                if (t + 2 < n_steps) step_a12(t + 2, a1, a0, q1, q2, q0);
            }
        } else {
            for (int t = 0; t < n_steps; t += 3) {
                step_b12(t, a0, a1, a2);
                if (t + 1 < n_steps) step_b12(t + 1, a2, a0, a1);
                if (t + 2 < n_steps) step_b12(t + 2, a1, a2, a0);
            }
        }
    } else if (IS67N && stg == 1) {
        if constexpr (IS67N) {
#if IE_K67N_PRIO
            __builtin_amdgcn_s_setprio(IE_K67N_PRIO);    // stage B (the younger half of the workgroup) is the longer chain of a step
#endif
            for (int t = 0; t < n_steps; t += 4) {      // (N, M, O, F) -> (F, N, M, O)
                step_b67n(t, a0, a1, a2, q0);
                if (t + 1 < n_steps) step_b67n(t + 1, q0, a0, a1, a2);
                if (t + 2 < n_steps) step_b67n(t + 2, a2, q0, a0, a1);
                if (t + 3 < n_steps) step_b67n(t + 3, a1, a2, q0, a0);
            }
        }
    } else if (IS67N && IE_K67N_STREAM_A && stg == 0) {
        if constexpr (IS67N) {
            for (int t = 0; t < n_steps; t += 3) {
                step_a67n(t, a0, a1, a2);
                if (t + 1 < n_steps) step_a67n(t + 1, a2, a0, a1);
                if (t + 2 < n_steps) step_a67n(t + 2, a1, a2, a0);
            }
        }
    } else if (stg == 0) {
        for (int t = 0; t < n_steps; t += 3) {
            step_a(t, a0, a1, a2, q0, q1, q2);
            if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1, q2, q0, q1);
            if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0, q1, q2, q0);
        }
    } else if constexpr (!IS67N) {
        for (int t = 0; t < n_steps; t += 3) {
            step_b(t, a0, a1, a2, q0, q1, q2);
            if (t + 1 < n_steps) step_b(t + 1, a2, a0, a1, q2, q0, q1);
            if (t + 2 < n_steps) step_b(t + 2, a1, a2, a0, q1, q2, q0);
        }
    }
#undef IE_STEP_HEAD
#undef IE_STEP_TAIL
#undef IE_B_STORE
#undef IE_PACK_TILE
#undef IE_STORE_PAIR
#undef IE_PARK_PAIR
#undef IE_SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef IE_STAMP
    if (p.stamps && lane == 0) {
        unsigned long long* d = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        d[0] = st_wait; d[1] = st_issue; d[2] = st_work; d[3] = n_steps;
        d[4] = st_e[0]; d[5] = st_e[1]; d[6] = st_e[2]; d[7] = st_e[3];
    }
#endif
}

// Concrete kernels (readable names in rocprof traces; hipcc/ROCm 7.2 emitted the host stub of only
// the first instantiation when the __global__ function itself was a template on KIND).
#define IE_DEF_FUSED(NAME, TA0, TA1, TH, TO, KIND) \
    __global__ __launch_bounds__(512, 2) void NAME(FusedParams p) { zdce_fused2_body<TA0, TA1, TH, TO, KIND>(p); }
IE_DEF_FUSED(zdce_k12_conv1_conv2_f16, _Float16, _Float16, _Float16, _Float16, K12)
IE_DEF_FUSED(zdce_k45_conv4_conv5_f16, _Float16, _Float16, _Float16, _Float16, K45)
IE_DEF_FUSED(zdce_k67_conv6_conv7_curve_f16, _Float16, _Float16, _Float16, _Float16, K67)
IE_DEF_FUSED(zdce_k12_conv1_conv2_bf16, __bf16, __bf16, __bf16, __bf16, K12)
IE_DEF_FUSED(zdce_k45_conv4_conv5_bf16, __bf16, __bf16, __bf16, __bf16, K45)
IE_DEF_FUSED(zdce_k67_conv6_conv7_curve_bf16, __bf16, __bf16, __bf16, __bf16, K67)
// the "bf16" schedule's last kernel: x2 (fp16) and x5 (bf16) in, x6 / P / e_conv7b in fp16
IE_DEF_FUSED(zdce_k67_conv6_conv7_curve_mix, _Float16, __bf16, _Float16, _Float16, K67)
// K67N: the shipped form (curve in stage B's registers; rocprof names zdce_k67n_*)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_f16, _Float16, _Float16, _Float16, _Float16, K67N)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_bf16, __bf16, __bf16, __bf16, __bf16, K67N)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_mix, _Float16, __bf16, _Float16, _Float16, K67N)
// ... compiled for `enhanced` fp32 as the only output and W % 4 == 0 (what the throughput path asks for)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_e_f16, _Float16, _Float16, _Float16, _Float16, K67F)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_e_bf16, __bf16, __bf16, __bf16, __bf16, K67F)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_e_mix, _Float16, __bf16, _Float16, _Float16, K67F)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_ea_f16, _Float16, _Float16, _Float16, _Float16, K67A)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_ea_bf16, __bf16, __bf16, __bf16, __bf16, K67A)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_ea_mix, _Float16, __bf16, _Float16, _Float16, K67A)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_u8_f16, _Float16, _Float16, _Float16, _Float16, K67U)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_u8_bf16, __bf16, __bf16, __bf16, __bf16, K67U)
IE_DEF_FUSED(zdce_k67n_conv6_conv7_curve_u8_mix, _Float16, __bf16, _Float16, _Float16, K67U)
#undef IE_DEF_FUSED
#define IE_DEF_FUSED3(NAME, TA0, T3, TH, TO) \
    __global__ __launch_bounds__(512, 2) void NAME(FusedParams p) { zdce_fused2_body<TA0, TA0, TH, TO, K345, T3>(p); }
IE_DEF_FUSED3(zdce_k345_conv3_conv4_conv5_f16, _Float16, _Float16, _Float16, _Float16)
IE_DEF_FUSED3(zdce_k345_conv3_conv4_conv5_bf16, __bf16, __bf16, __bf16, __bf16)
IE_DEF_FUSED3(zdce_k345_conv3_conv4_conv5_mix, _Float16, __bf16, __bf16, __bf16)      // x2 fp16 in; x3, x4, x5 bf16
#undef IE_DEF_FUSED3

typedef void (*FusedFn)(FusedParams);

template <int KIND>
static int launch_fused_t(FusedFn fn, const ZdceFusedLaunch& a, hipStream_t st) {
    FusedParams p;
    p.a_src0 = a.a_src0; p.a_src1 = a.a_src1; p.b_src = a.b_src; p.img = a.img;
    p.a_w = a.a_w; p.a_bias = a.a_bias; p.a2_w = a.a2_w; p.a2_bias = a.a2_bias; p.b_w = a.b_w; p.b_bias = a.b_bias; p.b2_bias = a.b2_bias; p.a_dst = a.a_dst; p.b_dst = a.b_dst;
    p.enhanced = a.enhanced; p.adjust = a.adjust; p.u8 = a.u8;
    p.N = a.N; p.H = a.H; p.W = a.W; p.y_begin = a.y_begin; p.y_end = a.y_end; p.rows_per_block = a.rows_per_block;
    p.n_begin = a.n_begin;
    p.stamps = (unsigned long long*)a.stamps;
    p.strips = ie_cdiv(a.W, KIND == K345 ? TW - 6 : ((KIND == K67 || KIND == K67N || KIND == K67F || KIND == K67U || KIND == K67A) ? TW - 4 : TS));
    if (KIND == K67N && a.adjust && (size_t)a.H * a.W * 24 * 4 >= 0x7fffffffull) IE_FAIL(IE_ERR_UNSUPPORTED, "zdce fused: frame too large for 31-bit offsets into `adjust`");
    p.chunks = ie_cdiv(a.y_end - a.y_begin, a.rows_per_block);
    const long long blocks = (long long)p.strips * p.chunks * a.n_count;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "zdce fused: bad grid %lld", blocks);
    if ((size_t)a.H * a.W * 32 * 2 >= 0x7fffffffull) IE_FAIL(IE_ERR_UNSUPPORTED, "zdce fused: frame too large for 31-bit offsets");
    const size_t lds = FusedGeom<KIND>::LDS_B;
    IE_GRANT_LDS(fn, lds);       // a per-(device, function) attribute
    void* args[] = {(void*)&p};
    IE_HIP_CHECK(hipLaunchKernel((const void*)fn, dim3((unsigned)blocks), dim3(512), args, lds, st));
    return IE_OK;
}

int ie_zdce_launch_fused(const ZdceFusedLaunch& a, hipStream_t st) {
    const bool h_all = a.dt_h == IE_F16 && a.dt_o == IE_F16 && a.dt_a0 == IE_F16 && a.dt_a1 == IE_F16;
    const bool b_all = a.dt_h == IE_BF16 && a.dt_o == IE_BF16 && a.dt_a0 == IE_BF16 && a.dt_a1 == IE_BF16;
    if (a.kind == 12) {
        if (h_all) return launch_fused_t<K12>(zdce_k12_conv1_conv2_f16, a, st);
        if (b_all) return launch_fused_t<K12>(zdce_k12_conv1_conv2_bf16, a, st);
    } else if (a.kind == 45) {
        if (h_all) return launch_fused_t<K45>(zdce_k45_conv4_conv5_f16, a, st);
        if (b_all) return launch_fused_t<K45>(zdce_k45_conv4_conv5_bf16, a, st);
    } else if (a.kind == 67) {
        if (h_all) return launch_fused_t<K67>(zdce_k67_conv6_conv7_curve_f16, a, st);
        if (b_all) return launch_fused_t<K67>(zdce_k67_conv6_conv7_curve_bf16, a, st);
        if (a.dt_a0 == IE_F16 && a.dt_a1 == IE_BF16 && a.dt_h == IE_F16) return launch_fused_t<K67>(zdce_k67_conv6_conv7_curve_mix, a, st);
    } else if (a.kind == 670 && a.enhanced && !a.adjust && !a.u8 && (a.W & 3) == 0 && ((size_t)a.enhanced & 15) == 0) {
        if (h_all) return launch_fused_t<K67F>(zdce_k67n_conv6_conv7_curve_e_f16, a, st);
        if (b_all) return launch_fused_t<K67F>(zdce_k67n_conv6_conv7_curve_e_bf16, a, st);
        if (a.dt_a0 == IE_F16 && a.dt_a1 == IE_BF16 && a.dt_h == IE_F16) return launch_fused_t<K67F>(zdce_k67n_conv6_conv7_curve_e_mix, a, st);
    } else if (a.kind == 670 && a.enhanced && a.adjust && !a.u8 && (a.W & 3) == 0 && ((size_t)a.enhanced & 15) == 0) {       // model.forward(): enhanced + curve maps
        if (h_all) return launch_fused_t<K67A>(zdce_k67n_conv6_conv7_curve_ea_f16, a, st);
        if (b_all) return launch_fused_t<K67A>(zdce_k67n_conv6_conv7_curve_ea_bf16, a, st);
        if (a.dt_a0 == IE_F16 && a.dt_a1 == IE_BF16 && a.dt_h == IE_F16) return launch_fused_t<K67A>(zdce_k67n_conv6_conv7_curve_ea_mix, a, st);
    } else if (a.kind == 670 && a.u8 && !a.enhanced && !a.adjust && (a.W & 3) == 0) {       // the uint8 pipeline: the HWC bytes alone
        if (h_all) return launch_fused_t<K67U>(zdce_k67n_conv6_conv7_curve_u8_f16, a, st);
        if (b_all) return launch_fused_t<K67U>(zdce_k67n_conv6_conv7_curve_u8_bf16, a, st);
        if (a.dt_a0 == IE_F16 && a.dt_a1 == IE_BF16 && a.dt_h == IE_F16) return launch_fused_t<K67U>(zdce_k67n_conv6_conv7_curve_u8_mix, a, st);
    } else if (a.kind == 670) {
        if (h_all) return launch_fused_t<K67N>(zdce_k67n_conv6_conv7_curve_f16, a, st);
        if (b_all) return launch_fused_t<K67N>(zdce_k67n_conv6_conv7_curve_bf16, a, st);
        if (a.dt_a0 == IE_F16 && a.dt_a1 == IE_BF16 && a.dt_h == IE_F16) return launch_fused_t<K67N>(zdce_k67n_conv6_conv7_curve_mix, a, st);
    } else if (a.kind == 345) {      // dt_a0 = x2, dt_a1 = x3 (LDS), dt_h = x4 (LDS), dt_o = x5
        if (h_all) return launch_fused_t<K345>(zdce_k345_conv3_conv4_conv5_f16, a, st);
        if (b_all) return launch_fused_t<K345>(zdce_k345_conv3_conv4_conv5_bf16, a, st);
        if (a.dt_a0 == IE_F16 && a.dt_a1 == IE_BF16 && a.dt_h == IE_BF16 && a.dt_o == IE_BF16)
            return launch_fused_t<K345>(zdce_k345_conv3_conv4_conv5_mix, a, st);
    }
    IE_FAIL(IE_ERR_UNSUPPORTED, "zdce fused: kind=%d dtypes a0=%d a1=%d h=%d o=%d", a.kind, a.dt_a0, a.dt_a1, a.dt_h, a.dt_o);
}
