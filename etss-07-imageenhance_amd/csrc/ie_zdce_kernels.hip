// Zero-DCE hot-path kernels for gfx950 (MI355X).
//
//   zdce_conv1_kernel      e_conv1: NCHW fp32 image (3 ch) -> NHWC T (32 ch), + bias + ReLU
//   zdce_conv3x3_mfma      e_conv2..7: NHWC T (32 ch, 1 or 2 sources = virtual concat)
//                          -> NHWC T (ReLU)  or  tanh + LE curve -> enhanced/adjust/u8
//
// Reference semantics: src/mon/vision/enhance/llie/zero_dce/zero_dce.py:173-191.
//
// Design of zdce_conv3x3_mfma (one workgroup = 4 waves = a 128-column strip of the image):
//   * implicit GEMM on v_mfma_f32_16x16x32_{f16,bf16}:
//         D[cout 16][pixel 16] += Wt[cout 16][k 32] * X[k 32][pixel 16]
//     one MFMA contracts all 32 channels of one source for one filter tap.  The weights are
//     the A operand and stay in registers for the whole kernel (9 taps x NSRC fragments per
//     wave); a wave owns 16 output channels x 64 columns (wave = cout-half * 2 + column-half).
//   * line buffer: the workgroup marches down RB output rows.  Every input row is DMA'd into
//     an LDS ring slot exactly once (buffer_load ... lds, 16 B per lane; the hardware range
//     check of the buffer descriptor supplies the zero padding) and each ds_read_b128 pixel
//     fragment feeds THREE MFMAs (ky = 0,1,2) into three rotating accumulator sets, so LDS
//     read traffic is one third of the MFMA B-operand traffic.
//   * LDS image of a ring slot: [130 pixels][32 ch] T; pixels 128/129 hold the left/right halo
//     columns.  The four 16-byte channel chunks of pixel q are XOR-swizzled by (q>>1)&3, which
//     makes the 16x16x32 B-operand read (16 consecutive pixels x 4 chunks per wave) bank
//     conflict free at every tap offset.  The DMA writes LDS linearly (base + lane*16), so the
//     swizzle is applied to the per-lane SOURCE address.
//   * ReLU epilogue: D has the pixel on the lane and 4 consecutive couts in registers; lanes
//     l / l+16 exchange halves (v_permlane16_swap) so each lane stores 16 contiguous bytes.
//   * curve epilogue (e_conv7): tanh(D) is staged through LDS as [24][128] fp32, then one
//     thread per pixel runs the 8-step LE curve on the fp32 input pixel (DMA'd into a small
//     LDS ring as well) and writes enhanced / adjust / uint8 with fully coalesced stores.
#include "ie_zdce_internal.h"

namespace {

constexpr int TW = 128;          // output columns per workgroup
constexpr int NWAVE = 4;         // waves per workgroup
constexpr int ROW_PX = TW + 2;   // pixels per ring row (body + 2 halo)
constexpr unsigned OOB = 0x80000000u;

enum { EPI_RELU = 0, EPI_CURVE = 1 };

struct ConvParams {
    const void* src0;   // NHWC T, 32 ch
    const void* src1;   // NHWC T, 32 ch (NSRC == 2)
    const void* wpk;    // packed weights [2 cout halves][NSRC*9][64 lanes][8] T
    const float* bias;  // [32] fp32 (zero padded)
    void* dst;          // EPI_RELU: NHWC T 32 ch
    const float* img;   // EPI_CURVE: NCHW fp32 (N,3,H,W)
    float* enhanced;    // EPI_CURVE: NCHW fp32 or null
    float* adjust;      // EPI_CURVE: NCHW fp32 (N,24,H,W) or null
    unsigned char* u8;  // EPI_CURVE: NHWC u8 (N,H,W,3) or null
    int N, H, W;
    int y_begin, y_end; // output rows computed by this launch: [y_begin, y_end)
    int rows_per_block; // RB
    int strips, chunks; // grid decomposition
    int n_begin;        // first frame of this launch
};

__device__ __forceinline__ void wait_vmcnt(int n) {
    // s_waitcnt takes an immediate; n is wave-uniform
    switch (n) {
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// T0 / T1: element types of source 0 / 1 (and of their weight fragments), TO: element type of the ReLU output.
// All three are 16-bit; they differ only in the "bf16" schedule of the plan, which holds x1, x2, x6 and P in fp16.
template <typename T0, typename T1, typename TO, int NSRC, int EPI, bool DMA, bool SYNC, bool WIDE>
__global__ __launch_bounds__(NWAVE * 64, (NSRC == 1 ? 3 : 2)) void zdce_conv3x3_mfma(ConvParams p) {
    using frag = ie_u32x4;
    constexpr int NR = (EPI == EPI_CURVE) ? 3 : 4;      // ring slots (input rows resident in LDS)
    constexpr int PF = NR - 1;                          // rows of DMA prefetch issued ahead
    constexpr int PXB = 32 * 2;                         // bytes per pixel per source (64)
    constexpr int SRC_ROW_B = ROW_PX * PXB;             // bytes per ring row per source
    constexpr int SLOT_B = SRC_ROW_B * NSRC;            // bytes per ring slot
    constexpr int RING_B = SLOT_B * NR;
    constexpr int STAGE_B = 24 * TW * 4;                // EPI_CURVE: tanh(r) staging [24][128] fp32
    constexpr int IMG_SLOT_B = 3 * TW * 4;              // EPI_CURVE: image row slice [3][128] fp32
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ie_fp16_saturate_on();
    char* const stage = smem + RING_B;
    char* const imgring = smem + RING_B + STAGE_B;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int ph = wave & 1;    // column half: columns [64ph, 64ph+64) of the strip
    const int ch = wave >> 1;   // cout half:   couts   [16ch, 16ch+16)

    // ---- block -> (frame, row chunk, column strip) ---------------------------
    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = p.n_begin + b;
    const int x0 = strip * TW;
    const int ys = p.y_begin + chunk * p.rows_per_block;
    const int ye = min(ys + p.rows_per_block, p.y_end);
    const int n_it = (ye - ys) + 2;                     // input rows ys-1 .. ye
    const int H = p.H, W = p.W;

    // ---- buffer descriptors (wave-uniform), one frame each -------------------
    const size_t frame_b = (size_t)H * W * PXB;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src0 + (size_t)n * frame_b), 0,
                                                       (int)frame_b, 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(NSRC == 2 ? p.src1 : p.src0) + (size_t)n * frame_b), 0, (int)frame_b, 0x00020000);
    const size_t plane = (size_t)H * W;
    const auto rsi = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(EPI == EPI_CURVE ? (const char*)(p.img + (size_t)n * 3 * plane) : (const char*)p.src0), 0,
        (int)(3 * plane * 4), 0x00020000);

    // ---- weights (A operand) into registers ----------------------------------
    // fragment f = s*9 + ky*3 + kx holds W[16ch + (lane&15)][s*32 + 8*(lane>>4) + j][ky][kx], j < 8
    frag wA[NSRC * 9];
    {
        const frag* wp = (const frag*)p.wpk + (size_t)ch * (NSRC * 9) * 64;
#pragma unroll
        for (int f = 0; f < NSRC * 9; ++f) wA[f] = wp[f * 64 + lane];
    }
    // accumulator rows of this lane: cout = 16ch + 4kq + i
    ie_f32x4 bias4;
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = p.bias[16 * ch + 4 * kq + i];

    // ---- per-lane DMA source geometry ----------------------------------------
    // body instruction j (0,1) of this wave covers ring pixels q = (2*wave+j)*16 + (lane>>2);
    // 16-byte position (lane&3) of pixel q holds channel chunk (lane&3) ^ ((q>>1)&3).
    int body_xoff[2];   // byte offset inside an image row, or -1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (2 * wave + j) * 16 + (lane >> 2);
        const int s = (lane & 3) ^ ((q >> 1) & 3);
        const int x = x0 + q;
        body_xoff[j] = (x < W) ? (x * PXB + s * 16) : -1;
    }
    // halo (wave 0): lanes 0-3 -> ring pixel 128 (x0-1), lanes 4-7 -> ring pixel 129 (x0+128);
    // (128>>1)&3 == (129>>1)&3 == 0, so position == chunk.
    int halo_xoff;
    {
        const int x = (lane < 4) ? (x0 - 1) : (x0 + TW);
        halo_xoff = (lane < 8 && x >= 0 && x < W) ? (x * PXB + (lane & 3) * 16) : -1;
    }
    // image row slice (EPI_CURVE): wave c < 3 loads plane c, two instructions of 64 floats
    int img_xoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = x0 + 64 * j + lane;
        img_xoff[j] = (x < W) ? x * 4 : -1;
    }

    auto issue_row = [&](int it) __attribute__((always_inline)) {
        // input row yi of iteration `it` into ring slot it % NR
        const int yi = ys - 1 + it;
        const bool row_ok = (yi >= 0) && (yi < H) && (it < n_it);
        const unsigned row_b = (unsigned)yi * (unsigned)W * PXB;
        char* slot = smem + (it % NR) * SLOT_B;
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const auto rs = s ? rs1 : rs0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned off = (row_ok && body_xoff[j] >= 0) ? row_b + (unsigned)body_xoff[j] : OOB;
                char* dstb = slot + s * SRC_ROW_B + (2 * wave + j) * 1024;
                if constexpr (DMA) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dstb, 16,
                                                             (int)off, 0, 0, 0);
                } else {
                    ie_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
                    *(ie_u32x4*)(dstb + lane * 16) = v;
                }
            }
        }
        if (wave == 0) {
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const auto rs = s ? rs1 : rs0;
                const unsigned off = (row_ok && halo_xoff >= 0) ? row_b + (unsigned)halo_xoff : OOB;
                char* dstb = slot + s * SRC_ROW_B + TW * PXB;
                if (lane < 8) {
                    if constexpr (DMA) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dstb, 16,
                                                                 (int)off, 0, 0, 0);
                    } else {
                        ie_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
                        *(ie_u32x4*)(dstb + lane * 16) = v;
                    }
                }
            }
        }
        if constexpr (EPI == EPI_CURVE) {
            // fp32 image row of the OUTPUT row consumed at iteration `it` (yo = ys-2+it), plane = wave
            if (wave < 3) {
                const int yo = ys - 2 + it;
                const bool ok = (yo >= ys) && (yo < ye);
                const unsigned base = ((unsigned)wave * (unsigned)plane + (unsigned)yo * (unsigned)W) * 4u;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned off = (ok && img_xoff[j] >= 0) ? base + (unsigned)img_xoff[j] : OOB;
                    char* dstb = imgring + (it % NR) * IMG_SLOT_B + wave * (TW * 4) + j * 256;
                    if constexpr (DMA) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsi, (__attribute__((address_space(3))) void*)dstb, 4,
                                                                 (int)off, 0, 0, 0);
                    } else {
                        unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rsi, (int)off, 0, 0);
                        *(unsigned*)(dstb + lane * 4) = v;
                    }
                }
            }
        }
    };

    // ---- per-lane LDS read offsets for the B operand --------------------------
    // tile nt (16 columns), tap kx: pixel xr = 64ph + 16nt + p16 + kx - 1, chunk kq.
    // For body pixels addr(q+16) = addr(q) + 1024, so one base per kx plus two halo specials.
    auto lds_px = [&](int q) __attribute__((always_inline)) { return q * PXB + ((kq ^ ((q >> 1) & 3)) << 4); };
    int rd_base[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) rd_base[kx] = lds_px(64 * ph + 16 + p16 + kx - 1) - 1024;
    int rd_first, rd_last;
    {
        const int xf = 64 * ph + p16 - 1;              // nt = 0, kx = 0
        rd_first = lds_px(xf < 0 ? TW : xf);
        const int xl = 64 * ph + 48 + p16 + 1;         // nt = 3, kx = 2
        rd_last = lds_px(xl >= TW ? TW + 1 : xl);
    }

    // ---- output geometry ---------------------------------------------------------
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)(EPI == EPI_RELU ? p.dst : (void*)p.src0) + (size_t)n * frame_b), 0, (int)frame_b, 0x00020000);

    // DMA instructions this wave issues per row; the wait leaves the PF-1 younger rows in flight.
    // Other VM ops issued in between (stores) are younger than the row waited for, so they can only
    // make the wait stricter, never weaker.
    const int L = 2 * NSRC + (wave == 0 ? NSRC : 0) + ((EPI == EPI_CURVE && wave < 3) ? 2 : 0);
    const int wait_n = (SYNC || !DMA) ? 0 : (PF - 1) * L;

    auto epilogue_relu = [&](const ie_f32x4 (&acc)[4], int yo) __attribute__((always_inline)) {
        const bool row_ok = (yo >= ys) && (yo < ye);
        ie_u32x2 pk[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            pk[nt][0] = ie_pack2_relu<TO>(acc[nt][0], acc[nt][1]);
            pk[nt][1] = ie_pack2_relu<TO>(acc[nt][2], acc[nt][3]);
        }
        const unsigned row_b = (unsigned)yo * (unsigned)W * PXB;
        if constexpr (WIDE) {
#pragma unroll
            for (int nt = 0; nt < 4; nt += 2) {
                // even-kq lanes end with couts 4kq..4kq+7 of tile nt, odd-kq lanes with couts
                // 4(kq-1)..4(kq-1)+7 of tile nt+1 (v_permlane16_swap: odd rows of vdst <-> even rows of src)
                auto sx = __builtin_amdgcn_permlane16_swap(pk[nt][0], pk[nt + 1][0], false, false);
                auto sy = __builtin_amdgcn_permlane16_swap(pk[nt][1], pk[nt + 1][1], false, false);
                ie_u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
                const int xo = x0 + 64 * ph + 16 * (nt + (kq & 1)) + p16;
                const unsigned off = (row_ok && xo < W)
                                         ? row_b + (unsigned)xo * PXB + (unsigned)((16 * ch + 4 * (kq & 2)) * 2)
                                         : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(v, rsd, (int)off, 0, 0);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int xo = x0 + 64 * ph + 16 * nt + p16;
                const unsigned off = (row_ok && xo < W)
                                         ? row_b + (unsigned)xo * PXB + (unsigned)((16 * ch + 4 * kq) * 2)
                                         : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(pk[nt], rsd, (int)off, 0, 0);
            }
        }
    };

    // EPI_CURVE part 1: tanh(acc) -> stage[k][px] fp32 (k = 16ch + 4kq + i < 24)
    auto stage_write = [&](const ie_f32x4 (&acc)[4]) __attribute__((always_inline)) {
        float* sp = (float*)stage;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 16 * ch + 4 * kq + i;
                if (k < 24) sp[k * TW + 64 * ph + 16 * nt + p16] = ie_tanh_fast(acc[nt][i]);
            }
    };
    // EPI_CURVE part 2: one thread per pixel (tid < 128)
    auto curve_consume = [&](int it) __attribute__((always_inline)) {
        const int yo = ys - 2 + it;
        const int xo = x0 + tid;
        if (tid < TW && xo < W && yo >= ys && yo < ye) {
            const float* sp = (const float*)stage;
            const float* ip = (const float*)(imgring + (it % NR) * IMG_SLOT_B);
            const size_t pix = (size_t)yo * W + xo;
            float y[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = ip[c * TW + tid];
#pragma unroll
                for (int i = 0; i < 8; ++i) v = ie_curve_step_fma(v, sp[(3 * i + c) * TW + tid]);
                y[c] = v;
            }
            if (p.enhanced) {
                float* ep = p.enhanced + (size_t)n * 3 * plane + pix;
#pragma unroll
                for (int c = 0; c < 3; ++c) ep[(size_t)c * plane] = y[c];
            }
            if (p.u8) {
                unsigned char* up = p.u8 + ((size_t)n * plane + pix) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) up[c] = ie_quant_u8(y[c]);
            }
            if (p.adjust) {
                float* ap = p.adjust + (size_t)n * 24 * plane + pix;
#pragma unroll
                for (int k = 0; k < 24; ++k) ap[(size_t)k * plane] = sp[k * TW + tid];
            }
        }
    };

    // one input row: 12*NSRC pixel fragments, each feeding the three live accumulator sets
    // (output rows yi+1, yi, yi-1).  Straight-line code: on the first/last two rows of a block the
    // contributions to rows outside [ys, ye) are computed and discarded (the kernel is HBM-bound;
    // see DESIGN.md), which keeps the loop free of branches and accumulator copies.
    auto row_mfma = [&](int it, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
        const char* slot = smem + (it % NR) * SLOT_B;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) accN[nt] = bias4;   // first contribution to output row yi+1 (ky = 0)
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int off = (kx == 0 && nt == 0) ? rd_first : (kx == 2 && nt == 3) ? rd_last : rd_base[kx] + 1024 * nt;
                    const frag bx = *(const frag*)(slot + s * SRC_ROW_B + off);
                    if (s == 0) {
                        accN[nt] = ie_mfma16_raw<T0>(wA[0 * 3 + kx], bx, accN[nt]);
                        accM[nt] = ie_mfma16_raw<T0>(wA[1 * 3 + kx], bx, accM[nt]);
                        accO[nt] = ie_mfma16_raw<T0>(wA[2 * 3 + kx], bx, accO[nt]);
                    } else {
                        accN[nt] = ie_mfma16_raw<T1>(wA[(NSRC - 1) * 9 + 0 * 3 + kx], bx, accN[nt]);
                        accM[nt] = ie_mfma16_raw<T1>(wA[(NSRC - 1) * 9 + 1 * 3 + kx], bx, accM[nt]);
                        accO[nt] = ie_mfma16_raw<T1>(wA[(NSRC - 1) * 9 + 2 * 3 + kx], bx, accO[nt]);
                    }
                }
        // software pipeline: keep 3 LDS reads in flight ahead of their MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int f = 0; f < NSRC * 12 - 3; ++f) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
    };

    auto step = [&](int it, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
        wait_vmcnt(wait_n);               // this wave's part of row `it` has landed
        __builtin_amdgcn_s_barrier();     // everyone's part landed; everyone is done with row it-1
        issue_row(it + PF);               // refill the slot row it-1 occupied
        if constexpr (!DMA) { __syncthreads(); }
        row_mfma(it, accN, accM, accO);   // accO now holds output row yi-1 = ys-2+it
        if constexpr (EPI == EPI_RELU) {
            if (it >= 2) epilogue_relu(accO, ys - 2 + it);
        } else {
            if (it >= 2) stage_write(accO);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // staged tanh values visible to the consumer threads
            if (it >= 2) curve_consume(it);
        }
    };

    // ---- prologue: PF rows in flight ----------------------------------------------
#pragma unroll
    for (int it = 0; it < PF; ++it) issue_row(it);
    if constexpr (!DMA) { __syncthreads(); }

    ie_f32x4 a0[4], a1[4], a2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { a0[nt] = bias4; a1[nt] = bias4; a2[nt] = bias4; }
    for (int it = 0; it < n_it; it += 3) {
        step(it, a0, a1, a2);
        if (it + 1 < n_it) step(it + 1, a2, a0, a1);
        if (it + 2 < n_it) step(it + 2, a1, a2, a0);
    }
    // drain outstanding DMA before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------
// e_conv1: 3 -> 32, direct fp32 on the VALU (1.1 % of the model's FLOPs).
// Thread = one pixel, 32 output channels in registers; weights [27][32] fp32 are
// wave-uniform and come through the scalar cache one filter tap at a time.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void zdce_conv1_kernel(const float* __restrict__ img, const float* __restrict__ wk,
                                                         const float* __restrict__ bias, T* __restrict__ dst, int N,
                                                         int H, int W, int y_begin, int y_end, int n_begin) {
    ie_fp16_saturate_on();
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = y_begin + blockIdx.y;
    const int n = n_begin + blockIdx.z;
    if (x >= W || y >= y_end) return;
    const size_t plane = (size_t)H * W;
    const float* ip = img + (size_t)n * 3 * plane;
    float acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = bias[o];
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {   // one filter tap (3 input channels) per trip: 96 uniform weights
        const int ky = t / 3, kx = t - 3 * ky;
        const int yy = y + ky - 1, xx = x + kx - 1;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float* wrow = wk + t * 96;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float vin = ok ? ip[(size_t)c * plane + (size_t)yy * W + xx] : 0.f;
#pragma unroll
            for (int o = 0; o < 32; ++o) acc[o] = fmaf(vin, wrow[c * 32 + o], acc[o]);
        }
    }
    ie_u32x4* out = (ie_u32x4*)(dst + (((size_t)n * H + y) * W + x) * 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ie_u32x4 o4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o4[i] = ie_pack2_relu<T>(acc[8 * k + 2 * i], acc[8 * k + 2 * i + 1]);
        out[k] = o4;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------
// host-side launchers (called from ie_zdce_plan.hip)
// ------------------------------------------------------------------------------------
template <typename T0, typename T1, typename TO, int NSRC, int EPI>
static int launch_mfma_t(const ZdceConvLaunch& a, hipStream_t st) {
    ConvParams p;
    p.src0 = a.src0; p.src1 = a.src1; p.wpk = a.wpk; p.bias = a.bias; p.dst = a.dst;
    p.img = a.img; p.enhanced = a.enhanced; p.adjust = a.adjust; p.u8 = a.u8;
    p.N = a.N; p.H = a.H; p.W = a.W; p.y_begin = a.y_begin; p.y_end = a.y_end;
    p.rows_per_block = a.rows_per_block; p.n_begin = a.n_begin;
    p.strips = ie_cdiv(a.W, TW);
    p.chunks = ie_cdiv(a.y_end - a.y_begin, a.rows_per_block);
    const long long blocks = (long long)p.strips * p.chunks * a.n_count;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "zdce conv: bad grid %lld", blocks);
    if ((size_t)a.H * a.W * 32 * 2 >= 0x7fffffffull) IE_FAIL(IE_ERR_UNSUPPORTED, "zdce conv: frame too large for 31-bit offsets");
    constexpr int NR = (EPI == EPI_CURVE) ? 3 : 4;
    const size_t lds = (size_t)NR * ROW_PX * 32 * 2 * NSRC + (EPI == EPI_CURVE ? (24 * TW * 4 + NR * 3 * TW * 4) : 0);
    const bool sync = a.variant & IE_ZDCE_VAR_SYNC_STAGE;
    const bool narrow = a.variant & IE_ZDCE_VAR_NARROW_STORE;
    const bool nodma = a.variant & IE_ZDCE_VAR_NO_DMA;
    dim3 grid((unsigned)blocks), block(NWAVE * 64);
#define IE_GO(DMA, SYNC, WIDE) hipLaunchKernelGGL((zdce_conv3x3_mfma<T0, T1, TO, NSRC, EPI, DMA, SYNC, WIDE>), grid, block, lds, st, p)
    if (nodma) { if (narrow) IE_GO(false, true, false); else IE_GO(false, true, true); }
    else if (sync) { if (narrow) IE_GO(true, true, false); else IE_GO(true, true, true); }
    else { if (narrow) IE_GO(true, false, false); else IE_GO(true, false, true); }
#undef IE_GO
    IE_LAUNCH_CHECK();
    return IE_OK;
}

int ie_zdce_launch_mfma(const ZdceConvLaunch& a, hipStream_t st) {
    // element types: source 0, source 1 (nsrc == 2), ReLU output (epi == RELU)
    const int d0 = a.dt0, d1 = a.nsrc == 2 ? a.dt1 : a.dt0, dO = a.epi == EPI_RELU ? a.dto : a.dt0;
#define IE_DISPATCH(T0, T1, TO)                                                                  \
    if (a.nsrc == 1 && a.epi == EPI_RELU) return launch_mfma_t<T0, T1, TO, 1, EPI_RELU>(a, st);  \
    if (a.nsrc == 2 && a.epi == EPI_RELU) return launch_mfma_t<T0, T1, TO, 2, EPI_RELU>(a, st);  \
    if (a.nsrc == 2 && a.epi == EPI_CURVE) return launch_mfma_t<T0, T1, TO, 2, EPI_CURVE>(a, st);
    if (d0 == IE_F16 && d1 == IE_F16 && dO == IE_F16) { IE_DISPATCH(_Float16, _Float16, _Float16) }
    else if (d0 == IE_BF16 && d1 == IE_BF16 && dO == IE_BF16) { IE_DISPATCH(__bf16, __bf16, __bf16) }
#undef IE_DISPATCH
    // the two mixed layers of the "bf16" schedule run one at a time only on the unfused (diagnostic) schedule and as
    // e_conv3 of the fused one: x2 (fp16) -> x3 (bf16), and [x2 (fp16), x5 (bf16)] -> x6 (fp16)
    else if (d0 == IE_F16 && dO == IE_BF16 && a.nsrc == 1 && a.epi == EPI_RELU) return launch_mfma_t<_Float16, _Float16, __bf16, 1, EPI_RELU>(a, st);
    else if (d0 == IE_F16 && d1 == IE_BF16 && dO == IE_F16 && a.nsrc == 2 && a.epi == EPI_RELU) return launch_mfma_t<_Float16, __bf16, _Float16, 2, EPI_RELU>(a, st);
    IE_FAIL(IE_ERR_UNSUPPORTED, "zdce mfma conv: nsrc=%d epi=%d dtypes=%d,%d->%d", a.nsrc, a.epi, d0, d1, dO);
}

int ie_zdce_launch_conv1(const float* img, const float* wk, const float* bias, void* dst, int dtype, int N, int H,
                         int W, int y_begin, int y_end, int n_begin, int n_count, hipStream_t st) {
    dim3 grid(ie_cdiv(W, 256), y_end - y_begin, n_count), block(256);
    if (dtype == IE_F16)
        hipLaunchKernelGGL(zdce_conv1_kernel<_Float16>, grid, block, 0, st, img, wk, bias, (_Float16*)dst, N, H, W, y_begin, y_end, n_begin);
    else if (dtype == IE_BF16)
        hipLaunchKernelGGL(zdce_conv1_kernel<__bf16>, grid, block, 0, st, img, wk, bias, (__bf16*)dst, N, H, W, y_begin, y_end, n_begin);
    else
        IE_FAIL(IE_ERR_UNSUPPORTED, "zdce conv1: dtype %d", dtype);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
