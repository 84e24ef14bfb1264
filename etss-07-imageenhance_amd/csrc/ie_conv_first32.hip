// The FIRST level of a 32-channel U-Net in one launch, straight from the caller's fp32 NCHW image (unet.py:43-48 `inc` = DoubleConv(3, 32), :70-72 the
// MaxPool2d(2) the first `Down` starts with): Conv3x3(3 -> 32) + BN + ReLU -> Conv3x3(32 -> 32) + BN + ReLU -> x1 (NHWC 16-bit, the skip of the last decoder level)
// and its 2x2 max-pooled copy.  Replaces three launches -- the layout pass (NCHW fp32 -> NHWC 8-channel 16-bit) and ie_mconv_first_pool_nhwc -- and is the two-stage
// eight-wave structure of ie_conv_updc.hip at the other end of the network:
//   * stage A (4 waves = 2 output-channel fragments x 2 halves of the strip): e_conv1 on the matrix cores with K = 8 kx taps x 4 channels per MFMA (the image row
//     as [R, G, B, 0] 16-bit pixels, ie_conv_first9.hip's packing; one fragment per tile feeds the three ky taps into three rotating accumulator sets); the image
//     arrives as three fp32 plane rows by 4-byte LDS-DMA and is converted one step ahead by the same waves; finished rows (ReLU, 16 bits) go to stage B through an
//     LDS ring;
//   * stage B (4 waves): Conv3x3(32 -> 32), weights in registers; finished rows are parked as whole 64-byte pixel lines, leave one step later (1 KiB of lines per
//     store instruction), and every second step the two newest parked rows give one pooled row (the values are ReLU outputs: compared as unsigned 16-bit integers).
// Strips of 124 stored columns out of 128 computed ones (no halo columns: each layer does not use the outermost computed column of the one before); 124 and the strip
// origin are even, so pooled pixels never straddle strips.  55 KiB of LDS.
#include "ie_conv_internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int TW = 128, TS = 124, MARG = 2;
constexpr int PXB = 64;                        // one pixel of a 32-channel 16-bit tensor
constexpr int ROW_B = TW * PXB;                // 8,192: hand-off and parked rows
constexpr int IPX = 136, IPB = 8;              // image ring row: pixel i = image column xc0 - 1 + i; fragments of the last tile read up to pixel 134 (zero weights there)
constexpr int IMG_ROW_B = IPX * IPB;           // 1,088
// raw rows: three fp32 plane rows of 136 pixels = 1,632 bytes, in a 2,048-byte slot (8 DMA instructions x 256 bytes)
constexpr int RAW_SLOT_B = 2048;
constexpr int PFR = 2, NRAW = PFR + 1, NRI = 2, NRH = 3, NPARK = 3;
constexpr int OFF_RAW = 0, OFF_IMG = OFF_RAW + NRAW * RAW_SLOT_B, OFF_H = OFF_IMG + NRI * IMG_ROW_B, OFF_PARK = OFF_H + NRH * ROW_B;
constexpr int OFF_DUMP = OFF_PARK + NPARK * ROW_B, LDS_B = OFF_DUMP + 2048;      // 6,144 + 2,176 + 24,576 + 24,576 + 2,048 = 59,520
constexpr unsigned OOB = 0x80000000u;

struct F32Params {
    const float* img; const void* w1; const float* b1; const void* w2; const float* b2; void* dst; void* pool;
    int N, H, W, rows_per_block, strips, chunks;
};

template <typename T>
__device__ __forceinline__ void first32_body(const F32Params& p) {
    using frag = ie_u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int stg = wave >> 2, w4 = wave & 3, cf = w4 & 1, hf = w4 >> 1;      // stage; 16-cout fragment; half of the strip (tiles 4 hf .. 4 hf + 3)

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = b;
    const int xc0 = strip * TS - MARG;                // image column of computed column 0 (even)
    const int H = p.H, W = p.W, Hp = H >> 1, Wp = W >> 1;
    const int ys = chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, H);          // ys even
    const int n_steps = (ye - ys) + 8;

    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    };
    const size_t plane = (size_t)H * W;
    const auto rsI = mk((const char*)(p.img + (size_t)n * 3 * plane), 3 * plane * 4);
    const auto rsD = mk((char*)p.dst + (size_t)n * plane * PXB, plane * PXB);
    const size_t pool_b = (size_t)Hp * Wp * PXB;
    const auto rsP = mk((char*)p.pool + (size_t)n * pool_b, pool_b);

    // ---- weights.  Stage A: 3 fragments [ky][16 couts][32 k], k = 4 kx + channel (ie_conv3x3_c4-style image, 2 fragments of 16 couts);
    //      stage B: 9 fragments of ie_mconv_pack_weights(channels {32}, Cout 32) ----
    frag wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (stg == 0) wv[t] = t < 3 ? *(const frag*)((const char*)p.w1 + ((size_t)(cf * 3 + t) * 16 + p16) * 64 + ((kq ^ ((p16 >> 1) & 3)) << 4)) : frag{0u, 0u, 0u, 0u};
        else wv[t] = *(const frag*)((const char*)p.w2 + ((size_t)t * 32 + cf * 16 + p16) * 64 + ((kq ^ ((p16 >> 1) & 3)) << 4));
    }
    ie_f32x4 bias4;
    {
        const float* bp = stg ? p.b2 : p.b1;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = bp ? bp[16 * cf + 4 * kq + i] : 0.f;
    }

    // ---- epilogue geometry shared by the two stages: tile pair jj of a finished row as 16-byte pieces (ie_conv_updc.hip) ----
    int em_off;
    unsigned em_in = 0;
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
        // ---- stage A ---------------------------------------------------------------------------------------------------------------------
        // raw image rows: instruction j of wave w4 covers flat indices (4 j + w4) * 64 + lane of [plane][pixel] (3 x 136 floats); image row of step t: ys - 2 + t
        unsigned raw_off[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = (4 * j + w4) * 64 + lane, pl = idx / IPX, px = idx - pl * IPX, x = xc0 - 1 + px;
            raw_off[j] = (idx < 3 * IPX && x >= 0 && x < W) ? (unsigned)(((size_t)pl * plane + x) * 4) : OOB;
        }
        auto issue_raw = [&](int t) __attribute__((always_inline)) {
            const int yi = ys - 2 + t;
            const unsigned rmask = (yi >= 0 && yi < H) ? 0u : OOB;
            const unsigned row_b = rmask ? 0u : (unsigned)yi * (unsigned)W * 4u;
            char* slot = smem + OFF_RAW + (t % NRAW) * RAW_SLOT_B;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (__attribute__((address_space(3))) void*)(slot + (4 * j + w4) * 256), 4, (int)((row_b + raw_off[j]) | rmask), 0, 0, 0);
        };
        // conversion: lane i of the stage's 256 converts pixel i (< 136): three floats -> [R, G, B, 0] 16-bit; the other lanes write into the dump area
        const int ci = w4 * 64 + lane;
        const bool cv_ok = ci < IPX;
        const int cv_src = (cv_ok ? ci : 0) * 4, cv_dst = cv_ok ? ci * IPB : -1;
        auto convert = [&](int t) __attribute__((always_inline)) {          // raw row of step t -> image ring slot t % NRI
            const char* rs = smem + OFF_RAW + (t % NRAW) * RAW_SLOT_B + cv_src;
            const float r = *(const float*)rs, g = *(const float*)(rs + IPX * 4), bb = *(const float*)(rs + 2 * IPX * 4);
            char* d = cv_dst < 0 ? smem + OFF_DUMP + lane * 8 : smem + OFF_IMG + (t % NRI) * IMG_ROW_B + cv_dst;
            *(ie_u32x2*)d = ie_u32x2{ie_pack2<T>(r, g), ie_pack2<T>(bb, 0.f)};
        };
        // fragment of tile j: lane (p16, kq) holds taps kx = 2 kq, 2 kq + 1 of computed column 16 (4 hf + j) + p16: 16 bytes at ring pixel (column + 2 kq); ring pixel 0 = column -1
        int rdi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rdi[j] = (16 * (4 * hf + j) + p16 + 2 * kq) * IPB;

#pragma unroll
        for (int t = 0; t <= PFR; ++t) issue_raw(t);
        ie_wait_vm_imm<2 * PFR>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // prologue barrier 1 (all eight waves): raw row 0 has landed
        convert(0);
        auto step_a = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            ie_wait_vm_imm<2>();                         // raw row t + 1 has landed (behind it: row t + 2)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* slot = smem + OFF_IMG + (t % NRI) * IMG_ROW_B;
            frag bx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const ie_u32x2 lo = *(const ie_u32x2*)(slot + rdi[j]), hi = *(const ie_u32x2*)(slot + rdi[j] + 8);
                bx[j] = frag{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accN[j] = ie_mfma16_raw<T>(wv[0], bx[j], bias4);
                accM[j] = ie_mfma16_raw<T>(wv[1], bx[j], accM[j]);
                accO[j] = ie_mfma16_raw<T>(wv[2], bx[j], accO[j]);
            }
            convert(t + 1);                              // next step's image row (its ring slot was last read at step t - 1)
            issue_raw(t + 1 + PFR);
            const int m = ys - 3 + t;                    // the intermediate's row finished now; outside the image it is e_conv2's zero padding
            put_row(accO, smem + OFF_H + (t % NRH) * ROW_B, (m >= 0 && m < H) ? em_in : 0u);
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_a(t, a0, a1, a2);
            if (t + 1 < n_steps) step_a(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_a(t + 2, a1, a2, a0);
        }
    } else {
        // ---- stage B ---------------------------------------------------------------------------------------------------------------------
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
        auto row_mfma = [&](const char* slot, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            frag bx[3];                                                    // fragments read two ahead (a third slot ahead costs the second workgroup per CU)
            auto rdf = [&](int g) __attribute__((always_inline)) {         // g = (kx, tile)
                const int kx = g >> 2, j = g & 3;
                const int off = j == 0 ? rd0[kx] : ((kx == 2 && j == 3) ? rd23 : rd1[kx] + 1024 * (j - 1));
                bx[g % 3] = *(const frag*)(slot + off);
            };
            rdf(0); rdf(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                const int kx = g >> 2, j = g & 3;
                if (g + 2 < 12) rdf(g + 2);
                accN[j] = ie_mfma16_raw<T>(wv[0 * 3 + kx], bx[g % 3], kx == 0 ? bias4 : accN[j]);
                accM[j] = ie_mfma16_raw<T>(wv[1 * 3 + kx], bx[g % 3], accM[j]);
                accO[j] = ie_mfma16_raw<T>(wv[2 * 3 + kx], bx[g % 3], accO[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // x1 stores: instruction i of wave w4 = pixels 32 w4 + 16 i .. + 15 of the parked row, lane = (pixel, 16-byte piece): 1 KiB of whole lines
        unsigned st_col[2];
        int st_lds[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 32 * w4 + 16 * i + (lane >> 2), c = lane & 3, x = xc0 + q;
            st_col[i] = (q >= MARG && q < MARG + TS && x < W) ? (unsigned)(x * PXB + c * 16) : OOB;
            st_lds[i] = q * PXB + ((c ^ ((q >> 1) & 3)) << 4);
        }
        // pooled row: lane = (pooled pixel xp = computed columns 2 xp, 2 xp + 1; 16-byte piece)
        const int xp = (w4 * 64 + lane) >> 2, pc = lane & 3;
        const int pl0 = (2 * xp) * PXB + ((pc ^ (xp & 3)) << 4), pl1 = (2 * xp + 1) * PXB + ((pc ^ (xp & 3)) << 4);      // ((2 xp) >> 1) & 3 == ((2 xp + 1) >> 1) & 3 == xp & 3
        const unsigned pl_col = (2 * xp >= MARG && 2 * xp + 1 < MARG + TS && (xc0 >> 1) + xp < Wp) ? (unsigned)(((xc0 >> 1) + xp) * PXB + pc * 16) : OOB;

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // prologue barrier 1 (stage A converts its first image row behind it)
        // step t: the intermediate's row ys - 5 + t (stage A's step t - 2) finishes output row ys - 6 + t; accN = the row finished at step t - 1 is parked into slot
        // t % 3; the row parked at step t - 1 (ys - 8 + t) leaves, and when it is an odd row it and the one before it give pooled row (ys - 9 + t) / 2
        auto step_b = [&](int t, ie_f32x4 (&accN)[4], ie_f32x4 (&accM)[4], ie_f32x4 (&accO)[4]) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int yo = ys - 8 + t;
            const bool row_ok = yo >= ys && yo < ye;
            const unsigned rmask = row_ok ? 0u : OOB;
            const char* islot = smem + OFF_PARK + ((t + 2) % NPARK) * ROW_B;        // parked at step t - 1
            const char* jslot = smem + OFF_PARK + ((t + 1) % NPARK) * ROW_B;        // parked at step t - 2
            put_row(accN, smem + OFF_PARK + (t % NPARK) * ROW_B, 0x3u);
            row_mfma(smem + OFF_H + ((t + 1) % NRH) * ROW_B, accN, accM, accO);
            // the leaving row and the pooled row are read AFTER the MFMA stream (their slots are not written during this step): 28 registers that do not have to live
            // across it -- the kernel fits 128 VGPRs = two workgroups per CU
            ie_u32x4 ln[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) ln[i] = *(const ie_u32x4*)(islot + st_lds[i]);
            const ie_u16x8 v0 = *(const ie_u16x8*)(jslot + pl0), v1 = *(const ie_u16x8*)(jslot + pl1), v2 = *(const ie_u16x8*)(islot + pl0), v3 = *(const ie_u16x8*)(islot + pl1);
            const ie_u16x8 pv = __builtin_elementwise_max(__builtin_elementwise_max(v0, v1), __builtin_elementwise_max(v2, v3));
            const bool pool_ok = row_ok && (yo & 1) && yo - 1 >= ys && (yo >> 1) < Hp;
            const unsigned row_b = rmask ? 0u : (unsigned)yo * (unsigned)W * PXB;
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_buffer_store_b128(ln[i], rsD, (int)((row_b + st_col[i]) | rmask), 0, 0);
            const unsigned pmask = pool_ok ? 0u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ie_u32x4, pv), rsP, (int)(((pmask ? 0u : (unsigned)(yo >> 1) * (unsigned)Wp * PXB) + pl_col) | pmask), 0, 0);
        };
        for (int t = 0; t < n_steps; t += 3) {
            step_b(t, a0, a1, a2);
            if (t + 1 < n_steps) step_b(t + 1, a2, a0, a1);
            if (t + 2 < n_steps) step_b(t + 2, a1, a2, a0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void unet_first32_pool_f16(F32Params p) { first32_body<_Float16>(p); }
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void unet_first32_pool_bf16(F32Params p) { first32_body<__bf16>(p); }

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------------------------------------------
IE_EXPORT size_t ie_unet_first_packed_bytes(void) { return (size_t)2 * 3 * 16 * 64; }

// w_oihw: (32, 3, 3, 3) fp32 on the host (BatchNorm folded) -> stage A's fragment image [cf][ky][16 couts][32 k], k = 4 kx + channel
IE_EXPORT int ie_unet_first_pack_weights(const float* w_oihw_host, int dtype, void* dst_dev, void* stream) {
    if (!w_oihw_host || !dst_dev || (dtype != IE_F16 && dtype != IE_BF16)) IE_FAIL(IE_ERR_INVALID, "ie_unet_first_pack_weights: bad args");
    const size_t nb = ie_unet_first_packed_bytes();
    unsigned short* host = (unsigned short*)calloc(nb / 2, 2);
    if (!host) IE_FAIL(IE_ERR_INVALID, "ie_unet_first_pack_weights: out of memory");
    for (int cf = 0; cf < 2; ++cf)
        for (int ky = 0; ky < 3; ++ky)
            for (int r = 0; r < 16; ++r) {
                unsigned short* row = host + ((size_t)(cf * 3 + ky) * 16 + r) * 32;
                const int co = cf * 16 + r, sw = (r >> 1) & 3;
                for (int k = 0; k < 32; ++k) {
                    const int kx = k >> 2, ci = k & 3;
                    if (kx >= 3 || ci >= 3) continue;
                    const float v = w_oihw_host[(((size_t)co * 3 + ci) * 3 + ky) * 3 + kx];
                    unsigned short u;
                    if (dtype == IE_F16) { _Float16 hv = (_Float16)v; memcpy(&u, &hv, 2); }
                    else {
                        unsigned int xb; memcpy(&xb, &v, 4);
                        if ((xb & 0x7fffffffu) > 0x7f800000u) u = (unsigned short)((xb >> 16) | 0x40);
                        else { xb += 0x7fffu + ((xb >> 16) & 1u); u = (unsigned short)(xb >> 16); }
                    }
                    row[(((k >> 3) ^ sw) << 3) + (k & 7)] = u;
                }
            }
    hipError_t e = hipMemcpyAsync(dst_dev, host, nb, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    if (e != hipSuccess) IE_FAIL(IE_ERR_HIP, "ie_unet_first_pack_weights: %s", hipGetErrorString(e));
    return IE_OK;
}

IE_EXPORT int ie_unet_first_applicable(int C, int H, int W) {
    static const int off = getenv("IE_UNET_FIRST32") ? !atoi(getenv("IE_UNET_FIRST32")) : 0;       // IE_UNET_FIRST32=0: A/B switch back to layout pass + ie_mconv_first_pool_nhwc
    return (!off && C == 32 && H >= 2 && W >= 64 && (double)H * W * 64.0 < 2147483647.0 && (double)H * W * 12.0 < 2147483647.0) ? 1 : 0;
}

IE_EXPORT int ie_unet_first_pool_nchw(const float* img_nchw, const void* packed_w1, const float* bias1, const void* packed_w2, const float* bias2, void* dst, void* pool_dst,
                                      int dtype, int N, int H, int W, void* stream) {
    if (!img_nchw || !packed_w1 || !packed_w2 || !dst || !pool_dst || N <= 0) IE_FAIL(IE_ERR_INVALID, "ie_unet_first_pool_nchw: null pointer / bad batch");
    if (dtype != IE_F16 && dtype != IE_BF16) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_unet_first_pool_nchw: f16 / bf16");
    if (!ie_unet_first_applicable(32, H, W)) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_unet_first_pool_nchw: H >= 2, W >= 64 -- ask ie_unet_first_applicable first");
    F32Params p;
    memset(&p, 0, sizeof(p));
    p.img = img_nchw; p.w1 = packed_w1; p.b1 = bias1; p.w2 = packed_w2; p.b2 = bias2; p.dst = dst; p.pool = pool_dst; p.N = N; p.H = H; p.W = W;
    p.strips = ie_cdiv(W, TS);
    {
        const int slots = 512;          // 55 KiB of LDS: two workgroups per CU
        const long long base = (long long)p.strips * N;
        int best_c = 1;
        double best = -1.0;
        for (int c = 1; c <= H && c <= 64; ++c) {
            int rpb = ie_cdiv(H, c);
            rpb += rpb & 1;
            const int cc = ie_cdiv(H, rpb);
            const long long blocks = base * cc;
            const long long rounds = (blocks + slots - 1) / slots;
            const double score = (double)blocks / (double)(rounds * slots) * ((double)rpb / (double)(rpb + 8));
            if (score > best + 1e-9) { best = score; best_c = cc; }
        }
        p.rows_per_block = ie_cdiv(H, best_c);
        p.rows_per_block += p.rows_per_block & 1;
        p.chunks = ie_cdiv(H, p.rows_per_block);
    }
    const long long blocks = (long long)p.strips * p.chunks * N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "ie_unet_first_pool_nchw: bad grid");
    typedef void (*Fn)(F32Params);
    const Fn fn = dtype == IE_F16 ? unet_first32_pool_f16 : unet_first32_pool_bf16;
    void* args[] = {(void*)&p};
    IE_HIP_CHECK(hipLaunchKernel((const void*)fn, dim3((unsigned)blocks), dim3(512), args, LDS_B, (hipStream_t)stream));
    return IE_OK;
}
