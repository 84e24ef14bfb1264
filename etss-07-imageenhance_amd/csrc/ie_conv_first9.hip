// RetinexNet DecomNet's first layer (retinexnet/model.py:26: Conv2d(4, 64, kernel_size 9, padding 4, padding_mode="replicate"), no activation) as a strip kernel:
// a 9x9 convolution of a FOUR-channel image is K = 324 per output -- too shallow a K for the tiled implicit GEMM (ie_conv_mfma.hip's thin form spends 27 MFMAs of
// K = 32 per 16 x 16 outputs on it: 4 kx taps x 8 channel slots, half of them padding).
//
// Here: the image arrives as 4-channel pixels (8 bytes: [max_c, R, G, B], ie_retinex_decom_input), so one MFMA's K = 32 is EIGHT kx taps x 4 channels: two
// kx groups cover the nine taps -> 18 MFMAs per 16 x 16 outputs.  A workgroup marches down a 64-column strip one INPUT row per step; a wave owns 2 pixel tiles x 16
// output channels (8 waves = 2 tile pairs x 4 channel fragments), holds its 18 weight fragments (9 ky x 2 kx groups) in registers for the whole launch and keeps NINE
// rotating accumulator sets: input row r feeds output rows r - 4 .. r + 4 (ky = 8 .. 0), so each of the four pixel fragments a wave reads per step feeds nine MFMAs,
// and the set whose row just received ky = 8 is finished -- it is parked in LDS as whole 128-byte pixel lines and leaves one step later.  The step loop is unrolled
// nine times so that every set index is a compile-time constant.  LDS holds four image rows (LDS-DMA, 4 bytes per lane so that replicate padding is a per-lane
// column clamp) and two parked rows: 20 KiB.
#include "ie_conv_internal.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr int TW = 64;                         // columns per strip: four 16-pixel tiles, all stored
constexpr int IPB = 8;                         // 8 bytes per pixel; the strip's image row has KS / 2 halo columns either side
constexpr int IROW_B = 768;                    // one ring row (3 DMA instructions x 256 bytes; 576 used)
constexpr int PF = 3, NRI = PF + 1;
constexpr int OPB = 128, OUT_ROW_B = TW * OPB; // 8 KiB
constexpr int OFF_OUT = NRI * IROW_B, LDS_B = OFF_OUT + 2 * OUT_ROW_B;
constexpr unsigned OOB = 0x80000000u;

struct F9Params {
    const void* src; const void* w; const float* bias; void* dst;
    int N, H, W, rows_per_block, strips, chunks;
};

__device__ __forceinline__ void wait_vm_cold(int n) {
    switch (n) {
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// KS = 9: DecomNet's first layer.  KS = 3: the same structure for RelightNet's first layer (`net2_conv0_1`, model.py:58: Conv 3x3, 4 -> 64, replicate, on cat(R, L)):
// one kx group, three accumulator sets; 6 MFMAs per wave and step -- that layer is bound by its 128 output bytes per pixel.
template <typename T, int KS>
__device__ __forceinline__ void first9_body(const F9Params& p) {
    constexpr int HALO = KS / 2, NG = (KS + 7) / 8, NW = KS * NG, IPX = TW + 2 * HALO;
    using frag = ie_u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int tp = wave & 1, cf = wave >> 1;          // tile pair (tiles 2 tp, 2 tp + 1), 16-cout fragment

    int b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int chunk = b % p.chunks; b /= p.chunks;
    const int n = b;
    const int x0 = strip * TW;
    const int H = p.H, W = p.W;
    const int ys = chunk * p.rows_per_block, ye = min(ys + p.rows_per_block, H);
    const int n_steps = (ye - ys) + KS;               // input rows ys - HALO .. ye - 1 + HALO, one more step for the last parked row

    auto mk = [&](const void* base, size_t bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    };
    const size_t in_b = (size_t)H * W * IPB, out_b = (size_t)H * W * OPB;
    const auto rsS = mk((const char*)p.src + (size_t)n * in_b, in_b);
    const auto rsD = mk((char*)p.dst + (size_t)n * out_b, out_b);

    // ---- weights: image [cf][ky][g][16 couts][32 k] (k = 8 kq' + (kx & 1) * 4 + ch, kx = 8 g + 2 kq' + (kx & 1)), 16-byte pieces swizzled like the strip kernels' ----
    frag wv[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) wv[i] = *(const frag*)((const char*)p.w + ((size_t)(cf * NW + i) * 16 + p16) * 64 + ((kq ^ ((p16 >> 1) & 3)) << 4));
    ie_f32x4 bias4;
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = p.bias ? p.bias[16 * cf + 4 * kq + i] : 0.f;

    // ---- image rows: waves 0 .. 2 issue one 4-byte-per-lane DMA each (lane = half a pixel); replicate padding = clamped column / row ----
    unsigned img_off = OOB;
    if (wave < 3) {
        const int idx = wave * 64 + lane, px = idx >> 1;                       // ring pixel 0 = column x0 - HALO
        const int x = min(max(x0 - HALO + px, 0), W - 1);
        img_off = px < IPX ? (unsigned)(x * IPB + (idx & 1) * 4) : OOB;
    }
    auto issue = [&](int t) __attribute__((always_inline)) {        // input row of step t (wave-uniform branch)
        if (wave < 3) {
            const int yi = min(max(ys - HALO + t, 0), H - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(smem + (t % NRI) * IROW_B + wave * 256), 4,
                                                     (int)((unsigned)yi * (unsigned)W * IPB + img_off), 0, 0, 0);
        }
    };
    // fragment of (tile j, kx group g): lane (p16, kq) holds taps kx = 8 g + 2 kq, + 1 of pixel 16 (2 tp + j) + p16: 16 bytes at ring pixel 16 (2 tp + j) + p16 + 8 g + 2 kq
    int rd[2][NG];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < NG; ++g) rd[j][g] = (16 * (2 * tp + j) + p16 + 8 * g + 2 * kq) * IPB;

    // ---- a finished row: the wave's two tiles as 16-byte pieces of 128-byte pixel lines (lanes l / l + 16 exchange halves, like ie_conv_pair.hip) ----
    int pk_off;
    {
        const int q = 16 * (2 * tp + (kq & 1)) + p16;
        pk_off = q * OPB + (((2 * cf + (kq >> 1)) ^ (q & 7)) << 4);
    }
    // stores: wave w sends pixels 8 w .. 8 w + 7 of the parked row, lane = (pixel, 16-byte piece): 1 KiB of whole lines per instruction
    unsigned st_col;
    int st_lds;
    {
        const int q = 8 * wave + (lane >> 3), c = lane & 7, x = x0 + q;
        st_col = x < W ? (unsigned)(x * OPB + c * 16) : OOB;
        st_lds = q * OPB + ((c ^ (q & 7)) << 4);
    }

    ie_f32x4 acc[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) { acc[s][0] = bias4; acc[s][1] = bias4; }

#pragma unroll
    for (int t = 0; t < PF; ++t) issue(t);

    // step t, unrolled copy U (t % KS == U): input row ys - HALO + t.  Output row o = (input row) - ky + HALO lives in set (U - ky + KS) % KS; ky = 0 starts its set
    // from the bias; the set of ky = KS - 1 (index (U + 1) % KS) is finished: output row ys - (KS - 1) + t.
    auto step = [&](int t, auto u_tag) __attribute__((always_inline)) {
        constexpr int U = decltype(u_tag)::value;
        if (wave < 3) {
            const int nw = t == 0 ? 2 : (t == 1 ? 3 : 4);       // in flight behind row t: rows t + 1, t + 2 and the stores since
            ie_wait_vm_hot<4>(nw, [](int n_) __attribute__((always_inline)) { wait_vm_cold(n_); });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {   // the row parked in the previous step leaves
            const int yo = ys - KS + t;
            const unsigned rmask = (yo >= ys && yo < ye) ? 0u : OOB;
            const ie_u32x4 ln = *(const ie_u32x4*)(smem + OFF_OUT + ((t + 1) & 1) * OUT_ROW_B + st_lds);
            __builtin_amdgcn_raw_buffer_store_b128(ln, rsD, (int)(((rmask ? 0u : (unsigned)yo * (unsigned)W * OPB) + st_col) | rmask), 0, 0);
        }
        const char* slot = smem + (t % NRI) * IROW_B;
        frag bx[2 * NG];
#pragma unroll
        for (int f = 0; f < 2 * NG; ++f) {
            const ie_u32x2 lo = *(const ie_u32x2*)(slot + rd[f / NG][f % NG]), hi = *(const ie_u32x2*)(slot + rd[f / NG][f % NG] + 8);
            bx[f] = frag{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int f = 0; f < 2 * NG; ++f) {
            const int j = f / NG, g = f % NG;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int s = (U - ky + KS) % KS;
                acc[s][j] = ie_mfma16_raw<T>(wv[ky * NG + g], bx[f], (ky == 0 && g == 0) ? bias4 : acc[s][j]);
            }
        }
        issue(t + PF);
        {   // park the finished set (output row ys - (KS - 1) + t): no activation
            constexpr int S = (U + 1) % KS;
            const ie_u32x2 a = {ie_pack2<T>(acc[S][0][0], acc[S][0][1]), ie_pack2<T>(acc[S][0][2], acc[S][0][3])};
            const ie_u32x2 b2 = {ie_pack2<T>(acc[S][1][0], acc[S][1][1]), ie_pack2<T>(acc[S][1][2], acc[S][1][3])};
            const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
            *(ie_u32x4*)(smem + OFF_OUT + (t & 1) * OUT_ROW_B + pk_off) = ie_u32x4{sx[0], sy[0], sx[1], sy[1]};
        }
    };
    for (int t = 0; t < n_steps; t += KS) {
        step(t, std::integral_constant<int, 0>{});
        if (t + 1 < n_steps) step(t + 1, std::integral_constant<int, 1 % KS>{});
        if (t + 2 < n_steps) step(t + 2, std::integral_constant<int, 2 % KS>{});
        if constexpr (KS == 9) {
            if (t + 3 < n_steps) step(t + 3, std::integral_constant<int, 3 % KS>{});
            if (t + 4 < n_steps) step(t + 4, std::integral_constant<int, 4 % KS>{});
            if (t + 5 < n_steps) step(t + 5, std::integral_constant<int, 5 % KS>{});
            if (t + 6 < n_steps) step(t + 6, std::integral_constant<int, 6 % KS>{});
            if (t + 7 < n_steps) step(t + 7, std::integral_constant<int, 7 % KS>{});
            if (t + 8 < n_steps) step(t + 8, std::integral_constant<int, 8 % KS>{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) void conv9x9_c4_o64_f16(F9Params p) { first9_body<_Float16, 9>(p); }
__global__ __launch_bounds__(512) void conv9x9_c4_o64_bf16(F9Params p) { first9_body<__bf16, 9>(p); }
__global__ __launch_bounds__(512) void conv3x3_c4_o64_f16(F9Params p) { first9_body<_Float16, 3>(p); }
__global__ __launch_bounds__(512) void conv3x3_c4_o64_bf16(F9Params p) { first9_body<__bf16, 3>(p); }

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------------------------------------------
static size_t packed_bytes_k(int ks) { return (size_t)4 * ks * ((ks + 7) / 8) * 16 * 64; }

// w_oihw: (64, 4, ks, ks) fp32 on the host -> the kernel's fragment image [cf][ky][g][16 couts][32 k] on the device
static int pack_k(const float* w_oihw_host, int ks, int dtype, void* dst_dev, void* stream, const char* who) {
    if (!w_oihw_host || !dst_dev || (dtype != IE_F16 && dtype != IE_BF16)) IE_FAIL(IE_ERR_INVALID, "%s: bad args", who);
    const int ng = (ks + 7) / 8, nw = ks * ng;
    const size_t nb = packed_bytes_k(ks);
    unsigned short* host = (unsigned short*)calloc(nb / 2, 2);
    if (!host) IE_FAIL(IE_ERR_INVALID, "%s: out of memory", who);
    for (int cf = 0; cf < 4; ++cf)
        for (int ky = 0; ky < ks; ++ky)
            for (int g = 0; g < ng; ++g)
                for (int r = 0; r < 16; ++r) {
                    unsigned short* row = host + ((size_t)(cf * nw + ky * ng + g) * 16 + r) * 32;
                    const int co = cf * 16 + r, sw = (r >> 1) & 3;
                    for (int k = 0; k < 32; ++k) {
                        const int kx = 8 * g + (k >> 2), ci = k & 3;
                        if (kx >= ks) continue;
                        const float v = w_oihw_host[(((size_t)co * 4 + ci) * ks + ky) * ks + kx];
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
    if (e != hipSuccess) IE_FAIL(IE_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    return IE_OK;
}

static int launch_k(const void* src4, const void* packed_w, const float* bias, void* dst, int ks, int dtype, int N, int H, int W, void* stream, const char* who) {
    if (!src4 || !packed_w || !dst || N <= 0 || H < 1 || W < 1) IE_FAIL(IE_ERR_INVALID, "%s: bad args", who);
    if (dtype != IE_F16 && dtype != IE_BF16) IE_FAIL(IE_ERR_UNSUPPORTED, "%s: f16 / bf16", who);
    if ((double)H * W * 128.0 >= 2147483647.0) IE_FAIL(IE_ERR_UNSUPPORTED, "%s: frame too large for 31-bit offsets", who);
    F9Params p;
    memset(&p, 0, sizeof(p));
    p.src = src4; p.w = packed_w; p.bias = bias; p.dst = dst; p.N = N; p.H = H; p.W = W;
    p.strips = ie_cdiv(W, TW);
    {
        const int slots = ks == 9 ? 256 : 512;          // the 3x3 form fits two workgroups per CU (registers)
        const long long base = (long long)p.strips * N;
        int best_c = 1;
        double best = -1.0;
        for (int c = 1; c <= H && c <= 64; ++c) {
            const int rpb = ie_cdiv(H, c), cc = ie_cdiv(H, rpb);
            const long long blocks = base * cc;
            const long long rounds = (blocks + slots - 1) / slots;
            const double score = (double)blocks / (double)(rounds * slots) * ((double)rpb / (double)(rpb + ks));
            if (score > best + 1e-9) { best = score; best_c = cc; }
        }
        p.rows_per_block = ie_cdiv(H, best_c);
        p.chunks = ie_cdiv(H, p.rows_per_block);
    }
    const long long blocks = (long long)p.strips * p.chunks * N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_INVALID, "%s: bad grid", who);
    typedef void (*Fn)(F9Params);
    const Fn fn = ks == 9 ? (dtype == IE_F16 ? conv9x9_c4_o64_f16 : conv9x9_c4_o64_bf16) : (dtype == IE_F16 ? conv3x3_c4_o64_f16 : conv3x3_c4_o64_bf16);
    void* args[] = {(void*)&p};
    IE_HIP_CHECK(hipLaunchKernel((const void*)fn, dim3((unsigned)blocks), dim3(512), args, LDS_B, (hipStream_t)stream));
    return IE_OK;
}

IE_EXPORT size_t ie_conv9x9_c4_packed_bytes(void) { return packed_bytes_k(9); }
IE_EXPORT int ie_conv9x9_c4_pack_weights(const float* w_oihw_host, int dtype, void* dst_dev, void* stream) {
    return pack_k(w_oihw_host, 9, dtype, dst_dev, stream, "ie_conv9x9_c4_pack_weights");
}
IE_EXPORT int ie_conv9x9_c4_nhwc(const void* src4, const void* packed_w, const float* bias, void* dst, int dtype, int N, int H, int W, void* stream) {
    return launch_k(src4, packed_w, bias, dst, 9, dtype, N, H, W, stream, "ie_conv9x9_c4_nhwc");
}
IE_EXPORT size_t ie_conv3x3_c4_packed_bytes(void) { return packed_bytes_k(3); }
IE_EXPORT int ie_conv3x3_c4_pack_weights(const float* w_oihw_host, int dtype, void* dst_dev, void* stream) {
    return pack_k(w_oihw_host, 3, dtype, dst_dev, stream, "ie_conv3x3_c4_pack_weights");
}
IE_EXPORT int ie_conv3x3_c4_nhwc(const void* src4, const void* packed_w, const float* bias, void* dst, int dtype, int N, int H, int W, void* stream) {
    return launch_k(src4, packed_w, bias, dst, 3, dtype, N, H, W, stream, "ie_conv3x3_c4_nhwc");
}
