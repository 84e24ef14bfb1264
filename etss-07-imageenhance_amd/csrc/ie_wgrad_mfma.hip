// Matrix-core weight gradient of a 3x3 / stride-1 / zero-padded convolution over 16-bit NHWC tensors (SURVEY 8 a13:
// torch.autograd's conv2d weight gradient inside the reference's training step, src/mon/nn/model.py:498-615).
//
//   dW[ky][kx][ci][co] += sum over pixels (y,x) of X[y+ky-1][x+kx-1][ci] * dY[y][x][co]
//
// GEMM view per tap: D[ci][co] = A[ci][k] * B[k][co] with k = 32 consecutive pixels of one image row, on
// v_mfma_f32_16x16x32_{f16,bf16}.  Both operands are k-strided in NHWC memory (k = pixel, the contiguous axis is
// the channel), so the LDS image keeps the NHWC order ([pixel][64 channels], 128 B per pixel) and the fragments are
// gathered with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-major).
//
// Workgroup = 64 ci x 64 co x 9 taps of one source over a column strip (32 pixels wide) of `rows_per_wg` image rows;
// wave w owns ci block 16w..16w+15 -> 9 x 4 accumulators of 16x16 (144 VGPRs).  Walking down the strip, each step
// stages ONE new input row (34 pixels) and one dY row (32 pixels) - the three input rows a step needs live in a
// 4-slot ring - so X and dY are read once per (ci tile, co tile) pair.  dY fragments are shared by the 9 taps; the
// tap's horizontal shift is a pixel offset of the transposed read.  Partial sums land in dW with fp32 atomics
// (split-K over strips), like the fp32 kernel in ie_train.hip.
//
// LDS bank swizzle: pixel stride 128 B = 32 banks, so the 8 pixels a 32-lane half reads in one ds_read_b64_tr_b16
// would share 16 banks (4-way).  The 8-byte unit u of pixel p is stored at unit u ^ (s(p) << 2),
// s(p) = bit1(p) | bit3(p) << 1: the 8 pixels {c..c+3, c+8..c+11} of a half then cover all 8 (p & 1, s) pairs.
#include "ie_common.h"
#include "ie_conv_internal.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short wm_s16x4;
typedef __attribute__((address_space(3))) wm_s16x4* wm_lds_ptr;

constexpr int WM_PX = 32;                 // pixels per step = K of one MFMA
constexpr int WM_XP = WM_PX + 2;          // staged input pixels per row (one halo pixel each side)
constexpr int WM_ROWB = WM_XP * 128;      // bytes of one staged input row
constexpr int WM_DYB = WM_PX * 128;       // bytes of one staged dY row
constexpr int WM_LDS = 4 * WM_ROWB + 2 * WM_DYB;

struct WgradMParams {
    const void* x; const void* dy; float* dw;
    int N, H, W, Cs, Cout, Cin_total, cbase, rows_per_wg, co_tiles, xchunks, nci;
    // Input geometry: the tensor is xH x xW and input pixel (y, x) of the pad-1 formulation is its pixel (xs y + xoy, xs x + xox); outside = zero padding.
    //   3x3 pad 1: xs 1, offsets 0, H x W.   3x3 pad 0 ("valid" conv on a tensor the caller padded, e.g. by replication): offsets 1, (H + 2) x (W + 2).
    //   block (by, bx) of a valid 9x9 kernel's taps: offsets 3 by + 1, 3 bx + 1, (H + 8) x (W + 8).
    //   4x4 stride 2 pad 1 (hinet.py:65): the four parity sub-images (py, px) of the input, xs 2, offsets py, px -- tap ky - 1 = 2 a + py is offset a in {-1, 0, 1} there.
    int xs, xoy, xox, xH, xW;
    // tap (i, j) of the 3x3 window of this launch is tap (ts i + toy, ts j + tox) of the layer's KH x KW kernel (outside: not a tap of the layer, dropped)
    int ts, toy, tox, KH, KW;
    float* slab;       // deterministic mode: [split = blockIdx.z * gridDim.y + blockIdx.y][tile = blockIdx.x][tap][64 ci][64 co], else NULL
};

__device__ __forceinline__ int wm_swz(int pix) { return ((pix >> 1) & 1) | (((pix >> 3) & 1) << 1); }
// byte offset of 16-byte chunk c (8 channels) of staged pixel pix
__device__ __forceinline__ int wm_chunk_off(int pix, int c) { return pix * 128 + ((c ^ (wm_swz(pix) << 1)) << 4); }
// byte offset of 8-byte unit u (4 channels) of staged pixel pix
__device__ __forceinline__ int wm_unit_off(int pix, int u) { return pix * 128 + ((u ^ (wm_swz(pix) << 2)) << 3); }

template <typename T>
__device__ __forceinline__ typename ie_elem<T>::frag wm_frag(const unsigned char* base, int off_lo, int off_hi) {
    struct { wm_s16x4 lo, hi; } v;
    v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wm_lds_ptr)(base + off_lo));
    v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wm_lds_ptr)(base + off_hi));
    return __builtin_bit_cast(typename ie_elem<T>::frag, v);
}

// KS = 3: 3x3 / pad 1 (9 taps).  KS = 1: 1x1 / pad 0 -- the same staging (the halo rows and pixels are loaded and unused; the
// layer is HBM-bound either way), one tap: the centre position of the 3x3 window.
// NCO = cout blocks (of 16) per wave.  4: wave w owns cin block w and all four cout blocks (64-wide tiles).  Narrow layers split the
// waves over cout instead so that none idles: p.nci = live cin blocks of the tile rounded up to a power of two (1, 2, 4), wave w owns
// cin block w % nci and cout blocks (w / nci) * NCO ... + NCO - 1 (32 -> 32: 2 x 2 waves with one cout block each; 16 -> 64, the RGB
// edge: one cin block, four waves with one cout block each).
template <typename T, int KS, int NCO>
__global__ __launch_bounds__(256, 2) void conv_wgrad_mfma_kernel(WgradMParams p) {
    constexpr int NT = KS * KS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[WM_LDS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cb = wave % p.nci, cob0 = (wave / p.nci) * NCO;
    const int co_t = (blockIdx.x % p.co_tiles) * 64, ci_t = (blockIdx.x / p.co_tiles) * 64;
    const int n = blockIdx.y / p.xchunks, x0 = (blockIdx.y % p.xchunks) * WM_PX;
    const int y0 = blockIdx.z * p.rows_per_wg, y1 = min(y0 + p.rows_per_wg, p.H);
    const T* xn = (const T*)p.x + (size_t)n * p.xH * p.xW * p.Cs;
    const T* dn = (const T*)p.dy + (size_t)n * p.H * p.W * p.Cout;

    // loader slots: 16-byte chunk (t & 7) of staged pixel (t >> 3); threads 0..15 also take pixels 32, 33 of the input row
    const int lpix = t >> 3, lch = t & 7;
    const bool x_ch_ok = ci_t + lch * 8 < p.Cs, d_ch_ok = co_t + lch * 8 < p.Cout;
    const int gx_a = x0 - 1 + lpix, gx_b = x0 + 31 + lpix;      // input columns of the two loader slots
    const bool xa_ok = x_ch_ok && p.xs * gx_a + p.xox >= 0 && p.xs * gx_a + p.xox < p.xW;
    const bool xb_ok = t < 16 && x_ch_ok && p.xs * gx_b + p.xox < p.xW;
    const bool d_ok = d_ch_ok && x0 + lpix < p.W;
    const int xoff_a = wm_chunk_off(lpix, lch), xoff_b = wm_chunk_off(32 + lpix, lch), doff = 4 * WM_ROWB + wm_chunk_off(lpix, lch);
    const ie_u32x4 zero4 = {0u, 0u, 0u, 0u};
    // Loads are unconditional and RAW (a masked slot reads the tensor's first 16 bytes); the zero-select of masked slots happens
    // where the value is stored to LDS, a step later.  With the same number of loads in flight on every path and no use of the
    // value near the load, the compiler waits for the OLDER prefetch set with a counted vmcnt and the newer set stays in flight.
    auto ld_x = [&](int y, bool ok, int gx) -> ie_u32x4 {
        const bool v = ok && p.xs * y + p.xoy >= 0 && p.xs * y + p.xoy < p.xH;
        return *(const ie_u32x4*)(v ? xn + ((size_t)(p.xs * y + p.xoy) * p.xW + p.xs * gx + p.xox) * p.Cs + ci_t + lch * 8 : xn);
    };
    auto ld_d = [&](int y) -> ie_u32x4 {
        const bool v = d_ok && y < p.H;
        return *(const ie_u32x4*)(v ? dn + ((size_t)y * p.W + x0 + lpix) * p.Cout + co_t + lch * 8 : dn);
    };
    auto sel = [&](ie_u32x4 r, bool v) -> ie_u32x4 { return v ? r : zero4; };
    auto row_ok = [&](int y) -> bool { return p.xs * y + p.xoy >= 0 && p.xs * y + p.xoy < p.xH; };
    auto st_x = [&](int rr, ie_u32x4 a, ie_u32x4 b) {
        unsigned char* row = lds + (rr & 3) * WM_ROWB;
        *(ie_u32x4*)(row + xoff_a) = a;
        if (t < 16) *(ie_u32x4*)(row + xoff_b) = b;
    };

    // fragment read offsets
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int aoff[3][2], boff[NCO][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) aoff[kx][h] = wm_unit_off(kx + g * 8 + h * 4 + q, cb * 4 + pp);
#pragma unroll
        for (int cob = 0; cob < NCO; ++cob) boff[cob][h] = wm_unit_off(g * 8 + h * 4 + q, (cob0 + cob) * 4 + pp);
    }
    const bool wave_live = ci_t + cb * 16 < p.Cs && cob0 < 4 && co_t + cob0 * 16 < p.Cout;

    ie_f32x4 acc[NT][NCO];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NCO; ++b) acc[a][b] = ie_f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: input rows y0-1, y0, y0+1 -> ring slots 0, 1, 2; dY row y0 -> slot 0
#pragma unroll
    for (int r = 0; r < 3; ++r)
        st_x(r, sel(ld_x(y0 - 1 + r, xa_ok, gx_a), xa_ok && row_ok(y0 - 1 + r)), sel(ld_x(y0 - 1 + r, xb_ok, gx_b), xb_ok && row_ok(y0 - 1 + r)));
    *(ie_u32x4*)(lds + doff) = sel(ld_d(y0), d_ok);

    // Global loads run TWO steps ahead of their use (two register sets, the loop is unrolled by two): a step's MFMA work
    // (36 per wave, ~0.3 us) is shorter than the HBM latency.  Measured +2 % over one step of look-ahead; the large steps were the
    // branch-free MFMA chain with prefetched A fragments (+30 %) and a grid that fills whole rounds of workgroup slots (+24 %).
    struct Pf { ie_u32x4 xa, xb, d; };
    auto fetch = [&](int yx, int yd) -> Pf { return Pf{ld_x(yx, xa_ok, gx_a), ld_x(yx, xb_ok, gx_b), ld_d(yd)}; };
    auto step = [&](int y, const Pf& wr, Pf& ld) {
        __syncthreads();
        const int rr = y - y0;                      // ring index of input row y-1
        ld = fetch(y + 3, y + 2);               // rows past the chunk / image: loaded (or zeroed) and never stored
        if (wave_live) {
            const unsigned char* dyb = lds + 4 * WM_ROWB + (rr & 1) * WM_DYB;
            typename ie_elem<T>::frag b[NCO];
#pragma unroll
            for (int cob = 0; cob < NCO; ++cob) b[cob] = wm_frag<T>(dyb, boff[cob][0], boff[cob][1]);
            // A fragments run two taps ahead of the MFMAs that consume them (3 rotating register sets); output channels past
            // Cout see zero dY fragments, so all four MFMAs of a tap issue unconditionally (no branches inside the chain)
            const unsigned char* xr[3] = {lds + (rr & 3) * WM_ROWB, lds + ((rr + 1) & 3) * WM_ROWB, lds + ((rr + 2) & 3) * WM_ROWB};
            typename ie_elem<T>::frag a[3];
            if constexpr (KS == 3) {
                a[0] = wm_frag<T>(xr[0], aoff[0][0], aoff[0][1]);
                a[1] = wm_frag<T>(xr[0], aoff[1][0], aoff[1][1]);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (tap + 2 < 9) a[(tap + 2) % 3] = wm_frag<T>(xr[(tap + 2) / 3], aoff[(tap + 2) % 3][0], aoff[(tap + 2) % 3][1]);
#pragma unroll
                    for (int cob = 0; cob < NCO; ++cob) acc[tap][cob] = ie_elem<T>::mfma16(a[tap % 3], b[cob], acc[tap][cob]);
                }
            } else {
                a[0] = wm_frag<T>(xr[1], aoff[1][0], aoff[1][1]);
#pragma unroll
                for (int cob = 0; cob < NCO; ++cob) acc[0][cob] = ie_elem<T>::mfma16(a[0], b[cob], acc[0][cob]);
            }
        }
        if (y + 1 < y1) {                           // rows fetched during the previous step: input row y+2, dY row y+1
            const bool rok = row_ok(y + 2);
            st_x(rr + 3, sel(wr.xa, xa_ok && rok), sel(wr.xb, xb_ok && rok));
            *(ie_u32x4*)(lds + doff + ((rr + 1) & 1) * WM_DYB) = sel(wr.d, d_ok);
        }
    };
    Pf pa = fetch(y0 + 2, y0 + 1), pb = {zero4, zero4, zero4};
    for (int y = y0; y < y1; y += 2) {
        step(y, pa, pb);
        if (y + 1 < y1) step(y + 1, pb, pa);
    }

    if (!wave_live) return;
    const int co_l = lane & 15;
    if (p.slab) {       // deterministic mode: park this wave's partial sums, wgrad_mfma_reduce_kernel adds the splits in order
        float* sl = p.slab + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (size_t)(NT * 4096);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap)
#pragma unroll
            for (int cob = 0; cob < NCO; ++cob) {
                if (cob0 + cob >= 4) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) sl[tap * 4096 + (cb * 16 + g * 4 + j) * 64 + (cob0 + cob) * 16 + co_l] = acc[tap][cob][j];
            }
        return;
    }
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
        const int ky = p.ts * (tap / 3) + p.toy, kx = p.ts * (tap % 3) + p.tox;          // wave-uniform
        if (ky < 0 || ky >= p.KH || kx < 0 || kx >= p.KW) continue;
        const int ltap = ky * p.KW + kx;
#pragma unroll
        for (int cob = 0; cob < NCO; ++cob) {
            const int co = co_t + (cob0 + cob) * 16 + co_l;
            if (cob0 + cob >= 4 || co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = ci_t + cb * 16 + g * 4 + j;
                if (ci < p.Cs) atomicAdd(p.dw + ((size_t)ltap * p.Cin_total + p.cbase + ci) * p.Cout + co, acc[tap][cob][j]);
            }
        }
    }
}

// deterministic mode, second stage: one thread per (tile, tap, ci, co); adds the splits' slabs in split order (fp64 accumulate)
__global__ void wgrad_mfma_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit, int tiles, int co_tiles, int nt,
                                         int Cs, int Cout, int Cin_total, int cbase) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_tile = (size_t)nt * 4096;
    if (e >= (size_t)tiles * per_tile) return;
    const int tile = (int)(e / per_tile), r = (int)(e % per_tile);
    const int tap = r / 4096, ci = (tile / co_tiles) * 64 + (r % 4096) / 64, co = (tile % co_tiles) * 64 + r % 64;
    if (ci >= Cs || co >= Cout) return;
    double s = 0.0;
    for (int z = 0; z < nsplit; ++z) s += (double)slab[((size_t)z * tiles + tile) * per_tile + r];
    float* d = dw + ((size_t)tap * Cin_total + cbase + ci) * Cout + co;
    *d = (float)((double)*d + s);
}

}  // namespace

// One source of a 3x3 / stride 1 / pad 1 / zero-padded layer (ksize 3) or of a 1x1 / pad 0 layer (ksize 1):
// dw_hwio[tap][cbase + ci][co] += X (*) dY.  dtype IE_F16 / IE_BF16,
// cs % 8 == 0 and Cout % 8 == 0 (16-byte channel chunks); the caller checked applicability with ie_wgrad_mfma_applicable.
bool ie_wgrad_mfma_applicable(int kh, int kw, int stride, int pad, int pad_mode, int cs, int cout) {
    // 3x3 / pad 0: the layers behind an explicit padding op (RetinexNet's replicate-padded DecomNet / RelightNet on the 16-bit graph, retinexnet/model.py:26-36)
    const bool k3 = kh == 3 && kw == 3 && ((pad == 1 && pad_mode == IE_PAD_ZEROS) || pad == 0), k1 = kh == 1 && kw == 1 && pad == 0;
    return (k3 || k1) && stride == 1 && (cs & 7) == 0 && (cout & 7) == 0;
}

// H x W: the OUTPUT (dY) size.  ksize 3: pad 1 (input H x W) or 0 (input (H + 2) x (W + 2)); ksize 1: pad 0.
// Not in deterministic mode (one launch per block of taps, the slab reduction knows 3x3 / 1x1 only):
//   ksize 9, pad 0 (input (H + 8) x (W + 8)): nine launches of the 3x3 form, one per 3x3 block of the 81 taps;
//   stride 2 (ksize 4 pad 1; ksize 3 pad 0 / 1; input in_h x in_w): four launches, one per parity sub-image of the input (each launch's 3x3 window holds up to 4 taps of the layer).
struct WgradGeom { int xs, xoy, xox, xH, xW, ts, toy, tox, KH, KW; };
static int wgrad_mfma_launch_one(const void* x, const void* dy, int dtype, float* dw_hwio, int N, int H, int W, int cs, int cout, int cin_total, int cbase,
                                 int ksize, const WgradGeom& g, hipStream_t stream);

bool ie_wgrad_mfma_applicable_blocks(int kh, int kw, int stride, int pad, int pad_mode, int cs, int cout) {
    const bool k9 = kh == 9 && kw == 9 && stride == 1 && pad == 0, k4 = kh == 4 && kw == 4 && stride == 2 && pad == 1 && pad_mode == IE_PAD_ZEROS;
    const bool k3s2 = kh == 3 && kw == 3 && stride == 2 && (pad == 0 || (pad == 1 && pad_mode == IE_PAD_ZEROS));      // RelightNet's stride-2 layers (model.py:60-62)
    return (k9 || k4 || k3s2) && (cs & 7) == 0 && (cout & 7) == 0 && !ie_get_deterministic();
}

int ie_wgrad_mfma_launch(const void* x, const void* dy, int dtype, float* dw_hwio, int N, int H, int W, int cs, int cout, int cin_total, int cbase,
                         int ksize, int pad, int stride, int in_h, int in_w, hipStream_t stream) {
    if (ksize == 9 || stride == 2) {
        if (ie_get_deterministic()) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_convn_wgrad_nhwc: the blocked matrix-core forms do not run in deterministic mode");
        const int nb = ksize == 9 ? 3 : 2;
        for (int by = 0; by < nb; ++by)
            for (int bx = 0; bx < nb; ++bx) {
                const WgradGeom g = ksize == 9 ? WgradGeom{1, 3 * by + 1, 3 * bx + 1, H + 8, W + 8, 1, 3 * by, 3 * bx, 9, 9}
                                               : WgradGeom{2, by, bx, in_h, in_w, 2, by + pad - 2, bx + pad - 2, ksize, ksize};      // tap ky - pad = 2 a + py, a = i - 1
                const int rc = wgrad_mfma_launch_one(x, dy, dtype, dw_hwio, N, H, W, cs, cout, cin_total, cbase, 3, g, stream);
                if (rc != IE_OK) return rc;
            }
        return IE_OK;
    }
    const bool valid3 = ksize == 3 && pad == 0;
    const int o = valid3 ? 1 : 0;
    return wgrad_mfma_launch_one(x, dy, dtype, dw_hwio, N, H, W, cs, cout, cin_total, cbase, ksize, WgradGeom{1, o, o, H + 2 * o, W + 2 * o, 1, 0, 0, ksize, ksize}, stream);
}

static int wgrad_mfma_launch_one(const void* x, const void* dy, int dtype, float* dw_hwio, int N, int H, int W, int cs, int cout, int cin_total, int cbase,
                                 int ksize, const WgradGeom& g, hipStream_t stream) {
    WgradMParams p;
    p.xs = g.xs; p.xoy = g.xoy; p.xox = g.xox; p.xH = g.xH; p.xW = g.xW; p.ts = g.ts; p.toy = g.toy; p.tox = g.tox; p.KH = g.KH; p.KW = g.KW;
    p.x = x; p.dy = dy; p.dw = dw_hwio; p.N = N; p.H = H; p.W = W; p.Cs = cs; p.Cout = cout; p.Cin_total = cin_total; p.cbase = cbase;
    p.co_tiles = ie_cdiv(cout, 64);
    const int ci_tiles = ie_cdiv(cs, 64);
    p.xchunks = ie_cdiv(W, WM_PX);
    const long tiles = (long)p.co_tiles * ci_tiles, strips = (long)N * p.xchunks;
    if (tiles > 0x7fffffffL || strips > 65535) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_convn_wgrad_nhwc: grid too large (%ld tiles, %ld strips)", tiles, strips);
    if (dtype != IE_F16 && dtype != IE_BF16) IE_FAIL(IE_ERR_INVALID, "ie_convn_wgrad_nhwc: matrix-core path needs a 16-bit dtype");
    // wave layout (see the kernel): nci cin blocks x (4 / nci) cout groups of NCO blocks
    p.nci = cs <= 16 ? 1 : (cs <= 32 ? 2 : 4);
    const int live_cob = ie_cdiv(cout < 64 ? cout : 64, 16), cow = 4 / p.nci;
    int nco = ie_cdiv(live_cob, cow);
    if (nco == 3) nco = 4;
    // Split the rows into `ychunks` per strip.  A CU holds 2 workgroups of the NCO = 4 kernel (219 VGPRs; 3 at NCO = 2, 5 at NCO = 1
    // for 3x3; the 1x1 kernels are lighter still) and every workgroup costs its rows plus a fixed ~12 rows' worth of prologue and
    // atomics, so pick the split that minimises rounds x (rows + 12): a grid of 1080 workgroups on 512 slots (2.1 rounds) ran 30 %
    // slower than 960.
    const long per_cu = ksize == 3 ? (nco == 4 ? 2 : (nco == 2 ? 3 : 5)) : 5;
    const long slots = 256 * per_cu;
    long best = 1, best_cost = -1;
    const long max_chunks = H >= 32 ? H / 16 : 1;
    for (long c = 1; c <= max_chunks && c <= 64; ++c) {
        const long rows = (H + c - 1) / c, wgs = tiles * strips * ((H + rows - 1) / rows);
        const long cost = ((wgs + slots - 1) / slots) * (rows + 12);
        if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
    }
    p.rows_per_wg = ie_cdiv(H, (int)best);
    const long ychunks = ie_cdiv(H, p.rows_per_wg);
    const dim3 grid((unsigned)tiles, (unsigned)strips, (unsigned)ychunks);
    const bool h = dtype == IE_F16;
    const int ntap = ksize == 3 ? 9 : 1;
    p.slab = nullptr;
    if (ie_get_deterministic()) {
        void* ws_det = nullptr;
        const int rc = ie_det_workspace((size_t)strips * ychunks * tiles * ntap * 4096 * sizeof(float), &ws_det);
        if (rc != IE_OK) return rc;
        p.slab = (float*)ws_det;
    }
#define WM_LAUNCH(KS_, NCO_)                                                                                              \
    do {                                                                                                                  \
        if (h) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<_Float16, KS_, NCO_>), grid, dim3(256), 0, stream, p);          \
        else hipLaunchKernelGGL((conv_wgrad_mfma_kernel<__bf16, KS_, NCO_>), grid, dim3(256), 0, stream, p);              \
    } while (0)
    if (ksize == 3) {
        if (nco == 1) WM_LAUNCH(3, 1); else if (nco == 2) WM_LAUNCH(3, 2); else WM_LAUNCH(3, 4);
    } else {
        if (nco == 1) WM_LAUNCH(1, 1); else if (nco == 2) WM_LAUNCH(1, 2); else WM_LAUNCH(1, 4);
    }
#undef WM_LAUNCH
    IE_LAUNCH_CHECK();
    if (p.slab) {
        const size_t total = (size_t)tiles * ntap * 4096;
        hipLaunchKernelGGL(wgrad_mfma_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p.slab, dw_hwio,
                           (int)(strips * ychunks), (int)tiles, p.co_tiles, ntap, cs, cout, cin_total, cbase);
        IE_LAUNCH_CHECK();
    }
    return IE_OK;
}
