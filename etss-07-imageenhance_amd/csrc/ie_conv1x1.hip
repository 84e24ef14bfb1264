// 1x1 convolution (+ bias + ReLU / LeakyReLU) over the channel concat of 16-bit NHWC sources as a streaming GEMM: the pixels of an NHWC tensor ARE the rows
// of a dense [pixel][Cin] matrix, so a workgroup takes 128 consecutive pixels, double-buffers their K chunks (32 channels = 64 B per pixel) through LDS with
// LDS-DMA and contracts them with weight fragments read straight from the packed image (L2-resident).  Reference call sites: HINet's `identity` / `csff` convs
// (hinet.py:60-83), LLUnet++'s skip convs (llunetpp.py:55-70), RelightNet's `net2_fusion` (retinexnet/model.py:66): every one of them moves more bytes than the
// matrix cores need time for (Cin Cout / (Cin + Cout) FLOP per byte, < 312 up to 512 -> 512), so the kernel is built for occupancy -- 24 KB of LDS and <= 80
// VGPRs per workgroup -- not for MFMA density.  (The tiled LDS-weights kernel these layers ran on stages a 16 x 32 pixel tile with its weights, single
// buffered: 2.1 TB/s.)
//
// Same packed weight image as ie_mconv_nhwc (ie_mconv_pack_weights, kh = kw = 1) and the same arithmetic: fp32 accumulation over the chunks in order, bias and
// activation in fp32, one rounding -- results are bit-identical to the tiled kernel's.
#include "ie_conv_internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int C1_PX = 128;                  // pixels per workgroup
constexpr int C1_BUF = C1_PX * 64;          // one K chunk of them in LDS
constexpr int C1_MAX_CHUNKS = 96;
constexpr unsigned C1_OOB = 0x80000000u;

struct C1Params {
    const void* cd_base[C1_MAX_CHUNKS];     // per K chunk: source tensor ...
    unsigned cd_info[C1_MAX_CHUNKS];        // ... (bytes per pixel of it) | (byte offset of the chunk's first channel << 16)
    int nchunks;
    const void* wpk; const float* bias; void* dst;
    long long npx;
    int Cout, cout_tiles, act;
    float slope;
};

template <int ACT> __device__ __forceinline__ float c1_act(float v, float slope) {
    if constexpr (ACT == IE_ACT_RELU) return fmaxf(v, 0.f);
    else if constexpr (ACT == IE_ACT_LRELU) return v > 0.f ? v : v * slope;
    else return v;
}

// NF = 16-cout fragments per cout tile (the packing's): the four waves are NF fragments x (4 / NF) groups of 8 NF / 4 pixel tiles
template <typename T, int NF, int ACT>
__global__ __launch_bounds__(256) void conv1x1_kernel(C1Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * C1_BUF];          // three buffers: two chunks in flight behind the one being contracted
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int RW = 4 / NF, NTW = 8 / RW;          // pixel-tile groups, pixel tiles (of 16) per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave % NF, rw = wave / NF;
    const int l16 = lane & 15, kq = lane >> 4;
    const int ct = blockIdx.x % p.cout_tiles;
    const long long px0 = (long long)(blockIdx.x / p.cout_tiles) * C1_PX;
    const int live = (int)((p.npx - px0) < C1_PX ? (p.npx - px0) : C1_PX);          // pixels of this workgroup that exist

    // this wave's two DMA instructions per chunk: pixels (2 wave + j) * 16 + (lane >> 2), 16-byte piece (lane & 3) stored at position piece ^ ((pixel >> 1) & 3)
    int dpx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) dpx[j] = (2 * wave + j) * 16 + (lane >> 2);
    auto issue = [&](int ch) __attribute__((always_inline)) {
        const unsigned info = p.cd_info[ch];
        const unsigned cs_b = info & 0xffffu, c0_b = info >> 16;
        const char* base = (const char*)p.cd_base[ch] + (size_t)px0 * cs_b;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((unsigned)live * cs_b), 0x00020000);
        unsigned char* buf = lds + (ch % 3) * C1_BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned piece = (unsigned)((lane & 3) ^ ((dpx[j] >> 1) & 3));
            const unsigned off = c0_b + piece * 16u;
            const unsigned voff = (off < cs_b && dpx[j] < live) ? (unsigned)dpx[j] * cs_b + off : C1_OOB;         // a short last chunk: missing channel pieces read as zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + (2 * wave + j) * 1024), 16, (int)voff, 0, 0, 0);
        }
    };
    const unsigned char* wbase = (const unsigned char*)p.wpk + (size_t)ct * p.nchunks * (16 * NF * 64) + (cw * 16 + l16) * 64 + ((kq ^ ((l16 >> 1) & 3)) << 4);
    auto load_w = [&](int ch) __attribute__((always_inline)) { return *(const ie_u32x4*)(wbase + (size_t)ch * (16 * NF * 64)); };

    ie_f32x4 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) acc[i] = ie_f32x4{0.f, 0.f, 0.f, 0.f};
    const int rd0 = (rw * NTW * 16 + l16) * 64 + ((kq ^ ((l16 >> 1) & 3)) << 4);       // tile i of the wave: + i * 1024

    // vector-memory order per chunk: [DMA, DMA, weight load].  At the top of chunk ch the three operations of chunk ch + 1 (issued one iteration ago) may stay
    // in flight; chunk ch + 2 is issued behind the barrier, into the buffer chunk ch - 1 was read from.
    issue(0);
    ie_u32x4 wn = load_w(0), wn2 = wn;
    if (p.nchunks > 1) { issue(1); wn2 = load_w(1); }
    for (int ch = 0; ch < p.nchunks; ++ch) {
        if (ch + 1 < p.nchunks) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // everyone's part of chunk ch has landed; everyone is done reading chunk ch - 1's buffer
        const ie_u32x4 w = wn;
        wn = wn2;
        if (ch + 2 < p.nchunks) { issue(ch + 2); wn2 = load_w(ch + 2); }
        const unsigned char* buf = lds + (ch % 3) * C1_BUF + rd0;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const ie_u32x4 b = *(const ie_u32x4*)(buf + i * 1024);
            acc[i] = ie_mfma16_raw<T>(w, b, acc[i]);
        }
    }

    // Epilogue through LDS so that every store instruction writes whole lines: lane (pixel l16 of its tile, kq) parks output channels 4 kq .. 4 kq + 3 of
    // its fragment as 8 bytes of a [pixel][32 NF bytes] image (16-byte pieces XOR-swizzled by the pixel), then the 256 threads store 16-byte pieces in pixel
    // order -- 2 NF consecutive lanes cover the cout tile's bytes of one pixel.  (8-byte pieces straight from the accumulators: 32 bytes per wave and pixel,
    // four waves per 128-byte line -- the write path's slow case, DESIGN 8 finding iv.)
    constexpr int PXB_O = 32 * NF, NPC = 2 * NF;          // bytes and 16-byte pieces of a pixel's cout tile
    __syncthreads();                                      // all reads of the input buffers are done: the image takes their place (128 x 128 B <= 16 KB)
    {
        const int co = (ct * NF + cw) * 16 + 4 * kq;
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = (p.bias && co + j < p.Cout) ? p.bias[co + j] : 0.f;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int px = (rw * NTW + i) * 16 + l16;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = c1_act<ACT>(acc[i][j] + bv[j], p.slope);
            const int piece = (cw * 2 + (kq >> 1)) ^ (px & (NPC - 1));
            *(ie_u32x2*)(lds + px * PXB_O + piece * 16 + (kq & 1) * 8) = ie_u32x2{ie_pack2<T>(v[0], v[1]), ie_pack2<T>(v[2], v[3])};
        }
    }
    __syncthreads();
    char* dst = (char*)p.dst + ((size_t)px0 * p.Cout + (size_t)ct * (16 * NF)) * sizeof(T);
    const int tile_b = min(PXB_O, (p.Cout - ct * 16 * NF) * (int)sizeof(T));          // a ragged last cout tile (Cout is a multiple of 16: whole 32-byte halves)
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int q = tid + 256 * k, px = q / NPC, part = q % NPC;
        if (px >= live || part * 16 >= tile_b) continue;
        const ie_u32x4 v = *(const ie_u32x4*)(lds + px * PXB_O + ((part ^ (px & (NPC - 1))) << 4));
        *(ie_u32x4*)(dst + (size_t)px * p.Cout * sizeof(T) + part * 16) = v;
    }
}

}  // namespace

bool ie_conv1x1_applicable(int nsrc, const int* cs, int Cout, int kh, int kw, int stride, int pad, int act, bool residual, bool clamp, bool out_f32,
                           bool pixel_shuffle, bool any_up) {
    static const int off = getenv("IE_CONV1X1") ? !atoi(getenv("IE_CONV1X1")) : 0;          // IE_CONV1X1=0: A/B switch back to the tiled kernel
    if (off || kh != 1 || kw != 1 || stride != 1 || pad != 0 || residual || clamp || out_f32 || pixel_shuffle || any_up || (Cout & 15)) return false;
    if (act != IE_ACT_NONE && act != IE_ACT_RELU && act != IE_ACT_LRELU) return false;
    int chunks = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (cs[s] <= 8 || (cs[s] & 7) || cs[s] * 2 > 0xffff) return false;          // (an 8-channel image source has its own thin packing)
        chunks += (cs[s] + 31) / 32;
    }
    return chunks <= C1_MAX_CHUNKS;
}

// srcs: nsrc NHWC tensors of npx pixels each; packed_w: ie_mconv_pack_weights(kh = kw = 1) image for those channel counts; nf / cout_tiles: the packing's.
int ie_conv1x1_launch(const void* const* srcs, const int* cs, int nsrc, const void* packed_w, const float* bias, void* dst, int dtype, long long npx, int Cout, int nf,
                      int cout_tiles, int act, float slope, hipStream_t st) {
    if (!srcs || !cs || !packed_w || !dst || npx <= 0) IE_FAIL(IE_ERR_INVALID, "ie_conv1x1: null pointer / no pixels");
    if (dtype != IE_F16 && dtype != IE_BF16) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_conv1x1: f16 / bf16");
    C1Params p;
    memset(&p, 0, sizeof(p));
    for (int s = 0; s < nsrc; ++s) {
        if (!srcs[s]) IE_FAIL(IE_ERR_INVALID, "ie_conv1x1: source %d is null", s);
        for (int c0 = 0; c0 < cs[s]; c0 += 32) {
            if (p.nchunks >= C1_MAX_CHUNKS) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_conv1x1: more than %d K chunks", C1_MAX_CHUNKS);
            p.cd_base[p.nchunks] = srcs[s];
            p.cd_info[p.nchunks] = (unsigned)(cs[s] * 2) | ((unsigned)(c0 * 2) << 16);
            ++p.nchunks;
        }
    }
    p.wpk = packed_w; p.bias = bias; p.dst = dst; p.npx = npx; p.Cout = Cout; p.cout_tiles = cout_tiles; p.act = act; p.slope = slope;
    const long long blocks = ((npx + C1_PX - 1) / C1_PX) * cout_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffLL) IE_FAIL(IE_ERR_UNSUPPORTED, "ie_conv1x1: bad grid");
    const bool h = dtype == IE_F16;
    void (*fn)(C1Params) = nullptr;
#define C1_PICK(NF_)                                                                                                                         \
    switch (act) {                                                                                                                           \
    case IE_ACT_RELU: fn = h ? conv1x1_kernel<_Float16, NF_, IE_ACT_RELU> : conv1x1_kernel<__bf16, NF_, IE_ACT_RELU>; break;                 \
    case IE_ACT_LRELU: fn = h ? conv1x1_kernel<_Float16, NF_, IE_ACT_LRELU> : conv1x1_kernel<__bf16, NF_, IE_ACT_LRELU>; break;              \
    default: fn = h ? conv1x1_kernel<_Float16, NF_, IE_ACT_NONE> : conv1x1_kernel<__bf16, NF_, IE_ACT_NONE>; break;                          \
    }
    if (nf == 4) { C1_PICK(4) } else if (nf == 2) { C1_PICK(2) } else if (nf == 1) { C1_PICK(1) } else IE_FAIL(IE_ERR_INVALID, "ie_conv1x1: nf %d", nf);
#undef C1_PICK
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, st, p);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
