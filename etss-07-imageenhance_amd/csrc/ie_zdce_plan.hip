// Zero-DCE plan: host side of the fused hot path (weights repack, workspace, launch schedule).
// Replaces ZeroDCE_RE.forward (src/mon/vision/enhance/llie/zero_dce/zero_dce.py:173-191).
//
// Schedule: the batch is processed in groups of `group_frames` frames and horizontal bands of
// `band_rows` rows; inside one (group, band) job the 7 layers run back to back on the caller's
// stream, layer k computing the band plus a (7-k)-row apron so the next layer's halo is there.
// The x1..x6 intermediates live in a plan-owned workspace sized for one group (NHWC, 32 ch).
#include "ie_common.h"
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <new>

#include "ie_zdce_internal.h"

void* g_ie_stamps = nullptr;   // diagnostic builds (-DIE_STAMP): device buffer for in-kernel cycle stamps
extern "C" __attribute__((visibility("default"))) void ie_debug_set_stamps(void* p) { g_ie_stamps = p; }

namespace {
constexpr int NL = 7;
const char* const kLayerNames[NL] = {"zdce_conv1", "zdce_conv2", "zdce_conv3", "zdce_conv4",
                                     "zdce_conv5", "zdce_conv6", "zdce_conv7_curve"};
// profiling slots used by the fused schedule (slot index = last layer of the kernel)
const char* const kFusedNames[NL] = {"", "zdce_k12_conv1_conv2", "zdce_conv3", "", "zdce_k45_conv4_conv5", "",
                                     "zdce_k67_conv6_conv7_curve"};
// default schedule: e_conv3 rides in the three-layer kernel (slot 4), slot 2 stays empty
const char* const kFused3Names[NL] = {"", "zdce_k12_conv1_conv2", "", "", "zdce_k345_conv3_conv4_conv5", "",
                                      "zdce_k67_conv6_conv7_curve"};

unsigned short f32_to_f16_bits(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
unsigned short f32_to_bf16_bits(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
int env_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
}  // namespace

struct ie_zdce_plan {
    ie_zdce_config cfg;
    // element type of every 16-bit intermediate x1..x6 (index 1..6; x1's slot holds P on the fused schedule).
    // compute_dtype F16: all fp16.  BF16 + IE_ZDCE_VAR_PURE_BF16: all bf16 (2e-3 of the fp32 result: bf16's 8-bit
    // mantissa on the WEIGHTS alone costs 1.3e-3, tools/bf16_attrib.py).  BF16 (default): the tensors that carry
    // the error -- x1, x2, x6, P and the weights contracted with them -- are fp16, x3/x4/x5 and theirs stay bf16.
    int xdt[7] = {0, 0, 0, 0, 0, 0, 0};
    std::vector<float> w[NL], b[NL];
    bool set[NL] = {false, false, false, false, false, false, false};
    bool finalized = false;
    // device weights: one allocation
    char* d_w = nullptr;
    size_t off_w[NL] = {0}, off_b[NL] = {0};
    size_t off_w1m = 0;   // e_conv1 as MFMA A fragments (fused K12 kernel)
    size_t off_w27 = 0;   // K12 stage B: e_conv2 fragments followed by the x1 half of e_conv7
    size_t off_w7b = 0;   // K67 stage B: the x6 half of e_conv7
    size_t off_b7p = 0;   // e_conv7's bias in K12's row order (what P is seeded with)
    size_t off_w1x = 0;   // round-4 schedule: e_conv1 as [cout half][ky][lane][8] fragments, K = 8 kx taps x 4 channels (K12X and K67X)
    size_t off_w1q = 0;   // K67X stage A (both cout fragments per wave): e_conv1 as [cout half][lane][8], one K = 32 fragment over a 3-row window
    size_t off_w7x = 0;   // K67X stage B: e_conv7 over [x1, x6], [row tile][source * 9 + tap][lane][8], rows permuted and scaled like off_w7b
    size_t off_b7x = 0;   // ... and its bias in that row order
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    float* img_ws = nullptr;      // ie_zdce_forward_u8: the current frame group as NCHW fp32
    size_t img_ws_bytes = 0;
    int cap_frames = 0, cap_H = 0, cap_W = 0;
    // schedule knobs
    int rows_per_block = 0, band_rows = 0, group_frames = 0;
    // last forward (for get_feature)
    int last_N = 0, last_H = 0, last_W = 0;
    bool last_whole = false;
    // profiling hooks
    bool prof = false;
    std::vector<hipEvent_t> ev;      // pairs
    std::vector<int> ev_layer;
};

static size_t elem_size(const ie_zdce_plan* p) { return ie_dtype_size(p->cfg.compute_dtype); }
static size_t frame_elems(int H, int W, int C) { return (size_t)H * W * C; }

static void pick_schedule(const ie_zdce_plan* p, int N, int H, int W, int* group, int* band, int* rb) {
    const bool f32 = p->cfg.compute_dtype == IE_F32;
    // The fused kernels run one 8-wave workgroup per CU.  A launch should be ONE full round of the 256 CUs with equal-length blocks
    // (no tail) and as few row chunks as possible -- every chunk re-runs 6-11 apron rows (4 chunks of 270 rows cost 3 %): at 1080p
    // (16 strips) that is 16 frames per group and whole-height strips (tools/tune.py: 20.36 -> 19.94 ms per 64 frames against 8 frames
    // x 4 chunks).  Fewer frames than that: more chunks per strip.
    const long long strips = (W + 123) / 124;
    int g = p->group_frames > 0 ? p->group_frames : (f32 ? 1 : (int)(256 / strips > 0 ? 256 / strips : 1));
    if (g > 32) g = 32;
    if (g > N) g = N;
    int bd = p->band_rows > 0 ? p->band_rows : H;
    if (bd > H) bd = H;
    int r = p->rows_per_block > 0 ? p->rows_per_block : 0;
    if (r <= 0) {
        long long chunks = 256 / ((long long)g * strips);
        if (chunks < 1) chunks = 1;
        long long rbv = (bd + chunks - 1) / chunks;
        if (rbv < 8) rbv = 8;
        r = (int)rbv;
    }
    (void)W;
    *group = g; *band = bd; *rb = r;
}

IE_EXPORT int ie_zdce_create(const ie_zdce_config* cfg, ie_zdce_plan** out) {
    if (!cfg || !out) IE_FAIL(IE_ERR_INVALID, "ie_zdce_create: null argument");
    if (cfg->in_channels != 3 || cfg->num_channels != 32 || cfg->num_iters != 8)
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_zdce_create: this build supports in_channels=3, num_channels=32, num_iters=8 (got %d, %d, %d)",
                cfg->in_channels, cfg->num_channels, cfg->num_iters);
    if (cfg->compute_dtype != IE_F32 && cfg->compute_dtype != IE_F16 && cfg->compute_dtype != IE_BF16)
        IE_FAIL(IE_ERR_INVALID, "ie_zdce_create: compute_dtype %d", cfg->compute_dtype);
    int ndev = 0;
    IE_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) IE_FAIL(IE_ERR_INVALID, "ie_zdce_create: device %d of %d", cfg->device, ndev);
    ie_zdce_plan* p = new (std::nothrow) ie_zdce_plan();
    if (!p) IE_FAIL(IE_ERR_NOMEM, "ie_zdce_create: out of host memory");
    p->cfg = *cfg;
    {
        const bool mixed = cfg->compute_dtype == IE_BF16 && !(cfg->variant & IE_ZDCE_VAR_PURE_BF16);
        for (int k = 1; k <= 6; ++k) p->xdt[k] = cfg->compute_dtype;
        if (mixed) { p->xdt[1] = p->xdt[2] = p->xdt[6] = IE_F16; }
    }
    p->rows_per_block = env_int("IE_ZDCE_RB", 0);
    p->band_rows = env_int("IE_ZDCE_BAND", 0);
    p->group_frames = env_int("IE_ZDCE_GROUP", 0);
    *out = p;
    return IE_OK;
}

IE_EXPORT int ie_zdce_set_tuning(ie_zdce_plan* p, int rows_per_block, int band_rows, int group_frames) {
    if (!p) IE_FAIL(IE_ERR_INVALID, "ie_zdce_set_tuning: null plan");
    p->rows_per_block = rows_per_block; p->band_rows = band_rows; p->group_frames = group_frames;
    return IE_OK;
}

IE_EXPORT int ie_zdce_set_layer(ie_zdce_plan* p, int layer, const float* w, const float* b, int Cout, int Cin) {
    if (!p || !w || !b) IE_FAIL(IE_ERR_INVALID, "ie_zdce_set_layer: null argument");
    if (layer < 1 || layer > NL) IE_FAIL(IE_ERR_INVALID, "ie_zdce_set_layer: layer %d not in 1..7", layer);
    const int nc = p->cfg.num_channels;
    const int eCo = layer == 7 ? 3 * p->cfg.num_iters : nc;
    const int eCi = layer == 1 ? p->cfg.in_channels : (layer >= 5 ? 2 * nc : nc);
    if (Cout != eCo || Cin != eCi)
        IE_FAIL(IE_ERR_INVALID, "ie_zdce_set_layer: e_conv%d expects (%d,%d,3,3), got (%d,%d,3,3)", layer, eCo, eCi, Cout, Cin);
    p->w[layer - 1].assign(w, w + (size_t)Cout * Cin * 9);
    p->b[layer - 1].assign(b, b + Cout);
    p->set[layer - 1] = true;
    p->finalized = false;
    return IE_OK;
}

IE_EXPORT int ie_zdce_finalize(ie_zdce_plan* p, void* stream) {
    if (!p) IE_FAIL(IE_ERR_INVALID, "ie_zdce_finalize: null plan");
    for (int l = 0; l < NL; ++l) if (!p->set[l]) IE_FAIL(IE_ERR_STATE, "ie_zdce_finalize: e_conv%d not set", l + 1);
    IE_HIP_CHECK(hipSetDevice(p->cfg.device));
    const int dt = p->cfg.compute_dtype;
    const int nc = 32;
    std::vector<char> host;
    auto align = [&](size_t a) { while (host.size() % a) host.push_back(0); };
    const int s0x[NL] = {0, 1, 2, 3, 3, 2, 1}, s1x[NL] = {0, 0, 0, 0, 4, 5, 6};   // source tensors of each layer
    for (int l = 0; l < NL; ++l) {
        const int Co = l == 6 ? 24 : nc;
        const int Ci = l == 0 ? 3 : (l >= 4 ? 64 : 32);
        const float* w = p->w[l].data();
        align(256);
        p->off_w[l] = host.size();
        if (dt == IE_F32) {
            // HWIO fp32, conv1's Cin padded 3 -> 4 (zero weights) so the input is a 16-byte NHWC4 pixel
            const int Cip = l == 0 ? 4 : Ci;
            std::vector<float> hw((size_t)9 * Cip * Co, 0.f);
            for (int o = 0; o < Co; ++o) for (int c = 0; c < Ci; ++c) for (int t = 0; t < 9; ++t)
                hw[((size_t)t * Cip + c) * Co + o] = w[((size_t)o * Ci + c) * 9 + t];
            const char* q = (const char*)hw.data();
            host.insert(host.end(), q, q + hw.size() * 4);
        } else if (l == 0) {
            // conv1 VALU kernel: [(ky*3+kx)*3 + c][32] fp32
            std::vector<float> hw((size_t)27 * 32, 0.f);
            for (int o = 0; o < 32; ++o) for (int c = 0; c < 3; ++c) for (int t = 0; t < 9; ++t)
                hw[((size_t)t * 3 + c) * 32 + o] = w[((size_t)o * 3 + c) * 9 + t];
            const char* q = (const char*)hw.data();
            host.insert(host.end(), q, q + hw.size() * 4);
        } else {
            // MFMA (16x16x32) A-operand fragments, per cout half hc:
            //   [hc][f = s*9 + ky*3 + kx][lane][j] = W[cout 16hc + (lane&15)][cin s*32 + 8*(lane>>4) + j][ky][kx]
            // couts >= Co (e_conv7: 24..31) are zero rows.
            const int nsrc = Ci / 32;
            std::vector<unsigned short> hw((size_t)2 * nsrc * 9 * 64 * 8, 0);
            for (int hc = 0; hc < 2; ++hc) for (int s = 0; s < nsrc; ++s) for (int t = 0; t < 9; ++t)
                for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
                    const int f = s * 9 + t;
                    const int co = 16 * hc + (lane & 15);
                    const int ci = s * 32 + 8 * (lane >> 4) + j;
                    const float v = co < Co ? w[((size_t)co * Ci + ci) * 9 + t] : 0.f;
                    const int sdt = p->xdt[s == 0 ? s0x[l] : s1x[l]];      // the weights take their source tensor's element type
                    hw[(((size_t)hc * nsrc * 9 + f) * 64 + lane) * 8 + j] = sdt == IE_F16 ? f32_to_f16_bits(v) : f32_to_bf16_bits(v);
                }
            const char* q = (const char*)hw.data();
            host.insert(host.end(), q, q + hw.size() * 2);
        }
        align(256);
        p->off_b[l] = host.size();
        std::vector<float> hb(32, 0.f);
        for (int o = 0; o < Co; ++o) hb[o] = p->b[l][o];
        const char* q = (const char*)hb.data();
        host.insert(host.end(), q, q + 32 * 4);
    }
    if (dt != IE_F32) {
        // e_conv1 for the fused K12 kernel: two K=32 fragments (m = 0,1) per cout half over the
        // interleaved [px][R,G,B,0] image rows:  [hc][m][lane][j] = W1[16hc + (lane&15)][c][ky][kx] with
        //   kq = lane>>4, ky = 2m + (kq>>1), kx = 2(kq&1) + (j>>2), c = j&3   (ky, kx, c == 3: zero)
        align(256);
        p->off_w1m = host.size();
        const float* w = p->w[0].data();
        std::vector<unsigned short> hw((size_t)2 * 2 * 64 * 8, 0);
        for (int hc = 0; hc < 2; ++hc) for (int m = 0; m < 2; ++m) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int kq = lane >> 4, ky = 2 * m + (kq >> 1), kx = 2 * (kq & 1) + (j >> 2), c = j & 3;
            float v = 0.f;
            if (ky < 3 && kx < 3 && c < 3) v = w[((size_t)(16 * hc + (lane & 15)) * 3 + c) * 9 + ky * 3 + kx];
            hw[(((size_t)hc * 2 + m) * 64 + lane) * 8 + j] = p->xdt[1] == IE_F16 ? f32_to_f16_bits(v) : f32_to_bf16_bits(v);
        }
        const char* q = (const char*)hw.data();
        host.insert(host.end(), q, q + hw.size() * 2);
    }
    if (dt != IE_F32) {
        // e_conv7([x1, x6]) = W7a * x1 + W7b * x6.  K12 computes P = W7a*x1 + bias7 next to e_conv2 (one fragment
        // buffer [hc][e_conv2 9 | e_conv7a 9]), K67 adds W7b * x6.  Same fragment order as above.
        // e_conv7's rows are PERMUTED for the shipped last kernel (K67N, ie_zdce_fused.hip): its stage B wants lane (pixel, kq) of the
        // two 16-row MFMA tiles to hold the eight curve maps r_0..r_7 of ONE colour kq, so row 4 kq + i of tile j is channel 3 (4 j + i) + kq
        // (r_i of colour c is channel 3 i + c, zero_dce.py:183-188; kq = 3: zero rows).  K12 writes P so that the 16 bytes a lane seeds both
        // tiles from are contiguous: position 8 kq + 4 j + i of the 64-byte P' record holds that same channel; positions 24..31 are zero
        // rows whose bytes K12 replaces by the pixel's fp32 R, G, B (step_a12).  IE_ZDCE_VAR_K67_LEGACY keeps channel order.
        const bool legacy67 = (p->cfg.variant & IE_ZDCE_VAR_K67_LEGACY) != 0;
        auto chan_k12 = [&](int pos) { return legacy67 ? (pos < 24 ? pos : -1) : (pos < 24 ? 3 * (4 * ((pos >> 2) & 1) + (pos & 3)) + (pos >> 3) : -1); };
        auto chan_k67 = [&](int row) { const int j = row >> 4, m = row & 15; return legacy67 ? (row < 24 ? row : -1) : ((m >> 2) < 3 ? 3 * (4 * j + (m & 3)) + (m >> 2) : -1); };
        auto chan_id = [&](int row) { return row; };
        // ... and SCALED by 2 / ln 2 (weights, bias, hence P): the accumulators then hold log2(e) * 2x and tanh(x) = 1 - 2 / (exp2(acc) + 1) needs no
        // multiply in the kernel (one vector instruction less per curve map and pixel).  The product is rounded to 16 bits once, like the weight itself.
        const float s7 = legacy67 ? 1.0f : 2.885390081777927f;
        auto pack9 = [&](std::vector<unsigned short>& hw, size_t base, int hc, const float* w, int Co, int Ci, int ci0, int sdt, auto&& row_chan, float scale = 1.0f) {
            for (int t = 0; t < 9; ++t) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
                const int co = row_chan(16 * hc + (lane & 15));
                const int ci = ci0 + 8 * (lane >> 4) + j;
                const float v = (co >= 0 && co < Co) ? scale * w[((size_t)co * Ci + ci) * 9 + t] : 0.f;
                hw[base + ((size_t)t * 64 + lane) * 8 + j] = sdt == IE_F16 ? f32_to_f16_bits(v) : f32_to_bf16_bits(v);
            }
        };
        align(256);
        p->off_w27 = host.size();
        {
            std::vector<unsigned short> hw((size_t)2 * 18 * 64 * 8, 0);
            for (int hc = 0; hc < 2; ++hc) {
                pack9(hw, ((size_t)hc * 18 + 0) * 64 * 8, hc, p->w[1].data(), 32, 32, 0, p->xdt[1], chan_id);
                pack9(hw, ((size_t)hc * 18 + 9) * 64 * 8, hc, p->w[6].data(), 24, 64, 0, p->xdt[1], chan_k12, s7);
            }
            const char* q = (const char*)hw.data();
            host.insert(host.end(), q, q + hw.size() * 2);
        }
        align(256);
        p->off_w7b = host.size();
        {
            std::vector<unsigned short> hw((size_t)2 * 9 * 64 * 8, 0);
            for (int hc = 0; hc < 2; ++hc) pack9(hw, (size_t)hc * 9 * 64 * 8, hc, p->w[6].data(), 24, 64, 32, p->xdt[6], chan_k67, s7);
            const char* q = (const char*)hw.data();
            host.insert(host.end(), q, q + hw.size() * 2);
        }
        align(256);
        p->off_b7p = host.size();
        {
            std::vector<float> hb(32, 0.f);
            for (int pos = 0; pos < 32; ++pos) { const int c = chan_k12(pos); hb[pos] = c >= 0 ? s7 * p->b[6][c] : 0.f; }
            const char* q = (const char*)hb.data();
            host.insert(host.end(), q, q + 32 * 4);
        }
    }
    if (dt != IE_F32) {
        // Round-4 schedule (ie_zdce_x.hip).  e_conv1 over [R, G, B, 0] 16-bit pixels, one K = 32 fragment per kernel row ky:
        //   [hc][ky][lane][j] = W1[16 hc + (lane & 15)][c][ky][kx],  kx = 2 (lane >> 4) + (j >> 2), c = j & 3   (kx >= 3, c == 3: zero)
        align(256);
        p->off_w1x = host.size();
        {
            const float* w = p->w[0].data();
            std::vector<unsigned short> hw((size_t)2 * 3 * 64 * 8, 0);
            for (int hc = 0; hc < 2; ++hc) for (int ky = 0; ky < 3; ++ky) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
                const int kx = 2 * (lane >> 4) + (j >> 2), c = j & 3;
                float v = 0.f;
                if (kx < 3 && c < 3) v = w[((size_t)(16 * hc + (lane & 15)) * 3 + c) * 9 + ky * 3 + kx];
                hw[(((size_t)hc * 3 + ky) * 64 + lane) * 8 + j] = p->xdt[1] == IE_F16 ? f32_to_f16_bits(v) : f32_to_bf16_bits(v);
            }
            const char* q = (const char*)hw.data();
            host.insert(host.end(), q, q + hw.size() * 2);
        }
        // ... and as ONE fragment per cout half over the 27 taps of a 3-row window: lane group kq < 3 carries window row ky = kq, k = 8 kq + j with j = 3 kx + c for the
        // first eight (kx, c) pairs of the row; the ninth -- (kx 2, B) -- of the three rows sits in group 3 (k = 24 + ky), the rest of group 3 is zero
        align(256);
        p->off_w1q = host.size();
        {
            const float* w = p->w[0].data();
            std::vector<unsigned short> hw((size_t)2 * 64 * 8, 0);
            for (int hc = 0; hc < 2; ++hc) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
                const int co = 16 * hc + (lane & 15), g = lane >> 4;
                float v = 0.f;
                if (g < 3) { const int kx = j / 3, c = j % 3; v = w[((size_t)co * 3 + c) * 9 + g * 3 + kx]; }
                else if (j < 3) v = w[((size_t)co * 3 + 2) * 9 + j * 3 + 2];
                hw[((size_t)hc * 64 + lane) * 8 + j] = p->xdt[1] == IE_F16 ? f32_to_f16_bits(v) : f32_to_bf16_bits(v);
            }
            const char* q = (const char*)hw.data();
            host.insert(host.end(), q, q + hw.size() * 2);
        }
        // e_conv7 over BOTH sources for K67X's stage B: tile j = 16 rows, row 4 kq + i = curve map r_{4j+i} of colour kq (channel 3 (4 j + i) + kq,
        // zero_dce.py:183-188; kq = 3: zero rows), scaled by 2 / ln 2 so that exp(2x) is one v_exp_f32 of the accumulator.
        const float s7 = 2.885390081777927f;
        auto chan67 = [&](int row) { const int j = row >> 4, m = row & 15; return (m >> 2) < 3 ? 3 * (4 * j + (m & 3)) + (m >> 2) : -1; };
        align(256);
        p->off_w7x = host.size();
        {
            const float* w = p->w[6].data();
            std::vector<unsigned short> hw((size_t)2 * 18 * 64 * 8, 0);
            for (int tj = 0; tj < 2; ++tj) for (int s = 0; s < 2; ++s) for (int t = 0; t < 9; ++t) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
                const int co = chan67(16 * tj + (lane & 15));
                const int ci = 32 * s + 8 * (lane >> 4) + j;
                const float v = co >= 0 ? s7 * w[((size_t)co * 64 + ci) * 9 + t] : 0.f;
                const int sdt = p->xdt[s == 0 ? 1 : 6];
                hw[(((size_t)tj * 18 + s * 9 + t) * 64 + lane) * 8 + j] = sdt == IE_F16 ? f32_to_f16_bits(v) : f32_to_bf16_bits(v);
            }
            const char* q = (const char*)hw.data();
            host.insert(host.end(), q, q + hw.size() * 2);
        }
        align(256);
        p->off_b7x = host.size();
        {
            std::vector<float> hb(32, 0.f);
            for (int row = 0; row < 32; ++row) { const int c = chan67(row); hb[row] = c >= 0 ? s7 * p->b[6][c] : 0.f; }
            const char* q = (const char*)hb.data();
            host.insert(host.end(), q, q + 32 * 4);
        }
    }
    if (p->d_w) { (void)hipFree(p->d_w); p->d_w = nullptr; }
    IE_HIP_CHECK(hipMalloc(&p->d_w, host.size()));
    hipStream_t st = (hipStream_t)stream;
    IE_HIP_CHECK(hipMemcpyAsync(p->d_w, host.data(), host.size(), hipMemcpyHostToDevice, st));
    IE_HIP_CHECK(hipStreamSynchronize(st));  // host staging vector dies at return
    p->finalized = true;
    return IE_OK;
}

// bytes of workspace per frame
static size_t ws_per_frame(const ie_zdce_plan* p, int H, int W) {
    if (p->cfg.compute_dtype == IE_F32) return (2 * frame_elems(H, W, 4) + 6 * frame_elems(H, W, 32) + frame_elems(H, W, 24)) * 4;
    return 6 * frame_elems(H, W, 32) * elem_size(p);
}

IE_EXPORT int ie_zdce_reserve(ie_zdce_plan* p, int N, int H, int W) {
    if (!p || N <= 0 || H <= 0 || W <= 0) IE_FAIL(IE_ERR_INVALID, "ie_zdce_reserve: bad args");
    int g, bd, rb;
    pick_schedule(p, N, H, W, &g, &bd, &rb);
    const size_t need = ws_per_frame(p, H, W) * g + 1024;
    if (need > p->ws_bytes) {
        IE_HIP_CHECK(hipSetDevice(p->cfg.device));
        if (p->ws) { IE_HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(p->ws); p->ws = nullptr; p->ws_bytes = 0; }
        hipError_t e = hipMalloc(&p->ws, need);
        if (e != hipSuccess) { ie_set_error("ie_zdce_reserve: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e)); return IE_ERR_NOMEM; }
        p->ws_bytes = need;
    }
    p->cap_frames = g; p->cap_H = H; p->cap_W = W;
    return IE_OK;
}

IE_EXPORT size_t ie_zdce_workspace_bytes(const ie_zdce_plan* p) { return p ? p->ws_bytes : 0; }

namespace {
struct ProfScope {
    ie_zdce_plan* p; hipStream_t st; bool on;
    ProfScope(ie_zdce_plan* p_, int layer, hipStream_t st_) : p(p_), st(st_), on(p_->prof) {
        if (!on) return;
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        p->ev.push_back(a); p->ev.push_back(b); p->ev_layer.push_back(layer);
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() { if (on) (void)hipEventRecord(p->ev.back(), st); }
};
}  // namespace

static int forward_f32(ie_zdce_plan* p, const float* image, float* enhanced, float* adjust, uint8_t* u8, int N, int H,
                       int W, hipStream_t st, int group) {
    // parity mode: every layer on the generic fp32 VALU conv; whole frames, `group` frames per pass
    const size_t e4 = frame_elems(H, W, 4), e32 = frame_elems(H, W, 32), e24 = frame_elems(H, W, 24);
    for (int n0 = 0; n0 < N; n0 += group) {
        const int g = (N - n0) < group ? (N - n0) : group;
        float* x0 = (float*)p->ws;
        float* x[7];
        x[0] = x0;
        for (int k = 1; k <= 6; ++k) x[k] = x0 + e4 * group + e32 * group * (k - 1);
        float* adj = x0 + e4 * group + e32 * group * 6;
        const float* img = image + (size_t)n0 * 3 * H * W;
        int rc = ie_nchw_f32_to_nhwc(img, x0, IE_F32, g, 3, H, W, 4, st);
        if (rc) return rc;
        const int s0i[NL] = {0, 1, 2, 3, 3, 2, 1}, s1i[NL] = {-1, -1, -1, -1, 4, 5, 6};
        for (int l = 0; l < NL; ++l) {
            ProfScope ps(p, l, st);
            ie_conv2d_desc d;
            memset(&d, 0, sizeof(d));
            d.N = g; d.H = H; d.W = W; d.C0 = l == 0 ? 4 : 32; d.C1 = s1i[l] >= 0 ? 32 : 0; d.Cout = l == 6 ? 24 : 32;
            d.kh = d.kw = 3; d.stride = 1; d.pad = 1; d.pad_mode = IE_PAD_ZEROS; d.groups = 1;
            d.act = l == 6 ? IE_ACT_TANH : IE_ACT_RELU;
            rc = ie_conv2d_nhwc_f32(&d, x[s0i[l]], s1i[l] >= 0 ? x[s1i[l]] : nullptr, (const float*)(p->d_w + p->off_w[l]),
                                    (const float*)(p->d_w + p->off_b[l]), l == 6 ? adj : x[l + 1], st);
            if (rc) return rc;
        }
        float* enh = enhanced ? enhanced + (size_t)n0 * 3 * H * W : nullptr;
        float* tmp_enh = enh;
        if (!tmp_enh) tmp_enh = adj + e24 * group;  // plan-owned scratch for u8-only output
        rc = ie_le_curve(img, adj, 24, tmp_enh, g, H, W, 8, st);
        if (rc) return rc;
        if (u8) { rc = ie_nchw_f32_to_u8_nhwc(tmp_enh, u8 + (size_t)n0 * H * W * 3, g, 3, H, W, 0, st); if (rc) return rc; }
        if (adjust) { rc = ie_nhwc_to_nchw_f32(adj, IE_F32, adjust + (size_t)n0 * 24 * H * W, g, 24, H, W, 24, st); if (rc) return rc; }
    }
    return IE_OK;
}

static int forward_mfma(ie_zdce_plan* p, const float* image, float* enhanced, float* adjust, uint8_t* u8, int N, int H,
                        int W, hipStream_t st, int group, int band, int rb) {
    const size_t fb = frame_elems(H, W, 32) * elem_size(p);   // bytes per frame of one intermediate
    for (int n0 = 0; n0 < N; n0 += group) {
        const int g = (N - n0) < group ? (N - n0) : group;
        // the kernels index frames by absolute n: bias the workspace pointers by -n0 frames
        char* x[7];
        x[0] = nullptr;
        for (int k = 1; k <= 6; ++k) x[k] = p->ws + fb * group * (k - 1) - fb * (size_t)n0;
        for (int b0 = 0; b0 < H; b0 += band) {
            const int b1 = (b0 + band) < H ? (b0 + band) : H;
            auto lo = [&](int k) { int v = b0 - (7 - k); return v < 0 ? 0 : v; };
            auto hi = [&](int k) { int v = b1 + (7 - k); return v > H ? H : v; };
            const int s0i[NL] = {0, 1, 2, 3, 3, 2, 1}, s1i[NL] = {-1, -1, -1, -1, 4, 5, 6};
            auto W_ = [&](int l) { return (const void*)(p->d_w + p->off_w[l]); };
            auto B_ = [&](int l) { return (const float*)(p->d_w + p->off_b[l]); };
            auto single = [&](int l) -> int {      // layer l (1..6, 0-based) as a stand-alone MFMA kernel
                ProfScope ps(p, l, st);
                ZdceConvLaunch a;
                memset(&a, 0, sizeof(a));
                a.src0 = x[s0i[l]]; a.src1 = s1i[l] >= 0 ? x[s1i[l]] : nullptr;
                a.wpk = W_(l); a.bias = B_(l);
                a.N = N; a.H = H; a.W = W; a.n_begin = n0; a.n_count = g; a.rows_per_block = rb;
                a.nsrc = s1i[l] >= 0 ? 2 : 1; a.variant = p->cfg.variant;
                a.dt0 = p->xdt[s0i[l]]; a.dt1 = s1i[l] >= 0 ? p->xdt[s1i[l]] : a.dt0; a.dto = l < 6 ? p->xdt[l + 1] : a.dt0;
                if (l < 6) { a.epi = 0; a.dst = x[l + 1]; a.y_begin = lo(l + 1); a.y_end = hi(l + 1); }
                else { a.epi = 1; a.img = image; a.enhanced = enhanced; a.adjust = adjust; a.u8 = u8; a.y_begin = b0; a.y_end = b1; }
                return ie_zdce_launch_mfma(a, st);
            };
            auto fused = [&](int kind, int prof_layer) -> int {
                // diagnostic (wrong results): IE_ZDCE_ONLY=12|345|67 launches just that kernel, e.g. to time it without its neighbours' power draw
                static const int only = getenv("IE_ZDCE_ONLY") ? atoi(getenv("IE_ZDCE_ONLY")) : 0;
                if (only && only != kind && !(only == 67 && kind == 670)) return IE_OK;    // (67 selects either form of the last kernel)
                ProfScope ps(p, prof_layer, st);
                ZdceFusedLaunch a;
                memset(&a, 0, sizeof(a));
                a.kind = kind; a.N = N; a.H = H; a.W = W; a.n_begin = n0; a.n_count = g; a.rows_per_block = rb;
                a.img = image;
                a.stamps = g_ie_stamps ? (void*)((unsigned long long*)g_ie_stamps + (size_t)(kind == 12 ? 0 : (kind == 45 || kind == 345) ? 1 : 2) * (1u << 20)) : nullptr;
                if (kind == 12) {
                    a.a_w = p->d_w + p->off_w1m; a.a_bias = B_(0); a.b_w = p->d_w + p->off_w27; a.b_bias = B_(1); a.b2_bias = (const float*)(p->d_w + p->off_b7p);
                    a.a_dst = x[1]; a.b_dst = x[2]; a.y_begin = lo(2); a.y_end = hi(2);   // x[1] holds P = W7a*x1 + bias7, not x1
                    a.dt_h = a.dt_a0 = a.dt_a1 = p->xdt[1]; a.dt_o = p->xdt[2];            // (P takes x2's element type = x6's)
                } else if (kind == 45) {
                    a.a_src0 = x[3]; a.a_w = W_(3); a.a_bias = B_(3); a.b_w = W_(4); a.b_bias = B_(4);
                    a.b_dst = x[5]; a.y_begin = lo(5); a.y_end = hi(5);
                    a.dt_a0 = a.dt_a1 = p->xdt[3]; a.dt_h = p->xdt[4]; a.dt_o = p->xdt[5];
                } else if (kind == 345) {
                    a.a_src0 = x[2]; a.a_w = W_(2); a.a_bias = B_(2); a.a2_w = W_(3); a.a2_bias = B_(3); a.b_w = W_(4); a.b_bias = B_(4);
                    a.b_dst = x[5]; a.y_begin = lo(5); a.y_end = hi(5);
                    a.dt_a0 = p->xdt[2]; a.dt_a1 = p->xdt[3]; a.dt_h = p->xdt[4]; a.dt_o = p->xdt[5];
                } else {
                    a.a_src0 = x[2]; a.a_src1 = x[5]; a.b_src = x[1]; a.a_w = W_(5); a.a_bias = B_(5); a.b_w = p->d_w + p->off_w7b; a.b_bias = B_(6);
                    a.enhanced = enhanced; a.adjust = adjust; a.u8 = u8; a.y_begin = b0; a.y_end = b1;
                    a.dt_a0 = p->xdt[2]; a.dt_a1 = p->xdt[5]; a.dt_h = a.dt_o = p->xdt[6];
                }
                return ie_zdce_launch_fused(a, st);
            };
            int rc = IE_OK;
            if (p->cfg.variant & IE_ZDCE_VAR_UNFUSED) {
                {
                    ProfScope ps(p, 0, st);
                    rc = ie_zdce_launch_conv1(image, (const float*)W_(0), B_(0), x[1], p->xdt[1], N, H, W, lo(1), hi(1), n0, g, st);
                }
                for (int l = 1; l < NL && rc == IE_OK; ++l) rc = single(l);
            } else if (!(p->cfg.variant & (IE_ZDCE_VAR_P_SCHEDULE | IE_ZDCE_VAR_NO_K345 | IE_ZDCE_VAR_K67_LEGACY))) {
                // round-4 schedule: K12X (image -> x2), K345 (x2 -> x5), K67X (image, x2, x5 -> outputs); x1 / x3 / x4 / x6 never reach HBM and P does not exist
                static const int only = getenv("IE_ZDCE_ONLY") ? atoi(getenv("IE_ZDCE_ONLY")) : 0;      // diagnostic (wrong results): time one kernel alone
                auto xk = [&](int kind, int prof_layer) -> int {
                    if (only && only != kind) return IE_OK;
                    ProfScope ps(p, prof_layer, st);
                    ZdceXLaunch a;
                    memset(&a, 0, sizeof(a));
                    a.kind = kind; a.N = N; a.H = H; a.W = W; a.n_begin = n0; a.n_count = g; a.rows_per_block = rb;
                    a.img = image; a.x2 = x[2]; a.x5 = x[5];
                    a.w1 = p->d_w + p->off_w1x; a.w1q = p->d_w + p->off_w1q; a.b1 = B_(0); a.w2 = W_(1); a.b2 = B_(1); a.w6 = W_(5); a.b6 = B_(5);
                    a.w7 = p->d_w + p->off_w7x; a.b7 = (const float*)(p->d_w + p->off_b7x);
                    a.w3 = W_(2); a.b3 = B_(2); a.w4 = W_(3); a.b4 = B_(3); a.w5 = W_(4); a.b5 = B_(4);
                    a.dt_h = p->xdt[1]; a.dt_a0 = p->xdt[2]; a.dt_a1 = p->xdt[5]; a.dt_3 = p->xdt[3];
                    a.stamps = g_ie_stamps ? (void*)((unsigned long long*)g_ie_stamps + (size_t)(kind == 12 ? 0 : 2) * (1u << 20)) : nullptr;
                    if (kind == 12) { a.y_begin = lo(2); a.y_end = hi(2); }
                    else if (kind == 345) { a.y_begin = lo(5); a.y_end = hi(5); }
                    else { a.enhanced = enhanced; a.adjust = adjust; a.u8 = u8; a.y_begin = b0; a.y_end = b1; }
                    return ie_zdce_launch_x(a, st);
                };
                rc = xk(12, 1);
                if (rc == IE_OK) rc = (p->cfg.variant & IE_ZDCE_VAR_K345_LEGACY) || p->xdt[3] != p->xdt[4] || p->xdt[3] != p->xdt[5] ? fused(345, 4) : xk(345, 4);
                if (rc == IE_OK) rc = xk(67, 6);
            } else {
                // round-3 schedule (IE_ZDCE_VAR_P_SCHEDULE): K12 also writes P = W7a * x1 + b7 into x1's slot, then conv3, K45, K67
                rc = fused(12, 1);
                if (p->cfg.variant & IE_ZDCE_VAR_NO_K345) {
                    if (rc == IE_OK) rc = single(2);
                    if (rc == IE_OK) rc = fused(45, 4);
                } else if (rc == IE_OK) {
                    rc = fused(345, 4);        // e_conv3 + e_conv4 + e_conv5: x3 and x4 never reach HBM
                }
                if (rc == IE_OK) rc = fused((p->cfg.variant & IE_ZDCE_VAR_K67_LEGACY) ? 67 : 670, 6);
            }
            if (rc) return rc;
        }
    }
    return IE_OK;
}

IE_EXPORT int ie_zdce_forward(ie_zdce_plan* p, const float* image, float* enhanced, float* adjust, uint8_t* u8, int N,
                              int H, int W, void* stream) {
    if (!p || !image) IE_FAIL(IE_ERR_INVALID, "ie_zdce_forward: null plan/image");
    if (!p->finalized) IE_FAIL(IE_ERR_STATE, "ie_zdce_forward: plan not finalized (set 7 layers, then ie_zdce_finalize)");
    if (N <= 0 || H <= 0 || W <= 0) IE_FAIL(IE_ERR_INVALID, "ie_zdce_forward: bad shape N=%d H=%d W=%d", N, H, W);
    if (!enhanced && !adjust && !u8) IE_FAIL(IE_ERR_INVALID, "ie_zdce_forward: no output requested");
    int g, bd, rb;
    pick_schedule(p, N, H, W, &g, &bd, &rb);
    if (p->cap_frames < g || p->cap_H != H || p->cap_W != W || ws_per_frame(p, H, W) * g + 1024 > p->ws_bytes) {
        int rc = ie_zdce_reserve(p, N, H, W);
        if (rc) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    p->last_N = N; p->last_H = H; p->last_W = W;
    p->last_whole = (g >= N);
    if (p->cfg.compute_dtype == IE_F32) return forward_f32(p, image, enhanced, adjust, u8, N, H, W, st, g);
    return forward_mfma(p, image, enhanced, adjust, u8, N, H, W, st, g, bd, rb);
}

// uint8 frames in (what the reference's loader produces: HWC uint8, core/image/io.py:82-89), uint8 / fp32 frames out: the
// /255 normalisation (to_image_tensor, core/image/utils.py:801-847) runs group by group into a plan-owned NCHW fp32 buffer that the
// first kernel reads straight away (12 B/px written and read once, L2 / Infinity-Cache resident for one group), the
// save_image quantisation is fused into the last kernel.
IE_EXPORT int ie_zdce_forward_u8(ie_zdce_plan* p, const uint8_t* image_u8_nhwc, float* enhanced, float* adjust, uint8_t* u8, int N, int H, int W,
                                 void* stream) {
    if (!p || !image_u8_nhwc) IE_FAIL(IE_ERR_INVALID, "ie_zdce_forward_u8: null plan/image");
    if (N <= 0 || H <= 0 || W <= 0) IE_FAIL(IE_ERR_INVALID, "ie_zdce_forward_u8: bad shape N=%d H=%d W=%d", N, H, W);
    int g, bd, rb;
    pick_schedule(p, N, H, W, &g, &bd, &rb);
    const size_t plane3 = (size_t)3 * H * W, need = plane3 * g * sizeof(float);
    if (need > p->img_ws_bytes) {
        IE_HIP_CHECK(hipSetDevice(p->cfg.device));
        if (p->img_ws) { IE_HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(p->img_ws); p->img_ws = nullptr; p->img_ws_bytes = 0; }
        hipError_t e = hipMalloc(&p->img_ws, need);
        if (e != hipSuccess) { ie_set_error("ie_zdce_forward_u8: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e)); return IE_ERR_NOMEM; }
        p->img_ws_bytes = need;
    }
    for (int n0 = 0; n0 < N; n0 += g) {
        const int gn = (N - n0) < g ? (N - n0) : g;
        int rc = ie_u8_nhwc_to_nchw_f32(image_u8_nhwc + (size_t)n0 * plane3, p->img_ws, gn, 3, H, W, stream);
        if (rc) return rc;
        rc = ie_zdce_forward(p, p->img_ws, enhanced ? enhanced + (size_t)n0 * plane3 : nullptr, adjust ? adjust + (size_t)n0 * 24 * H * W : nullptr,
                             u8 ? u8 + (size_t)n0 * plane3 : nullptr, gn, H, W, stream);
        if (rc) return rc;
    }
    p->last_N = N;
    p->last_whole = (g >= N);
    return IE_OK;
}

IE_EXPORT int ie_zdce_get_feature(ie_zdce_plan* p, int idx, float* dst, void* stream) {
    if (!p || !dst || idx < 1 || idx > 6) IE_FAIL(IE_ERR_INVALID, "ie_zdce_get_feature: bad args");
    if (!p->last_whole) IE_FAIL(IE_ERR_STATE, "ie_zdce_get_feature: last forward did not keep the whole batch in one group");
    const int N = p->last_N, H = p->last_H, W = p->last_W;
    int g, bd, rb;
    pick_schedule(p, N, H, W, &g, &bd, &rb);
    if (p->cfg.compute_dtype == IE_F32) {
        const float* x0 = (const float*)p->ws;
        const float* xk = x0 + frame_elems(H, W, 4) * g + frame_elems(H, W, 32) * g * (idx - 1);
        return ie_nhwc_to_nchw_f32(xk, IE_F32, dst, N, 32, H, W, 32, stream);
    }
    const size_t fb = frame_elems(H, W, 32) * elem_size(p);
    return ie_nhwc_to_nchw_f32(p->ws + fb * g * (idx - 1), (ie_dtype)p->xdt[idx], dst, N, 32, H, W, 32, stream);
}

IE_EXPORT int ie_zdce_destroy(ie_zdce_plan* p) {
    if (!p) return IE_OK;
    (void)hipSetDevice(p->cfg.device);
    (void)hipDeviceSynchronize();
    if (p->d_w) (void)hipFree(p->d_w);
    if (p->ws) (void)hipFree(p->ws);
    if (p->img_ws) (void)hipFree(p->img_ws);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    delete p;
    return IE_OK;
}

IE_EXPORT int ie_zdce_time_forward(ie_zdce_plan* p, const float* image, float* enhanced, float* adjust, uint8_t* u8,
                                   int N, int H, int W, int iters, void* stream, float* ms_total) {
    if (!p || iters <= 0 || !ms_total) IE_FAIL(IE_ERR_INVALID, "ie_zdce_time_forward: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t a, b;
    IE_HIP_CHECK(hipEventCreate(&a));
    IE_HIP_CHECK(hipEventCreate(&b));
    int rc = IE_OK;
    IE_HIP_CHECK(hipEventRecord(a, st));
    for (int i = 0; i < iters && rc == IE_OK; ++i) rc = ie_zdce_forward(p, image, enhanced, adjust, u8, N, H, W, stream);
    IE_HIP_CHECK(hipEventRecord(b, st));
    IE_HIP_CHECK(hipEventSynchronize(b));
    if (rc == IE_OK) IE_HIP_CHECK(hipEventElapsedTime(ms_total, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return rc;
}

IE_EXPORT int ie_zdce_profile_forward(ie_zdce_plan* p, const float* image, float* enhanced, float* adjust, uint8_t* u8,
                                      int N, int H, int W, void* stream, float* ms, const char** names, int* launches,
                                      int cap, int* n) {
    if (!p || !ms || !names || !n || cap < NL) IE_FAIL(IE_ERR_INVALID, "ie_zdce_profile_forward: bad args (cap >= 7)");
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    p->ev.clear(); p->ev_layer.clear();
    p->prof = true;
    int rc = ie_zdce_forward(p, image, enhanced, adjust, u8, N, H, W, stream);
    p->prof = false;
    if (rc) return rc;
    IE_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    const bool fused = p->cfg.compute_dtype != IE_F32 && !(p->cfg.variant & IE_ZDCE_VAR_UNFUSED);
    const bool k345 = fused && !(p->cfg.variant & IE_ZDCE_VAR_NO_K345);
    for (int l = 0; l < NL; ++l) { ms[l] = 0.f; names[l] = k345 ? kFused3Names[l] : (fused ? kFusedNames[l] : kLayerNames[l]); if (launches) launches[l] = 0; }
    if (fused && !(p->cfg.variant & IE_ZDCE_VAR_K67_LEGACY)) names[6] = "zdce_k67n_conv6_conv7_curve";
    if (fused && !(p->cfg.variant & (IE_ZDCE_VAR_P_SCHEDULE | IE_ZDCE_VAR_NO_K345 | IE_ZDCE_VAR_K67_LEGACY))) {
        names[1] = "zdce_k12x_conv1_conv2"; names[6] = "zdce_k67x_conv1_conv6_conv7_curve";
        if (!(p->cfg.variant & IE_ZDCE_VAR_K345_LEGACY) && p->xdt[3] == p->xdt[4] && p->xdt[3] == p->xdt[5]) names[4] = "zdce_k345x_conv3_conv4_conv5";
    }
    for (size_t i = 0; i < p->ev_layer.size(); ++i) {
        float t = 0.f;
        IE_HIP_CHECK(hipEventElapsedTime(&t, p->ev[2 * i], p->ev[2 * i + 1]));
        ms[p->ev_layer[i]] += t;
        if (launches) launches[p->ev_layer[i]] += 1;
    }
    *n = NL;
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    p->ev.clear(); p->ev_layer.clear();
    return IE_OK;
}
