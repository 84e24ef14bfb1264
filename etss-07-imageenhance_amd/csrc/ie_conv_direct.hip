// Generic fp32 NHWC conv2d on the VALU: the parity-mode / any-shape path.
//
// Replaces torch.nn.Conv2d as used by every generator on the path
// (src/mon/nn/modules/conv/base.py:45 aliases it): k in {1,3,4,9}, stride {1,2}, zero or
// replicate padding (mon_extra/.../retinexnet/model.py:26-66), dense or depthwise
// (nn/modules/conv/dsconv.py:122-134), optional second source = virtual channel concat.
//
// Mapping: lane = output pixel (64 consecutive pixels of one frame per wave), wave = 8 output
// channels; the weight address is therefore wave-uniform and goes through the scalar cache,
// the input pixel is one 16-byte load per 4 input channels.  fp32 FMA chain per output,
// k-order (ky, kx, source, cin) -- within ~1e-6 of any other fp32 summation order.
#include "ie_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
    case IE_ACT_RELU: return fmaxf(v, 0.f);
    case IE_ACT_LRELU: return v > 0.f ? v : v * slope;
    case IE_ACT_TANH: return tanhf(v);
    case IE_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
    }
}

struct DirectParams {
    const float* src0; const float* src1; const float* w; const float* bias; float* dst;
    int N, H, W, C0, C1, Cout, kh, kw, stride, pad, pad_mode, act, Ho, Wo;
    float slope;
};

constexpr int CT = 8;  // output channels per thread

__global__ __launch_bounds__(256) void conv2d_direct_kernel(DirectParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.z;
    const int opix = blockIdx.x * 64 + lane;
    const int co0 = (blockIdx.y * 4 + wave) * CT;
    if (co0 >= p.Cout) return;  // wave-uniform
    const bool live = opix < p.Ho * p.Wo;
    const int ho = live ? opix / p.Wo : 0, wo = live ? opix % p.Wo : 0;
    const int Cin = p.C0 + p.C1;
    const int nco = min(CT, p.Cout - co0);

    float acc[CT];
#pragma unroll
    for (int o = 0; o < CT; ++o) acc[o] = (p.bias && o < nco) ? p.bias[co0 + o] : 0.f;

    for (int i = 0; i < p.kh; ++i)
        for (int j = 0; j < p.kw; ++j) {
            int hi = ho * p.stride + i - p.pad, wi = wo * p.stride + j - p.pad;
            bool inb = true;
            if (p.pad_mode == IE_PAD_REPLICATE) {
                hi = min(max(hi, 0), p.H - 1);
                wi = min(max(wi, 0), p.W - 1);
            } else {
                inb = hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
            }
            const size_t pix = ((size_t)n * p.H + (inb ? hi : 0)) * p.W + (inb ? wi : 0);
            const float* wtap = p.w + (size_t)(i * p.kw + j) * Cin * p.Cout + co0;
            for (int s = 0; s < 2; ++s) {
                const int Cs = s ? p.C1 : p.C0;
                if (Cs == 0) continue;
                const float* sp = (s ? p.src1 : p.src0) + pix * Cs;
                const float* ws = wtap + (size_t)(s ? p.C0 : 0) * p.Cout;
                int c = 0;
                if ((Cs & 3) == 0) {
                    for (; c < Cs; c += 4) {
                        ie_f32x4 v = inb ? *(const ie_f32x4*)(sp + c) : ie_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float* wr = ws + (size_t)(c + k) * p.Cout;
                            if (nco == CT) {
#pragma unroll
                                for (int o = 0; o < CT; ++o) acc[o] = fmaf(v[k], wr[o], acc[o]);
                            } else {
#pragma unroll
                                for (int o = 0; o < CT; ++o) if (o < nco) acc[o] = fmaf(v[k], wr[o], acc[o]);
                            }
                        }
                    }
                } else {
                    for (; c < Cs; ++c) {
                        const float v = inb ? sp[c] : 0.f;
                        const float* wr = ws + (size_t)c * p.Cout;
#pragma unroll
                        for (int o = 0; o < CT; ++o) if (o < nco) acc[o] = fmaf(v, wr[o], acc[o]);
                    }
                }
            }
        }
    if (!live) return;
    float* d = p.dst + (((size_t)n * p.Ho + ho) * p.Wo + wo) * p.Cout + co0;
#pragma unroll
    for (int o = 0; o < CT; ++o) if (o < nco) d[o] = apply_act(acc[o], p.act, p.slope);
}

// depthwise: thread = one output pixel x one channel quad; weights HWIO with I = 1: [kh][kw][C]
__global__ __launch_bounds__(256) void conv2d_depthwise_kernel(DirectParams p) {
    const int C = p.C0;
    const int cq = (C + 3) / 4;
    const size_t total = (size_t)p.Ho * p.Wo * cq;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.z;
    if (t >= total) return;
    const int q = (int)(t % cq);
    const int opix = (int)(t / cq);
    const int ho = opix / p.Wo, wo = opix % p.Wo;
    const int c0 = q * 4, nc = min(4, C - c0);
    float acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = (p.bias && k < nc) ? p.bias[c0 + k] : 0.f;
    for (int i = 0; i < p.kh; ++i)
        for (int j = 0; j < p.kw; ++j) {
            int hi = ho * p.stride + i - p.pad, wi = wo * p.stride + j - p.pad;
            if (p.pad_mode == IE_PAD_REPLICATE) {
                hi = min(max(hi, 0), p.H - 1);
                wi = min(max(wi, 0), p.W - 1);
            } else if (hi < 0 || hi >= p.H || wi < 0 || wi >= p.W) {
                continue;
            }
            const float* sp = p.src0 + (((size_t)n * p.H + hi) * p.W + wi) * C + c0;
            const float* wp = p.w + (size_t)(i * p.kw + j) * C + c0;
            for (int k = 0; k < nc; ++k) acc[k] = fmaf(sp[k], wp[k], acc[k]);
        }
    float* d = p.dst + (((size_t)n * p.Ho + ho) * p.Wo + wo) * C + c0;
    for (int k = 0; k < nc; ++k) d[k] = apply_act(acc[k], p.act, p.slope);
}

__global__ void repack_oihw_to_hwio(const float* __restrict__ src, float* __restrict__ dst, int Co, int Ci, int kh, int kw) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = Co * Ci * kh * kw;
    if (t >= total) return;
    // t indexes dst [kh][kw][Ci][Co]
    const int o = t % Co; int r = t / Co;
    const int c = r % Ci; r /= Ci;
    const int j = r % kw; const int i = r / kw;
    dst[t] = src[((o * Ci + c) * kh + i) * kw + j];
}

}  // namespace

IE_EXPORT int ie_conv2d_nhwc_f32(ie_conv2d_desc* d, const float* src0, const float* src1, const float* w,
                                 const float* bias, float* dst, void* stream) {
    if (!d || !src0 || !w || !dst) IE_FAIL(IE_ERR_INVALID, "ie_conv2d_nhwc_f32: null pointer");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C0 <= 0 || d->C1 < 0 || d->Cout <= 0 || d->kh <= 0 || d->kw <= 0 ||
        d->stride <= 0 || d->pad < 0)
        IE_FAIL(IE_ERR_INVALID, "ie_conv2d_nhwc_f32: bad shape N=%d H=%d W=%d C0=%d C1=%d Cout=%d k=%dx%d s=%d p=%d",
                d->N, d->H, d->W, d->C0, d->C1, d->Cout, d->kh, d->kw, d->stride, d->pad);
    if (d->C1 > 0 && !src1) IE_FAIL(IE_ERR_INVALID, "ie_conv2d_nhwc_f32: C1 > 0 but src1 is null");
    const bool depthwise = d->groups != 1;
    if (depthwise && !(d->groups == d->C0 && d->Cout == d->C0 && d->C1 == 0))
        IE_FAIL(IE_ERR_UNSUPPORTED, "ie_conv2d_nhwc_f32: groups=%d only supports depthwise (groups == C0 == Cout)", d->groups);
    d->Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    d->Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    if (d->Ho <= 0 || d->Wo <= 0) IE_FAIL(IE_ERR_INVALID, "ie_conv2d_nhwc_f32: empty output");
    DirectParams p;
    p.src0 = src0; p.src1 = src1; p.w = w; p.bias = bias; p.dst = dst;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C0 = d->C0; p.C1 = d->C1; p.Cout = d->Cout; p.kh = d->kh; p.kw = d->kw;
    p.stride = d->stride; p.pad = d->pad; p.pad_mode = d->pad_mode; p.act = d->act; p.Ho = d->Ho; p.Wo = d->Wo;
    p.slope = d->slope;
    hipStream_t st = (hipStream_t)stream;
    if (depthwise) {
        const size_t total = (size_t)d->Ho * d->Wo * ((d->C0 + 3) / 4);
        dim3 grid((unsigned)((total + 255) / 256), 1, d->N);
        hipLaunchKernelGGL(conv2d_depthwise_kernel, grid, dim3(256), 0, st, p);
    } else {
        dim3 grid(ie_cdiv(d->Ho * d->Wo, 64), ie_cdiv(d->Cout, 4 * CT), d->N);
        hipLaunchKernelGGL(conv2d_direct_kernel, grid, dim3(256), 0, st, p);
    }
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_pack_weight_hwio_f32(const float* w_oihw_host, int Cout, int Cin, int kh, int kw, float* dst_dev, void* stream) {
    if (!w_oihw_host || !dst_dev || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0) IE_FAIL(IE_ERR_INVALID, "ie_pack_weight_hwio_f32: bad args");
    const size_t n = (size_t)Cout * Cin * kh * kw;
    float* tmp = nullptr;
    IE_HIP_CHECK(hipMalloc(&tmp, n * sizeof(float)));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(tmp, w_oihw_host, n * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(repack_oihw_to_hwio, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tmp, dst_dev, Cout, Cin, kh, kw);
        e = hipGetLastError();
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e != hipSuccess) IE_FAIL(IE_ERR_HIP, "ie_pack_weight_hwio_f32: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) IE_FAIL(IE_ERR_HIP, "ie_pack_weight_hwio_f32: %s", hipGetErrorString(e2));
    return IE_OK;
}
