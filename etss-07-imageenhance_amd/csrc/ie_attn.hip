// Per-(sample, channel) statistics and the layers built on them:
//   LearnableInstanceNorm2d  src/mon/nn/modules/normalization.py:177-224   y = IN(x) r + x (1 - r), r per channel
//   SimAM                    src/mon/nn/modules/attention.py:1156-1182     y = x sigmoid((x - mu)^2 / (4 (v + lambda)) + 0.5)
//   channel attention        src/mon/vision/enhance/multitask/mprnet/mprnet.py:51-82 (global average pool -> 1x1 -> ReLU -> 1x1 ->
//                            sigmoid -> per-channel scale)
// fp32 NHWC.  Statistics are accumulated in fp64 (one atomic pair per workgroup and channel), the apply passes are streaming.
#include "ie_common.h"

namespace {

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

// stats[n][c] = {sum, sum of squares} over H*W.  Workgroup = 256 threads over a pixel chunk of one sample; thread = (pixel lane, channel).
template <typename T>
__global__ void channel_stats_kernel(const T* __restrict__ x, double* __restrict__ stats, int HW, int C, int chunk) {
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, HW);
    const int tpc = 256 / min(C, 256);                 // pixel lanes per channel when C < 256
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int cn = min(C - c0, 256);
        const int lanes = 256 / cn;
        const int c = threadIdx.x % cn, pl = threadIdx.x / cn;
        double s = 0.0, ss = 0.0;
        if (pl < lanes) {
            for (int p = p0 + pl; p < p1; p += lanes) {
                const float v = (float)x[((size_t)n * HW + p) * C + c0 + c];
                s += v; ss += (double)v * v;
            }
            atomicAdd(&stats[((size_t)n * C + c0 + c) * 2 + 0], s);
            atomicAdd(&stats[((size_t)n * C + c0 + c) * 2 + 1], ss);
        }
    }
    (void)tpc;
}

// 16-bit activations, C % 8 == 0: thread = 8 channels (16 bytes) of a pixel, fp32 partial sums over the thread's pixels, one LDS fp64
// atomic per partial, then one global fp64 atomic per channel and workgroup (the scalar kernel above ran at 0.8 TB/s on MPRNet's gates)
template <typename T>
__global__ void channel_stats_v8_kernel(const T* __restrict__ x, double* __restrict__ stats, int HW, int C, int chunk) {
    extern __shared__ double acc[];                    // [C][2]
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, HW);
    const int cg = C >> 3;                             // channel groups
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) acc[i] = 0.0;
    __syncthreads();
    for (int g0 = 0; g0 < cg; g0 += 256) {             // C <= 2048 per pass
        const int gn = min(cg - g0, 256);
        const int lanes = 256 / gn;
        const int g = threadIdx.x % gn, pl = threadIdx.x / gn;
        if (pl < lanes) {
            float s[8], ss[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] = 0.f; ss[k] = 0.f; }
            typedef __attribute__((ext_vector_type(8))) T t8;
            for (int p = p0 + pl; p < p1; p += lanes) {
                const t8 v = *(const t8*)(x + ((size_t)n * HW + p) * C + (size_t)(g0 + g) * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float f = (float)v[k]; s[k] += f; ss[k] = fmaf(f, f, ss[k]); }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                atomicAdd(&acc[((g0 + g) * 8 + k) * 2 + 0], (double)s[k]);
                atomicAdd(&acc[((g0 + g) * 8 + k) * 2 + 1], (double)ss[k]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&stats[(size_t)n * C * 2 + i], acc[i]);
}

// LearnableInstanceNorm2d coefficients: y = x * scale + shift with
//   scale = gamma r / sqrt(var + eps) + (1 - r),  shift = (beta - mean gamma / sqrt(var + eps)) r       (biased variance, like F.instance_norm)
__global__ void lin_coeff_kernel(const double* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ r, float* __restrict__ scale, float* __restrict__ shift, int NC, int C, double inv_hw,
                                 float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    const int c = i % C;
    const double mean = stats[2 * i] * inv_hw;
    double var = stats[2 * i + 1] * inv_hw - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? gamma[c] : 1.0, b = beta ? beta[c] : 0.0, rr = r ? r[c] : 1.0;
    scale[i] = (float)(g * rr * inv + (1.0 - rr));
    shift[i] = (float)((b - mean * g * inv) * rr);
}
// mean[n][c] = sum / HW (global average pool)
__global__ void stats_mean_kernel(const double* __restrict__ stats, float* __restrict__ mean, int NC, double inv_hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NC) mean[i] = (float)(stats[2 * i] * inv_hw);
}

// y = act(x * scale[n][c] + shift[n][c]); shift may be NULL
template <typename T>
__global__ void channel_affine_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, const T* __restrict__ res,
                                      T* __restrict__ y, size_t hwc, int C, int act, float slope, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / hwc;
    const int c = (int)(i % C);
    const float v = (float)x[i] * scale[n * C + c] + (shift ? shift[n * C + c] : 0.f);
    y[i] = (T)(ie_act_apply(v, act, slope) + (res ? (float)res[i] : 0.f));
}

template <typename T>
__global__ void channel_affine_v8_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                         const T* __restrict__ res, T* __restrict__ y, size_t hwc8, int C8, int act, float slope, size_t total8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one 8-channel group
    if (i >= total8) return;
    const size_t n = i / hwc8;
    const int c0 = (int)(i % C8) * 8;
    typedef __attribute__((ext_vector_type(8))) T t8;
    const t8 v = *(const t8*)(x + i * 8);
    const float* sc = scale + n * (size_t)C8 * 8 + c0;
    const float* sh = shift ? shift + n * (size_t)C8 * 8 + c0 : nullptr;
    t8 o, r;
    if (res) r = *(const t8*)(res + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (T)(ie_act_apply((float)v[k] * sc[k] + (sh ? sh[k] : 0.f), act, slope) + (res ? (float)r[k] : 0.f));
    *(t8*)(y + i * 8) = o;
}

// SimAM: d = (x - mu)^2; v = sum d / (HW - 1); y = x * sigmoid(d / (4 (v + lambda)) + 0.5)
template <typename T>
__global__ void simam_kernel(const T* __restrict__ x, const double* __restrict__ stats, T* __restrict__ y, size_t hwc, int C, int HW,
                             float lambda, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / hwc;
    const int c = (int)(i % C);
    const double s = stats[(n * C + c) * 2], ss = stats[(n * C + c) * 2 + 1];
    const double mu = s / HW;
    double sd = ss - s * mu;                        // sum (x - mu)^2
    sd = sd > 0.0 ? sd : 0.0;
    const float v = (float)(sd / (HW - 1));
    const float xv = (float)x[i];
    const float d = (xv - (float)mu) * (xv - (float)mu);
    const float e = d / (4.f * (v + lambda)) + 0.5f;
    y[i] = (T)(xv * (1.0f / (1.0f + expf(-e))));
}

}  // namespace

template <typename T>
static int channel_stats_impl(const void* x, double* stats, int N, int H, int W, int C, void* stream) {
    if (!x || !stats || N <= 0 || H <= 0 || W <= 0 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_channel_stats_nhwc: bad args");
    hipStream_t st = (hipStream_t)stream;
    IE_HIP_CHECK(hipMemsetAsync(stats, 0, sizeof(double) * 2 * (size_t)N * C, st));
    const int HW = H * W;
    if constexpr (sizeof(T) == 2) {
        if ((C & 7) == 0 && C <= 2048) {
            int chunks = (1024 + N - 1) / N;                 // about four workgroups per CU over the batch
            if (chunks > (HW + 255) / 256) chunks = (HW + 255) / 256;
            if (chunks < 1) chunks = 1;
            const int chunk = (HW + chunks - 1) / chunks;
            hipLaunchKernelGGL(channel_stats_v8_kernel<T>, dim3((unsigned)((HW + chunk - 1) / chunk), (unsigned)N), dim3(256), sizeof(double) * 2 * C, st,
                               (const T*)x, stats, HW, C, chunk);
            IE_LAUNCH_CHECK();
            return IE_OK;
        }
    }
    int chunks = (HW + 4095) / 4096;
    if (chunks > 1024) chunks = 1024;
    const int chunk = (HW + chunks - 1) / chunks;
    hipLaunchKernelGGL(channel_stats_kernel<T>, dim3((unsigned)((HW + chunk - 1) / chunk), (unsigned)N), dim3(256), 0, st, (const T*)x, stats, HW, C, chunk);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_channel_stats_nhwc_f32(const float* x, double* stats, int N, int H, int W, int C, void* stream) {
    return channel_stats_impl<float>(x, stats, N, H, W, C, stream);
}
IE_EXPORT int ie_channel_stats_nhwc(const void* x, int dtype, double* stats, int N, int H, int W, int C, void* stream) {
    switch (dtype) {
    case IE_F32: return channel_stats_impl<float>(x, stats, N, H, W, C, stream);
    case IE_F16: return channel_stats_impl<_Float16>(x, stats, N, H, W, C, stream);
    case IE_BF16: return channel_stats_impl<__bf16>(x, stats, N, H, W, C, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_channel_stats_nhwc: bad dtype");
    }
}
IE_EXPORT int ie_lin_coeffs_f32(const double* stats, const float* gamma, const float* beta, const float* r, float* scale, float* shift, int N, int C,
                                int HW, float eps, void* stream) {
    if (!stats || !scale || !shift || N <= 0 || C <= 0 || HW <= 0) IE_FAIL(IE_ERR_INVALID, "ie_lin_coeffs_f32: bad args");
    hipLaunchKernelGGL(lin_coeff_kernel, dim3(nblk((size_t)N * C)), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, r, scale, shift, N * C, C,
                       1.0 / HW, eps);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_stats_mean_f32(const double* stats, float* mean, int N, int C, int HW, void* stream) {
    if (!stats || !mean || N <= 0 || C <= 0 || HW <= 0) IE_FAIL(IE_ERR_INVALID, "ie_stats_mean_f32: bad args");
    hipLaunchKernelGGL(stats_mean_kernel, dim3(nblk((size_t)N * C)), dim3(256), 0, (hipStream_t)stream, stats, mean, N * C, 1.0 / HW);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
template <typename T>
static int channel_affine_impl(const void* x, const float* scale, const float* shift, const void* res, void* y, int N, int H, int W, int C, int act,
                               float slope, void* stream) {
    if (!x || !scale || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_channel_affine_nhwc: bad args");
    const size_t hwc = (size_t)H * W * C, total = hwc * N;
    if constexpr (sizeof(T) == 2) {
        if ((C & 7) == 0) {
            hipLaunchKernelGGL(channel_affine_v8_kernel<T>, dim3(nblk(total / 8)), dim3(256), 0, (hipStream_t)stream, (const T*)x, scale, shift, (const T*)res,
                               (T*)y, hwc / 8, C / 8, act, slope, total / 8);
            IE_LAUNCH_CHECK();
            return IE_OK;
        }
    }
    hipLaunchKernelGGL(channel_affine_kernel<T>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, scale, shift, (const T*)res, (T*)y, hwc,
                       C, act, slope, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_channel_affine_nhwc_f32(const float* x, const float* scale, const float* shift, float* y, int N, int H, int W, int C, int act,
                                         float slope, void* stream) {
    return channel_affine_impl<float>(x, scale, shift, nullptr, y, N, H, W, C, act, slope, stream);
}
IE_EXPORT int ie_channel_affine_nhwc(const void* x, const float* scale, const float* shift, void* y, int dtype, int N, int H, int W, int C, int act,
                                     float slope, void* stream) {
    return ie_channel_affine_add_nhwc(x, scale, shift, nullptr, y, dtype, N, H, W, C, act, slope, stream);
}
// act(x * scale[n, c] + shift[n, c]) + res: the channel-attention block's gate and its skip connection in one pass (mprnet.py:156-161:
// `y = self.CA(y); y += x`), one rounding of the fp32 sum
IE_EXPORT int ie_channel_affine_add_nhwc(const void* x, const float* scale, const float* shift, const void* res, void* y, int dtype, int N, int H, int W,
                                         int C, int act, float slope, void* stream) {
    switch (dtype) {
    case IE_F32: return channel_affine_impl<float>(x, scale, shift, res, y, N, H, W, C, act, slope, stream);
    case IE_F16: return channel_affine_impl<_Float16>(x, scale, shift, res, y, N, H, W, C, act, slope, stream);
    case IE_BF16: return channel_affine_impl<__bf16>(x, scale, shift, res, y, N, H, W, C, act, slope, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_channel_affine_nhwc: bad dtype");
    }
}
template <typename T>
static int simam_impl(const void* x, const double* stats, void* y, int N, int H, int W, int C, float lambda, void* stream) {
    if (!x || !stats || !y || N <= 0 || H * W <= 1 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_simam_nhwc: bad args");
    const size_t hwc = (size_t)H * W * C, total = hwc * N;
    hipLaunchKernelGGL(simam_kernel<T>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, stats, (T*)y, hwc, C, H * W, lambda, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_simam_nhwc_f32(const float* x, const double* stats, float* y, int N, int H, int W, int C, float lambda, void* stream) {
    return simam_impl<float>(x, stats, y, N, H, W, C, lambda, stream);
}
IE_EXPORT int ie_simam_nhwc(const void* x, const double* stats, void* y, int dtype, int N, int H, int W, int C, float lambda, void* stream) {
    switch (dtype) {
    case IE_F32: return simam_impl<float>(x, stats, y, N, H, W, C, lambda, stream);
    case IE_F16: return simam_impl<_Float16>(x, stats, y, N, H, W, C, lambda, stream);
    case IE_BF16: return simam_impl<__bf16>(x, stats, y, N, H, W, C, lambda, stream);
    default: IE_FAIL(IE_ERR_INVALID, "ie_simam_nhwc: bad dtype");
    }
}
