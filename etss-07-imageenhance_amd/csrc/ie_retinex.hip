// RetinexNet's decomposition / relighting losses and their gradients, fp32 NHWC:
//   src/mon_extra/vision/enhance/llie/retinexnet/model.py:199-238  (RetinexNet.forward: the loss terms)
//   model.py:240-260                                             (gradient / ave_gradient / smooth)
// Every entry point ADDS `weight * term` to a double accumulator and `weight * d term` to the gradient tensors it is given (callers zero
// them once per step), so one step's loss is a handful of calls on the same buffers.  Plus replicate padding as its own op
// (forward + adjoint): a replicate-padded convolution's backward pass is the zero-padding one on the explicitly padded input.
#include "ie_common.h"

namespace {

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }      // torch: d|x|/dx = sign(x), 0 at 0

inline unsigned nblk_capped(size_t n) { const unsigned b = nblk(n); return b < 2048u ? b : 2048u; }      // grid-stride kernels: one loss atomic per workgroup

__device__ __forceinline__ void block_add(double v, double* dst) {
    // wave reduction, the four waves through LDS, then ONE fp64 atomic per workgroup (256 threads).  One atomic per wave on one address serialised in L2:
    // 65 k of them made the L1 term of two 1080p frames a 2.3 ms kernel (its operands take 30 us to read) -- round 3 profile of the relight phase.
    __shared__ double part[4];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (part[0] + part[1]) + (part[2] + part[3]);
        if (t != 0.0) atomicAdd(dst, t);
    }
}

// loss += w * mean|R * I - T| over N*H*W*3;  dR += w sign(.) I / numel,  dI += w sum_c sign(.) R_c / numel       (model.py:216-221)
// R / dR have a pixel stride of rs floats, I / dI of is (3 and 1 for separate tensors, 4 and 4 for DecomNet's (R, L) output)
__global__ void recon_kernel(const float* __restrict__ R, int rs, const float* __restrict__ I, int is, const float* __restrict__ T,
                             float w_over_numel, float* __restrict__ dR, float* __restrict__ dI, double* __restrict__ loss, size_t npx) {
    double acc = 0.0;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (size_t)gridDim.x * blockDim.x) {
        const float i = I[is * p];
        float gi = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float r = R[rs * p + c], d = r * i - T[3 * p + c];
            acc += fabsf(d);
            const float s = sgn(d) * w_over_numel;
            if (dR) dR[rs * p + c] += s * i;
            gi += s * r;
        }
        if (dI) dI[is * p] += gi;
    }
    block_add(acc * (double)w_over_numel, loss);
}

// loss += w * mean|A - B| (B detached);  dA += w sign(A - B) / numel                                               (model.py:220)
// A, B, dA: C channels of pixels with a stride of st floats
__global__ void l1_detached_kernel(const float* __restrict__ A, const float* __restrict__ B, int st, int C, float w_over_numel, float* __restrict__ dA,
                                   double* __restrict__ loss, size_t n) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = (i / C) * st + (i % C);
        const float d = A[o] - B[o];
        acc += fabsf(d);
        if (dA) dA[o] += sgn(d) * w_over_numel;
    }
    block_add(acc * (double)w_over_numel, loss);
}

// smooth(I, R) = mean over the (H+1) x (W+1) grid of  |gx I| exp(-10 avg3x3 |gx Rg|) + |gy I| exp(-10 avg3x3 |gy Rg|)   (model.py:240-260)
//   gx f [i][j] = f[i][j] - f[i][j-1],  gy f [i][j] = f[i][j] - f[i-1][j]   (conv2d with the 2x2 kernels [[0,0],[-1,1]] and its transpose,
//   padding 1: zeros outside, so the grid has one more row and column than the image);  Rg = 0.299 R0 + 0.587 R1 + 0.114 R2;
//   avg3x3 = F.avg_pool2d(k 3, s 1, p 1): zero padded, always divided by 9.
__device__ __forceinline__ float at(const float* f, int H, int W, int i, int j) {
    return (i >= 0 && i < H && j >= 0 && j < W) ? f[(size_t)i * W + j] : 0.f;
}
// gray map of R and a dense copy of I (both may be strided views of one tensor)
__global__ void gray_kernel(const float* __restrict__ R, int rs, const float* __restrict__ I, int is, float* __restrict__ g, float* __restrict__ ic,
                            size_t npx) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < npx) {
        g[p] = 0.299f * R[rs * p] + 0.587f * R[rs * p + 1] + 0.114f * R[rs * p + 2];
        ic[p] = I[is * p];
    }
}
// grid maps: hx, hy = |gx Rg|, |gy Rg|
__global__ void smooth_h_kernel(const float* __restrict__ g, float* __restrict__ hx, float* __restrict__ hy, int H, int W) {
    const int n = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j > W) return;
    const float* gn = g + (size_t)n * H * W;
    const size_t o = ((size_t)n * (H + 1) + i) * (W + 1) + j;
    hx[o] = fabsf(at(gn, H, W, i, j) - at(gn, H, W, i, j - 1));
    hy[o] = fabsf(at(gn, H, W, i, j) - at(gn, H, W, i - 1, j));
}
__device__ __forceinline__ float avg9(const float* m, int H1, int W1, int i, int j) {
    float s = 0.f;
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
        for (int b = -1; b <= 1; ++b) s += at(m, H1, W1, i + a, j + b);
    return s * (1.f / 9.f);
}
// wx = exp(-10 avg hx), wy likewise; term = |gx I| wx + |gy I| wy;  qx = |gx I| (-10) wx = d term / d (avg hx), qy likewise
__global__ void smooth_term_kernel(const float* __restrict__ I, const float* __restrict__ hx, const float* __restrict__ hy, float* __restrict__ wx,
                                   float* __restrict__ wy, float* __restrict__ qx, float* __restrict__ qy, double* __restrict__ loss, float w_over_numel,
                                   int H, int W) {
    const int n = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (j <= W)
    for (int i = blockIdx.y; i <= H; i += gridDim.y) {          // rows strided over gridDim.y workgroups: one loss atomic per workgroup, not per row
        const float* In = I + (size_t)n * H * W;
        const size_t base = (size_t)n * (H + 1) * (W + 1), o = base + (size_t)i * (W + 1) + j;
        const float gix = fabsf(at(In, H, W, i, j) - at(In, H, W, i, j - 1)), giy = fabsf(at(In, H, W, i, j) - at(In, H, W, i - 1, j));
        const float ex = expf(-10.f * avg9(hx + base, H + 1, W + 1, i, j)), ey = expf(-10.f * avg9(hy + base, H + 1, W + 1, i, j));
        wx[o] = ex; wy[o] = ey;
        qx[o] = -10.f * gix * ex; qy[o] = -10.f * giy * ey;
        acc += (double)gix * ex + (double)giy * ey;
    }
    block_add(acc * (double)w_over_numel, loss);
}
// per image pixel (i, j): dI and dRg gathered from the grid points that touch it
__global__ void smooth_grad_kernel(const float* __restrict__ I, const float* __restrict__ g, const float* __restrict__ wx, const float* __restrict__ wy,
                                   const float* __restrict__ qx, const float* __restrict__ qy, float* __restrict__ dI, int is, float* __restrict__ dR,
                                   int rs, float w_over_numel, int H, int W) {
    const int n = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= W) return;
    const float* In = I + (size_t)n * H * W;
    const float* gn = g + (size_t)n * H * W;
    const size_t base = (size_t)n * (H + 1) * (W + 1);
    const int W1 = W + 1, H1 = H + 1;
    const size_t p = (size_t)n * H * W + (size_t)i * W + j;
    if (dI) {
        // f[i][j] enters gx at (i, j) with +, at (i, j+1) with -, gy at (i, j) with +, at (i+1, j) with -
        const float c = In[(size_t)i * W + j];
        const float sx0 = sgn(c - at(In, H, W, i, j - 1)), sx1 = sgn(at(In, H, W, i, j + 1) - c);
        const float sy0 = sgn(c - at(In, H, W, i - 1, j)), sy1 = sgn(at(In, H, W, i + 1, j) - c);
        const float v = sx0 * wx[base + (size_t)i * W1 + j] - sx1 * wx[base + (size_t)i * W1 + j + 1] + sy0 * wy[base + (size_t)i * W1 + j] -
                        sy1 * wy[base + (size_t)(i + 1) * W1 + j];
        dI[is * p] += v * w_over_numel;
    }
    if (dR) {
        // d term / d hx(a, b) = avg9(qx)(a, b) (the pooling is its own adjoint); hx(a, b) = |Rg[a][b] - Rg[a][b-1]|
        const float c = gn[(size_t)i * W + j];
        const float sx0 = sgn(c - at(gn, H, W, i, j - 1)), sx1 = sgn(at(gn, H, W, i, j + 1) - c);
        const float sy0 = sgn(c - at(gn, H, W, i - 1, j)), sy1 = sgn(at(gn, H, W, i + 1, j) - c);
        const float v = sx0 * avg9(qx + base, H1, W1, i, j) - sx1 * avg9(qx + base, H1, W1, i, j + 1) + sy0 * avg9(qy + base, H1, W1, i, j) -
                        sy1 * avg9(qy + base, H1, W1, i + 1, j);
        const float d = v * w_over_numel;
        dR[rs * p + 0] += 0.299f * d; dR[rs * p + 1] += 0.587f * d; dR[rs * p + 2] += 0.114f * d;
    }
}

// replicate padding and its adjoint (every padded position adds its gradient to the edge pixel it copied)
__global__ void pad_replicate_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int pad, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int Wp = W + 2 * pad, Hp = H + 2 * pad;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int w = (int)(q % Wp); q /= Wp;
    const int h = (int)(q % Hp);
    const size_t n = q / Hp;
    const int sh = min(max(h - pad, 0), H - 1), sw = min(max(w - pad, 0), W - 1);
    y[i] = x[((n * H + sh) * W + sw) * C + c];
}
__global__ void pad_replicate_backward_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C, int pad, int accumulate,
                                              size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int Wp = W + 2 * pad, Hp = H + 2 * pad;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int w = (int)(q % W); q /= W;
    const int h = (int)(q % H);
    const size_t n = q / H;
    // padded rows / columns that clamp onto (h, w)
    const int h0 = h == 0 ? 0 : h + pad, h1 = h == H - 1 ? Hp - 1 : h + pad;
    const int w0 = w == 0 ? 0 : w + pad, w1 = w == W - 1 ? Wp - 1 : w + pad;
    float s = 0.f;
    for (int a = h0; a <= h1; ++a)
        for (int b = w0; b <= w1; ++b) s += dy[((n * Hp + a) * Wp + b) * C + c];
    dx[i] = (accumulate ? dx[i] : 0.f) + s;
}

// DecomNet's input in one pass (model.py:45-46: input_max = max over channels; input_img = cat(input_max, input_im)): fp32 NCHW image -> 16-bit NHWC padded
// to 8 channels [max, R, G, B, 0, 0, 0, 0], one 16-byte store per pixel.
template <typename T, int CP>
__global__ void decom_input_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t plane, size_t total_px) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    const float* s = src + n * 3 * plane + pix;
    const float r = s[0], g = s[plane], b = s[2 * plane];
    const float m = fmaxf(fmaxf(r, g), b);
    if constexpr (CP == 8) *(ie_u32x4*)(dst + i * 8) = ie_u32x4{ie_pack2<T>(m, r), ie_pack2<T>(g, b), 0u, 0u};
    else *(ie_u32x2*)(dst + i * 4) = ie_u32x2{ie_pack2<T>(m, r), ie_pack2<T>(g, b)};
}

// The model's outputs in one pass (model.py:72-74, 229-231): rl = (R, L) fp32 pixel-major 4 floats, I_delta fp32 one float per pixel ->
// R (N,3,H,W), L (N,1,H,W), S = R * I_delta (N,3,H,W), all NCHW fp32.  A thread = one pixel: plane-wise coalesced stores.
__global__ void outputs_kernel(const float* __restrict__ rl, const float* __restrict__ idl, float* __restrict__ R, float* __restrict__ L,
                               float* __restrict__ S, size_t plane, size_t total_px) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    const ie_f32x4 v = *(const ie_f32x4*)(rl + 4 * i);
    const float d = idl[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (R) R[(n * 3 + c) * plane + pix] = v[c];
        S[(n * 3 + c) * plane + pix] = v[c] * d;
    }
    if (L) L[i] = v[3];
}

}  // namespace

IE_EXPORT int ie_retinex_decom_input(const float* x_nchw, void* dst_nhwc, int dtype, int cpad, int N, int H, int W, void* stream) {
    if (!x_nchw || !dst_nhwc || N <= 0 || H <= 0 || W <= 0 || (cpad != 4 && cpad != 8)) IE_FAIL(IE_ERR_INVALID, "ie_retinex_decom_input: bad args (cpad 4 or 8)");
    const size_t plane = (size_t)H * W, total = plane * N;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == IE_F16 && cpad == 8) hipLaunchKernelGGL((decom_input_kernel<_Float16, 8>), dim3(nblk(total)), dim3(256), 0, st, x_nchw, (_Float16*)dst_nhwc, plane, total);
    else if (dtype == IE_F16) hipLaunchKernelGGL((decom_input_kernel<_Float16, 4>), dim3(nblk(total)), dim3(256), 0, st, x_nchw, (_Float16*)dst_nhwc, plane, total);
    else if (dtype == IE_BF16 && cpad == 8) hipLaunchKernelGGL((decom_input_kernel<__bf16, 8>), dim3(nblk(total)), dim3(256), 0, st, x_nchw, (__bf16*)dst_nhwc, plane, total);
    else if (dtype == IE_BF16) hipLaunchKernelGGL((decom_input_kernel<__bf16, 4>), dim3(nblk(total)), dim3(256), 0, st, x_nchw, (__bf16*)dst_nhwc, plane, total);
    else IE_FAIL(IE_ERR_UNSUPPORTED, "ie_retinex_decom_input: dtype %d (IE_F16 / IE_BF16)", dtype);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_retinex_outputs_f32(const float* rl_nhwc4, const float* i_delta, float* R_nchw, float* L_nchw, float* S_nchw, int N, int H, int W, void* stream) {
    if (!rl_nhwc4 || !i_delta || !S_nchw || N <= 0 || H <= 0 || W <= 0) IE_FAIL(IE_ERR_INVALID, "ie_retinex_outputs_f32: bad args");
    const size_t plane = (size_t)H * W, total = plane * N;
    hipLaunchKernelGGL(outputs_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, rl_nhwc4, i_delta, R_nchw, L_nchw, S_nchw, plane, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_retinex_recon_f32(const float* R, int r_stride, const float* I, int i_stride, const float* T, float weight, float* dR, float* dI,
                                   double* loss, int N, int H, int W, void* stream) {
    if (!R || !I || !T || !loss || N <= 0 || H <= 0 || W <= 0 || r_stride < 3 || i_stride < 1) IE_FAIL(IE_ERR_INVALID, "ie_retinex_recon_f32: bad args");
    const size_t npx = (size_t)N * H * W;
    hipLaunchKernelGGL(recon_kernel, dim3(nblk_capped(npx)), dim3(256), 0, (hipStream_t)stream, R, r_stride, I, i_stride, T, weight / (float)(3.0 * (double)npx),
                       dR, dI, loss, npx);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_l1_detached_f32(const float* A, const float* B, int stride, int C, float weight, float* dA, double* loss, size_t npx, void* stream) {
    if (!A || !B || !loss || npx == 0 || C <= 0 || stride < C) IE_FAIL(IE_ERR_INVALID, "ie_l1_detached_f32: bad args");
    const size_t n = npx * C;
    hipLaunchKernelGGL(l1_detached_kernel, dim3(nblk_capped(n)), dim3(256), 0, (hipStream_t)stream, A, B, stride, C, weight / (float)n, dA, loss, n);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT size_t ie_retinex_smooth_ws_floats(int N, int H, int W) { return 2 * (size_t)N * H * W + 6 * (size_t)N * (H + 1) * (W + 1); }

IE_EXPORT int ie_retinex_smooth_f32(const float* I, int i_stride, const float* R, int r_stride, float weight, float* dI, float* dR, double* loss, float* ws,
                                    int N, int H, int W, void* stream) {
    if (!I || !R || !loss || !ws || N <= 0 || H <= 0 || W <= 0 || N > 65535 || H + 1 > 65535 || r_stride < 3 || i_stride < 1)
        IE_FAIL(IE_ERR_INVALID, "ie_retinex_smooth_f32: bad args");
    const size_t npx = (size_t)N * H * W, ng = (size_t)N * (H + 1) * (W + 1);
    float *g = ws, *ic = g + npx;
    float *hx = ic + npx, *hy = hx + ng, *wx = hy + ng, *wy = wx + ng, *qx = wy + ng, *qy = qx + ng;
    const float wn = weight / (float)ng;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gray_kernel, dim3(nblk(npx)), dim3(256), 0, st, R, r_stride, I, i_stride, g, ic, npx);
    const dim3 gg((unsigned)((W + 1 + 255) / 256), (unsigned)(H + 1), (unsigned)N), gi((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)N);
    hipLaunchKernelGGL(smooth_h_kernel, gg, dim3(256), 0, st, g, hx, hy, H, W);
    const dim3 gt(gg.x, (unsigned)(H + 1 < 128 ? H + 1 : 128), gg.z);
    hipLaunchKernelGGL(smooth_term_kernel, gt, dim3(256), 0, st, ic, hx, hy, wx, wy, qx, qy, loss, wn, H, W);
    if (dI || dR) hipLaunchKernelGGL(smooth_grad_kernel, gi, dim3(256), 0, st, ic, g, wx, wy, qx, qy, dI, i_stride, dR, r_stride, wn, H, W);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_pad_replicate_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int pad, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || pad < 0) IE_FAIL(IE_ERR_INVALID, "ie_pad_replicate_nhwc_f32: bad args");
    const size_t total = (size_t)N * (H + 2 * pad) * (W + 2 * pad) * C;
    hipLaunchKernelGGL(pad_replicate_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, pad, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_pad_replicate_backward_nhwc_f32(const float* dy, float* dx, int N, int H, int W, int C, int pad, int accumulate, void* stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || pad < 0) IE_FAIL(IE_ERR_INVALID, "ie_pad_replicate_backward_nhwc_f32: bad args");
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(pad_replicate_backward_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W, C, pad, accumulate, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
