// Zero-reference training of Zero-DCE: the four losses of zero_dce.py:34-79 with their gradients, and
// the backward pass of the LE curve.  fp32; images NCHW (N,3,H,W) like the reference, the adjust maps NHWC (N,H,W,Cp).
//   spatial consistency  nn/loss/image.py:905-1270 (num_regions = 4, patch 4): mean over pooled cells of the squared
//                        differences of the 4-neighbour contrasts of avg-pool4(mean_c(enhanced)) and avg-pool4(mean_c(image))
//   exposure control     nn/loss/image.py:534-568: mean((avg-pool16(mean_c(enhanced)) - E)^2)
//   colour constancy     nn/loss/image.py:191-217: mean_n sqrt((mr-mg)^4 + (mr-mb)^4 + (mb-mg)^4) of the per-frame channel means
//   illumination TV      nn/loss/image.py:1355-1392: 2 (sum dh^2 / count_h + sum dw^2 / count_w) / N on the adjust maps
// All four are reductions over the frame: HBM-bound single passes (one over enhanced+image, one over adjust, one writing
// the image gradient), fp64 accumulators.
#include "ie_common.h"
#include <math.h>

namespace {

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

// One block per 16x16 pixel tile of one frame: D = pool4(mean_c enh) - pool4(mean_c img) for its 4x4 cells, M = pool16(mean_c enh),
// per-frame channel sums of enh.  AvgPool2d floors: cells that do not fit are not produced (their pixels still count for colour).
__global__ __launch_bounds__(256) void zref_pool_kernel(const float* __restrict__ enh, const float* __restrict__ img, float* __restrict__ dmap,
                                                        float* __restrict__ mmap, double* __restrict__ chansum, int H, int W, int Hp4, int Wp4,
                                                        int Hp16, int Wp16) {
    __shared__ float se[256], sd[256], sc[3][256];
    const int n = blockIdx.z, tyb = blockIdx.y, txb = blockIdx.x;
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    const int y = tyb * 16 + ly, x = txb * 16 + lx;
    const bool in = y < H && x < W;
    const size_t plane = (size_t)H * W, base = (size_t)n * 3 * plane + (size_t)y * W + x;
    float e[3] = {0.f, 0.f, 0.f}, o[3] = {0.f, 0.f, 0.f};
    if (in) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { e[c] = enh[base + c * plane]; o[c] = img[base + c * plane]; }
    }
    const float em = (e[0] + e[1] + e[2]) * (1.f / 3.f), om = (o[0] + o[1] + o[2]) * (1.f / 3.f);
    se[threadIdx.x] = em; sd[threadIdx.x] = em - om;
#pragma unroll
    for (int c = 0; c < 3; ++c) sc[c][threadIdx.x] = e[c];
    __syncthreads();
    if (threadIdx.x < 16) {                          // one thread per 4x4 cell of the tile
        const int cy = threadIdx.x >> 2, cx = threadIdx.x & 3;
        const int gy = tyb * 4 + cy, gx = txb * 4 + cx;
        if (gy < Hp4 && gx < Wp4) {
            float s = 0.f;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) s += sd[(cy * 4 + i) * 16 + cx * 4 + j];
            dmap[((size_t)n * Hp4 + gy) * Wp4 + gx] = s * (1.f / 16.f);
        }
    } else if (threadIdx.x == 16) {
        if (tyb < Hp16 && txb < Wp16) {
            float s = 0.f;
            for (int i = 0; i < 256; ++i) s += se[i];
            mmap[((size_t)n * Hp16 + tyb) * Wp16 + txb] = s * (1.f / 256.f);
        }
    } else if (threadIdx.x >= 32 && threadIdx.x < 35) {
        const int c = threadIdx.x - 32;
        double s = 0.0;
        for (int i = 0; i < 256; ++i) s += sc[c][i];
        atomicAdd(&chansum[n * 3 + c], s);
    }
}

__device__ __forceinline__ double block_sum(double v) {
    __shared__ double sh[256];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    return sh[0];
}

// spatial consistency on the D map: loss = sum over cells and the 4 directions of (D_c - D_nb)^2 (D_nb = 0 outside: the
// reference's conv2d zero padding); gD = d(sum)/dD_c = sum_dirs 2 (D_c - D_nb) + sum_{in-range neighbours} 2 (D_c - D_x)
__global__ void zref_spa_kernel(const float* __restrict__ dmap, float* __restrict__ gd, double* __restrict__ acc, int Hp, int Wp, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double loss = 0.0;
    if (i < total) {
        const int x = (int)(i % Wp);
        const int y = (int)((i / Wp) % Hp);
        const float d = dmap[i];
        const bool hl = x > 0, hr = x < Wp - 1, hu = y > 0, hd = y < Hp - 1;
        const float l = hl ? dmap[i - 1] : 0.f, r = hr ? dmap[i + 1] : 0.f, u = hu ? dmap[i - Wp] : 0.f, dn = hd ? dmap[i + Wp] : 0.f;
        const float a = d - l, b = d - r, c = d - u, e = d - dn;
        loss = (double)a * a + (double)b * b + (double)c * c + (double)e * e;
        float g = 2.f * (a + b + c + e);
        if (hl) g += 2.f * a;
        if (hr) g += 2.f * b;
        if (hu) g += 2.f * c;
        if (hd) g += 2.f * e;
        gd[i] = g;
    }
    loss = block_sum(loss);
    if (threadIdx.x == 0) atomicAdd(acc, loss);
}

__global__ void zref_exp_kernel(const float* __restrict__ mmap, float* __restrict__ gm, double* __restrict__ acc, float mean_val, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double loss = 0.0;
    if (i < total) {
        const float d = mmap[i] - mean_val;
        loss = (double)d * d;
        gm[i] = 2.f * d;
    }
    loss = block_sum(loss);
    if (threadIdx.x == 0) atomicAdd(acc, loss);
}

// colour constancy per frame from the channel sums: loss_n = sqrt(a^4 + b^4 + c^4), a = mr-mg, b = mr-mb, c = mb-mg
__global__ void zref_col_kernel(const double* __restrict__ chansum, float* __restrict__ gcol, double* __restrict__ acc, int N, double inv_hw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double mr = chansum[n * 3] * inv_hw, mg = chansum[n * 3 + 1] * inv_hw, mb = chansum[n * 3 + 2] * inv_hw;
    const double a = mr - mg, b = mr - mb, c = mb - mg;
    const double l = sqrt(a * a * a * a + b * b * b * b + c * c * c * c);
    atomicAdd(acc, l);
    const double inv = l > 0.0 ? 1.0 / l : 0.0;                  // torch: d sqrt(0) -> inf * 0 = nan; means are never exactly equal in practice
    const double ga = 2.0 * a * a * a * inv, gb = 2.0 * b * b * b * inv, gc = 2.0 * c * c * c * inv;   // d l / d a etc.
    gcol[n * 3 + 0] = (float)(ga + gb);
    gcol[n * 3 + 1] = (float)(-ga - gc);
    gcol[n * 3 + 2] = (float)(-gb + gc);
}

// d(loss)/d(enhanced): the three image losses folded into one pass over the frame
__global__ void zref_grad_image_kernel(const float* __restrict__ gd, const float* __restrict__ gm, const float* __restrict__ gcol,
                                       float* __restrict__ genh, int H, int W, int Hp4, int Wp4, int Hp16, int Wp16, float s_spa, float s_exp,
                                       float s_col, int accumulate, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    size_t r = i / W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % 3);
    const size_t n = r / 3;
    float g = s_col * gcol[n * 3 + c];
    const int y4 = y >> 2, x4 = x >> 2, y16 = y >> 4, x16 = x >> 4;
    if (y4 < Hp4 && x4 < Wp4) g += s_spa * gd[(n * Hp4 + y4) * Wp4 + x4];
    if (y16 < Hp16 && x16 < Wp16) g += s_exp * gm[(n * Hp16 + y16) * Wp16 + x16];
    genh[i] = (accumulate ? genh[i] : 0.f) + g;
}

// illumination-smoothness TV on NHWC maps: sums of squared vertical / horizontal differences + gradient.
// Grid-stride over V-channel pieces (V = 4: 16-byte accesses when C % 4 == 0) with per-thread double partials and ONE pair of
// atomics per workgroup: one thread per element with a pair of same-address atomics per 256 elements serialised in L2
// (3.1 M atomics on two addresses for 8 x 1080p x 24 maps: 37 ms; this form: 1.5 ms).
template <int V>
__global__ __launch_bounds__(256) void zref_tv_kernel(const float* __restrict__ a, float* __restrict__ grad, double* __restrict__ acc, int H, int W,
                                                      int C, float gh, float gw, int accumulate, size_t total_v) {
    typedef __attribute__((ext_vector_type(V))) float FV;
    const int CV = C / V;
    const size_t rs = (size_t)W * C;
    double sh = 0.0, sw = 0.0;
    for (size_t iv = (size_t)blockIdx.x * blockDim.x + threadIdx.x; iv < total_v; iv += (size_t)gridDim.x * blockDim.x) {
        size_t r = iv / CV;
        const size_t i = r * C + (iv - r * CV) * V;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const FV v = *(const FV*)(a + i);
        FV g;
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = 0.f;
        if (y > 0) {
            const FV d = v - *(const FV*)(a + i - rs);
#pragma unroll
            for (int e = 0; e < V; ++e) { sh += (double)d[e] * d[e]; g[e] += gh * d[e]; }
        }
        if (y < H - 1) g -= gh * (*(const FV*)(a + i + rs) - v);
        if (x > 0) {
            const FV d = v - *(const FV*)(a + i - C);
#pragma unroll
            for (int e = 0; e < V; ++e) { sw += (double)d[e] * d[e]; g[e] += gw * d[e]; }
        }
        if (x < W - 1) g -= gw * (*(const FV*)(a + i + C) - v);
        if (grad) {
            if (accumulate) g += *(const FV*)(grad + i);
            *(FV*)(grad + i) = g;
        }
    }
    sh = block_sum(sh);
    __syncthreads();
    sw = block_sum(sw);
    if (threadIdx.x == 0) { atomicAdd(acc, sh); atomicAdd(acc + 1, sw); }
}

// backward of y_{i+1} = y_i + r_i (y_i^2 - y_i): dL/dr_i = g (y_i^2 - y_i), g <- g (1 + r_i (2 y_i - 1)); the image gets no gradient.
// thread = one pixel, all three colour channels: its 3*iters curve values are contiguous in the NHWC map (96 bytes for 8 iterations),
// so map reads and gradient writes are dense; the NCHW image / gradient planes are read coalesced across the pixels of a wave.
__global__ __launch_bounds__(256) void le_curve_backward_kernel(const float* __restrict__ img, const float* __restrict__ adj,
                                                                const float* __restrict__ genh, float* __restrict__ gadj, int H, int W, int Cp,
                                                                int iters, int accumulate, size_t total_px) {
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // over (n, h, w)
    if (ip >= total_px) return;
    const size_t hw = (size_t)H * W;
    const size_t n = ip / hw, q = ip - n * hw;
    const size_t pa = ip * Cp;
    const bool shared_map = Cp < 3 * iters;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t i = (n * 3 + c) * hw + q;
        float ys[16], rs[16];
        float v = img[i];
        for (int k = 0; k < iters; ++k) {
            rs[k] = adj[pa + (shared_map ? c : 3 * k + c)];
            ys[k] = v;
            v = ie_curve_step(v, rs[k]);
        }
        float g = genh[i], gshared = 0.f;
        for (int k = iters - 1; k >= 0; --k) {
            const float yk = ys[k];
            const float gr = g * (yk * yk - yk);
            if (shared_map) gshared += gr;
            else gadj[pa + 3 * k + c] = (accumulate ? gadj[pa + 3 * k + c] : 0.f) + gr;
            g = g * (1.f + rs[k] * (2.f * yk - 1.f));
        }
        if (shared_map) gadj[pa + c] = (accumulate ? gadj[pa + c] : 0.f) + gshared;
    }
}

// Fast path for the reference configuration (8 iterations, 24 maps per pixel, not shared): the pixel's 24 curve values arrive as six
// 16-byte loads and its 24 gradients leave as six 16-byte stores (the scalar form issues 48 four-byte accesses 96 bytes apart per lane).
__global__ __launch_bounds__(256) void le_curve_backward8_kernel(const float* __restrict__ img, const float* __restrict__ adj,
                                                                 const float* __restrict__ genh, float* __restrict__ gadj, int H, int W,
                                                                 int accumulate, size_t total_px) {
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // over (n, h, w)
    if (ip >= total_px) return;
    const size_t hw = (size_t)H * W;
    const size_t n = ip / hw, q = ip - n * hw;
    float r[24], gr[24];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const ie_f32x4 v = *(const ie_f32x4*)(adj + ip * 24 + j * 4);
        r[4 * j] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t i = (n * 3 + c) * hw + q;
        float ys[8];
        float v = img[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) { ys[k] = v; v = ie_curve_step(v, r[3 * k + c]); }
        float g = genh[i];
#pragma unroll
        for (int k = 7; k >= 0; --k) {
            const float yk = ys[k];
            gr[3 * k + c] = g * (yk * yk - yk);
            g = g * (1.f + r[3 * k + c] * (2.f * yk - 1.f));
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        ie_f32x4 o = {gr[4 * j], gr[4 * j + 1], gr[4 * j + 2], gr[4 * j + 3]};
        if (accumulate) o += *(const ie_f32x4*)(gadj + ip * 24 + j * 4);
        *(ie_f32x4*)(gadj + ip * 24 + j * 4) = o;
    }
}

}  // namespace

IE_EXPORT size_t ie_zref_workspace_floats(int N, int H, int W) {
    const size_t c4 = (size_t)N * (H / 4) * (W / 4), c16 = (size_t)N * (H / 16) * (W / 16);
    return 2 * c4 + 2 * c16 + 3 * (size_t)N + 16;
}

// sums_dev[0] = spa sum, [1] = exp sum, [2] = col sum (over frames), [3..3+3N) = channel sums (scratch)
IE_EXPORT int ie_zref_image_losses_f32(const float* enhanced, const float* image, int N, int H, int W, float exp_mean, float w_spa, float w_exp,
                                       float w_col, double* sums_dev, float* ws_dev, float* grad_enhanced, int accumulate, void* stream) {
    if (!enhanced || !image || !sums_dev || !ws_dev || N <= 0 || H < 16 || W < 16) IE_FAIL(IE_ERR_INVALID, "ie_zref_image_losses_f32: bad args (H, W >= 16)");
    hipStream_t st = (hipStream_t)stream;
    const int Hp4 = H / 4, Wp4 = W / 4, Hp16 = H / 16, Wp16 = W / 16;
    const size_t c4 = (size_t)N * Hp4 * Wp4, c16 = (size_t)N * Hp16 * Wp16;
    float* dmap = ws_dev; float* gd = dmap + c4; float* mmap = gd + c4; float* gm = mmap + c16; float* gcol = gm + c16;
    IE_HIP_CHECK(hipMemsetAsync(sums_dev, 0, sizeof(double) * (3 + 3 * (size_t)N), st));
    hipLaunchKernelGGL(zref_pool_kernel, dim3(ie_cdiv(W, 16), ie_cdiv(H, 16), N), dim3(256), 0, st, enhanced, image, dmap, mmap, sums_dev + 3, H, W,
                       Hp4, Wp4, Hp16, Wp16);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(zref_spa_kernel, dim3(nblk(c4)), dim3(256), 0, st, dmap, gd, sums_dev + 0, Hp4, Wp4, c4);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(zref_exp_kernel, dim3(nblk(c16)), dim3(256), 0, st, mmap, gm, sums_dev + 1, exp_mean, c16);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(zref_col_kernel, dim3(nblk(N)), dim3(256), 0, st, sums_dev + 3, gcol, sums_dev + 2, N, 1.0 / ((double)H * W));
    IE_LAUNCH_CHECK();
    if (grad_enhanced) {
        // d mean_cells / d pixel: the cell means divide by 16 (256) pixels and 3 channels; colour: / (H W) per channel, / N frames
        const float s_spa = w_spa / (float)c4 / 48.f, s_exp = w_exp / (float)c16 / 768.f, s_col = w_col / (float)N / ((float)H * (float)W);
        const size_t total = (size_t)N * 3 * H * W;
        hipLaunchKernelGGL(zref_grad_image_kernel, dim3(nblk(total)), dim3(256), 0, st, gd, gm, gcol, grad_enhanced, H, W, Hp4, Wp4, Hp16, Wp16,
                           s_spa, s_exp, s_col, accumulate, total);
        IE_LAUNCH_CHECK();
    }
    return IE_OK;
}

// sums_dev[0] = sum of squared vertical differences, [1] = horizontal; loss = weight * 2 (s0/count_h + s1/count_w) / N
IE_EXPORT int ie_tv_loss_nhwc_f32(const float* a, int N, int H, int W, int C, float weight, double* sums_dev, float* grad, int accumulate, void* stream) {
    if (!a || !sums_dev || N <= 0 || H < 2 || W < 2 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_tv_loss_nhwc_f32: bad args");
    hipStream_t st = (hipStream_t)stream;
    IE_HIP_CHECK(hipMemsetAsync(sums_dev, 0, 2 * sizeof(double), st));
    const double count_h = (double)C * (H - 1) * W, count_w = (double)C * H * (W - 1);
    const float gh = (float)(weight * 2.0 * 2.0 / (count_h * N)), gw = (float)(weight * 2.0 * 2.0 / (count_w * N));
    const size_t total = (size_t)N * H * W * C;
    if ((C & 3) == 0) {
        const size_t blocks = (total / 4 + 255) / 256;
        hipLaunchKernelGGL(zref_tv_kernel<4>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, a, grad, sums_dev, H, W, C, gh, gw,
                           accumulate, total / 4);
    } else {
        const size_t blocks = (total + 255) / 256;
        hipLaunchKernelGGL(zref_tv_kernel<1>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, a, grad, sums_dev, H, W, C, gh, gw,
                           accumulate, total);
    }
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_le_curve_backward(const float* img, const float* adjust_nhwc, int Cp, const float* grad_enhanced, float* grad_adjust_nhwc, int N, int H,
                                   int W, int iters, int accumulate, void* stream) {
    if (!img || !adjust_nhwc || !grad_enhanced || !grad_adjust_nhwc || N <= 0 || H <= 0 || W <= 0 || iters < 1 || iters > 16 || Cp < 3)
        IE_FAIL(IE_ERR_INVALID, "ie_le_curve_backward: bad args (1 <= iters <= 16)");
    const size_t total = (size_t)N * H * W;
    if (iters == 8 && Cp == 24)
        hipLaunchKernelGGL(le_curve_backward8_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, img, adjust_nhwc, grad_enhanced,
                           grad_adjust_nhwc, H, W, accumulate, total);
    else
        hipLaunchKernelGGL(le_curve_backward_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, img, adjust_nhwc, grad_enhanced,
                           grad_adjust_nhwc, H, W, Cp, iters, accumulate, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
