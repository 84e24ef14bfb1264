// Colour-space, box / guided-filter and generalised LE-curve kernels of the V-channel / guided curve models:
//   ZeroDCEV  src/mon/vision/enhance/llie/zero_dce/zero_dce_v.py:172-230  (HSV split, curve on V at 256 x 256, FastGuidedFilter up)
//   GCENet    src/mon/vision/enhance/llie/gcenet/gcenet.py:460-502       (brightness-attention curve, GuidedFilter)
//   filters   src/mon/vision/filtering/guided_filter.py:99-204, box_filter.py:24-110
// kornia (rgb_to_hsv, hsv_to_rgb, median_blur, sobel, rgb_to_grayscale) is a third-party dependency that is not in the
// reference tree: its published algorithms are restated here ("parity unpinned" for those).
// All tensors fp32 NHWC; every kernel is one streaming pass (HBM-bound, a few channels per pixel).
#include "ie_common.h"

namespace {

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

// kornia.color.rgb_to_hsv: h in [0, 2pi), s, v in [0, 1]
__global__ void rgb_to_hsv_kernel(const float* __restrict__ x, float* __restrict__ hsv, float* __restrict__ vout, size_t npix, float eps) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float r = x[i * 3], g = x[i * 3 + 1], b = x[i * 3 + 2];
    float mx = r; int am = 0;
    if (g > mx) { mx = g; am = 1; }
    if (b > mx) { mx = b; am = 2; }
    const float mn = fminf(r, fminf(g, b));
    float dc = mx - mn;
    const float s = dc / (mx + eps);
    dc = dc == 0.f ? 1.f : dc;
    const float rc = mx - r, gc = mx - g, bc = mx - b;
    float h = am == 0 ? (bc - gc) : (am == 1 ? (rc - bc) + 2.f * dc : (gc - rc) + 4.f * dc);
    h = h / dc;
    h = h / 6.f;
    h = h - floorf(h);                       // (h / 6) % 1.0 with the divisor's sign
    h = 6.283185307179586f * h;
    if (hsv) { hsv[i * 3] = h; hsv[i * 3 + 1] = s; hsv[i * 3 + 2] = mx; }
    if (vout) vout[i] = mx;
}

// kornia.color.hsv_to_rgb; v_new (one channel) replaces V when given (zero_dce_v.py:225-230)
__global__ void hsv_to_rgb_kernel(const float* __restrict__ hsv, const float* __restrict__ v_new, float* __restrict__ rgb, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float h = hsv[i * 3] / 6.283185307179586f, s = hsv[i * 3 + 1], v = v_new ? v_new[i] : hsv[i * 3 + 2];
    const float h6 = h * 6.f;
    float hi = floorf(h6);
    hi = hi - 6.f * floorf(hi / 6.f);        // floor(h*6) % 6
    const float h6m = h6 - 6.f * floorf(h6 / 6.f);
    const float f = h6m - hi;
    const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
    const int k = (int)hi;
    float R, G, B;
    switch (k) {
    case 0: R = v; G = t; B = p; break;
    case 1: R = q; G = v; B = p; break;
    case 2: R = p; G = v; B = t; break;
    case 3: R = p; G = q; B = v; break;
    case 4: R = t; G = p; B = v; break;
    default: R = v; G = p; B = q; break;
    }
    rgb[i * 3] = R; rgb[i * 3 + 1] = G; rgb[i * 3 + 2] = B;
}

// y <- y + r_k (y^2 - y) on C channels (C = 1: the V channel of ZeroDCEV, 15 iterations; zero_dce_v.py:190-191), or GCENet's
// brightness-attention form (gcenet.py:478-484): bright = y (1 - bam); dark = y bam; y = bright + dark + r (dark^2 - dark),
// with ONE C-channel map r for all iterations (shared) or C*iters maps.
__global__ void le_curve_c_kernel(const float* __restrict__ x, const float* __restrict__ adj, const float* __restrict__ bam,
                                  float* __restrict__ y, size_t npix, int C, int Ca, int iters, int shared) {
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * C) return;
    const size_t px = i / C;
    const int c = (int)(i % C);
    const float* a = adj + px * Ca;
    float v = x[i];
    if (bam) {
        const float m = bam[px * (size_t)1];
        for (int k = 0; k < iters; ++k) {
            const float r = a[shared ? c : C * k + c];
            const float bright = v * (1.f - m);
            const float dark = v * m;
            const float d2 = dark * dark;
            const float u = d2 - dark;
            const float w = r * u;
            const float bd = bright + dark;
            v = bd + w;
        }
    } else {
        for (int k = 0; k < iters; ++k) v = ie_curve_step(v, a[shared ? c : C * k + c]);
    }
    y[i] = v;
}

// box sum over the (2r+1)^2 window clipped to the image == diff_y(diff_x(cumsum)) of box_filter.py:24-110
__global__ void box_filter_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int r, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int w0 = (int)(q % W); q /= W;
    const int h0 = (int)(q % H);
    const size_t n = q / H;
    double s = 0.0;
    for (int dy = -r; dy <= r; ++dy) {
        const int hh = h0 + dy;
        if (hh < 0 || hh >= H) continue;
        for (int dx = -r; dx <= r; ++dx) {
            const int ww = w0 + dx;
            if (ww < 0 || ww >= W) continue;
            s += (double)x[((n * H + hh) * W + ww) * C + c];
        }
    }
    y[i] = (float)s;
}

// guided-filter coefficients (guided_filter.py:126-144 / 184-200): per pixel and channel of y
//   A = cov_xy / (var_x + eps), b = mean_y - A mean_x    with window means over the clipped (2r+1)^2 box (sums in fp64)
// x has Cx = 1 or Cy channels.
__global__ void gf_coeff_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ A, float* __restrict__ B,
                                int H, int W, int Cx, int Cy, int r, float eps, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cy);
    size_t q = i / Cy;
    const int w0 = (int)(q % W); q /= W;
    const int h0 = (int)(q % H);
    const size_t n = q / H;
    const int cx = Cx == 1 ? 0 : c;
    double sx = 0, sy = 0, sxy = 0, sxx = 0;
    int cnt = 0;
    for (int dy = -r; dy <= r; ++dy) {
        const int hh = h0 + dy;
        if (hh < 0 || hh >= H) continue;
        for (int dx = -r; dx <= r; ++dx) {
            const int ww = w0 + dx;
            if (ww < 0 || ww >= W) continue;
            const size_t p = (n * H + hh) * W + ww;
            const float xv = x[p * Cx + cx], yv = y[p * Cy + c];
            sx += xv; sy += yv; sxy += (double)(xv * yv); sxx += (double)(xv * xv);     // products in fp32 like the reference's x * y
            ++cnt;
        }
    }
    const double inv = 1.0 / cnt;
    const float mean_x = (float)(sx * inv), mean_y = (float)(sy * inv);
    const float cov = (float)(sxy * inv) - mean_x * mean_y;
    const float var = (float)(sxx * inv) - mean_x * mean_x;
    const float a = cov / (var + eps);
    A[i] = a;
    B[i] = mean_y - a * mean_x;
}

// GuidedFilter output: mean_A * x + mean_b with mean_* = box(.) / N (guided_filter.py:145-147)
__global__ void gf_apply_kernel(const float* __restrict__ x, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out,
                                int H, int W, int Cx, int Cy, int r, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cy);
    size_t q = i / Cy;
    const int w0 = (int)(q % W); q /= W;
    const int h0 = (int)(q % H);
    const size_t n = q / H;
    double sa = 0, sb = 0;
    int cnt = 0;
    for (int dy = -r; dy <= r; ++dy) {
        const int hh = h0 + dy;
        if (hh < 0 || hh >= H) continue;
        for (int dx = -r; dx <= r; ++dx) {
            const int ww = w0 + dx;
            if (ww < 0 || ww >= W) continue;
            const size_t p = ((n * H + hh) * W + ww) * Cy + c;
            sa += A[p]; sb += B[p];
            ++cnt;
        }
    }
    const float ma = (float)(sa / cnt), mb = (float)(sb / cnt);
    out[i] = ma * x[(i / Cy) * Cx + (Cx == 1 ? 0 : c)] + mb;
}

// torch bicubic (A = -0.75), align_corners = True, border indices clamped
__device__ __forceinline__ float cc1(float x) { return ((-0.75f + 2.f) * x - (-0.75f + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x) { return ((-0.75f * x - 5.f * -0.75f) * x + 8.f * -0.75f) * x - 4.f * -0.75f; }
__device__ __forceinline__ float bicubic_at(const float* __restrict__ src, size_t n, int h, int w, int C, int c, int Ho, int Wo, int ho, int wo) {
    const float sh = Ho > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    const float ry = sh * ho, rx = sw * wo;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    const float ty = ry - iy, tx = rx - ix;
    const float wy[4] = {cc2(ty + 1.f), cc1(ty), cc1(1.f - ty), cc2(2.f - ty)};
    const float wx[4] = {cc2(tx + 1.f), cc1(tx), cc1(1.f - tx), cc2(2.f - tx)};
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), h - 1);
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), w - 1);
            row += src[((n * h + yy) * w + xx) * C + c] * wx[b];
        }
        acc += row * wy[a];
    }
    return acc;
}
// FastGuidedFilter output (guided_filter.py:201-204): interpolate(A) * x_hr + interpolate(b), optionally clipped (zero_dce_v.py:221)
__global__ void fgf_apply_kernel(const float* __restrict__ xhr, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out,
                                 int h, int w, int H, int W, int Cx, int Cy, int clip, float lo, float hi, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cy);
    size_t q = i / Cy;
    const int wo = (int)(q % W); q /= W;
    const int ho = (int)(q % H);
    const size_t n = q / H;
    const float a = bicubic_at(A, n, h, w, Cy, c, H, W, ho, wo), b = bicubic_at(B, n, h, w, Cy, c, H, W, ho, wo);
    float v = a * xhr[(i / Cy) * Cx + (Cx == 1 ? 0 : c)] + b;
    if (clip) v = fminf(fmaxf(v, lo), hi);
    out[i] = v;
}
__global__ void resize_bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int H, int W, int C, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int wo = (int)(q % W); q /= W;
    const int ho = (int)(q % H);
    dst[i] = bicubic_at(src, q / H, h, w, C, c, H, W, ho, wo);
}

// global max in two deterministic stages (zero_dce_v.py:205 `x / torch.max(x)`; boundary prior's g_max)
__global__ void max_partial_kernel(const float* __restrict__ a, float* __restrict__ partial, size_t total) {
    __shared__ float sm[256];
    float m = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, a[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
__global__ void max_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float sm[256];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, partial[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sm[0];
}

// kornia.filters.median_blur: k x k window over the ZERO-padded image, median of the k*k values, per channel.
// Selection by counting: the median is the window value with exactly (k*k-1)/2 smaller-or-equal-earlier values.
template <int K>
__global__ void median_blur_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int w0 = (int)(q % W); q /= W;
    const int h0 = (int)(q % H);
    const size_t n = q / H;
    constexpr int R = K / 2, NV = K * K;
    float v[NV];
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            const int hh = h0 + dy - R, ww = w0 + dx - R;
            v[dy * K + dx] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? x[((n * H + hh) * W + ww) * C + c] : 0.f;
        }
    float med = v[0];
    for (int a = 0; a < NV; ++a) {
        int rank = 0;
        for (int b = 0; b < NV; ++b) rank += (v[b] < v[a]) || (v[b] == v[a] && b < a);
        if (rank == NV / 2) med = v[a];
    }
    y[i] = med;
}

// kornia.filters.sobel(normalized=False, eps): magnitude sqrt(gx^2 + gy^2 + eps) of the 3x3 Sobel responses with REPLICATE
// padding (spatial_gradient pads with 'replicate'); per channel.
__global__ void sobel_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, float eps, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t q = i / C;
    const int w0 = (int)(q % W); q /= W;
    const int h0 = (int)(q % H);
    const size_t n = q / H;
    float v[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int hh = min(max(h0 + dy - 1, 0), H - 1), ww = min(max(w0 + dx - 1, 0), W - 1);
            v[dy][dx] = x[((n * H + hh) * W + ww) * C + c];
        }
    const float gx = (v[0][2] - v[0][0]) + 2.f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
    const float gy = (v[2][0] - v[0][0]) + 2.f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]);
    y[i] = sqrtf(gx * gx + gy * gy + eps);
}

// core.RGBToHVI.rgb_to_hvi (src/mon/core/image/color.py:332-361; HVI-CIDNet's transform): hue by the reference's assignment order (the LAST matching
// assignment wins: min == max -> 0, else R == max, else G == max, else B == max), saturation, intensity-dependent radius (sin(v pi / 2) + eps)^k.
__global__ void rgb_to_hvi_kernel(const float* __restrict__ x, float* __restrict__ y, size_t npix, float k, float eps) {
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float pi = 3.141592653589793f;
    const float r = x[i * 3], g = x[i * 3 + 1], b = x[i * 3 + 2];
    const float v = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float d = v - mn + eps;
    float hue;
    if (mn == v) hue = 0.f;
    else if (r == v) { hue = (g - b) / d; hue = hue - 6.f * floorf(hue / 6.f); }       // torch `%`: the divisor's sign
    else if (g == v) hue = 2.f + (b - r) / d;
    else hue = 4.f + (r - g) / d;
    hue = hue / 6.f;
    const float sat = v == 0.f ? 0.f : (v - mn) / (v + eps);
    const float cs = powf(sinf(v * 0.5f * pi) + eps, k);
    y[i * 3] = cs * sat * cosf(2.f * pi * hue);
    y[i * 3 + 1] = cs * sat * sinf(2.f * pi * hue);
    y[i * 3 + 2] = v;
}

// core.RGBToHVI.hvi_to_rgb (color.py:363-434, gated / gated2 off as constructed), optionally followed by clamp(0, 1) (lllinet.py:383-384)
__global__ void hvi_to_rgb_kernel(const float* __restrict__ x, float* __restrict__ y, size_t npix, float k, float eps, int clamp01) {
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float pi = 3.141592653589793f;
    float H = fminf(fmaxf(x[i * 3], -1.f), 1.f), V = fminf(fmaxf(x[i * 3 + 1], -1.f), 1.f);
    const float v = fminf(fmaxf(x[i * 3 + 2], 0.f), 1.f);
    const float cs = powf(sinf(v * 0.5f * pi) + eps, k);
    H = fminf(fmaxf(H / (cs + eps), -1.f), 1.f);
    V = fminf(fmaxf(V / (cs + eps), -1.f), 1.f);
    float h = atan2f(V, H) / (2.f * pi);
    h = h - floorf(h);                                   // % 1
    const float s = fminf(fmaxf(sqrtf(H * H + V * V), 0.f), 1.f);
    const float hi = floorf(h * 6.f), f = h * 6.f - hi;
    const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
    float R = 0.f, G = 0.f, B = 0.f;                     // hi == 6 (h rounding up to 1): the reference's zeros stay
    switch ((int)hi) {
    case 0: R = v; G = t; B = p; break;
    case 1: R = q; G = v; B = p; break;
    case 2: R = p; G = v; B = t; break;
    case 3: R = p; G = q; B = v; break;
    case 4: R = t; G = p; B = v; break;
    case 5: R = v; G = p; B = q; break;
    default: break;
    }
    if (clamp01) { R = fminf(fmaxf(R, 0.f), 1.f); G = fminf(fmaxf(G, 0.f), 1.f); B = fminf(fmaxf(B, 0.f), 1.f); }
    y[i * 3] = R; y[i * 3 + 1] = G; y[i * 3 + 2] = B;
}

}  // namespace

IE_EXPORT int ie_rgb_to_hvi_nhwc_f32(const float* x, float* hvi, size_t npix, float density_k, float eps, void* stream) {
    if (!x || !hvi || npix == 0) IE_FAIL(IE_ERR_INVALID, "ie_rgb_to_hvi_nhwc_f32: bad args");
    hipLaunchKernelGGL(rgb_to_hvi_kernel, dim3(nblk(npix)), dim3(256), 0, (hipStream_t)stream, x, hvi, npix, density_k, eps);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_hvi_to_rgb_nhwc_f32(const float* hvi, float* rgb, size_t npix, float density_k, float eps, int clamp01, void* stream) {
    if (!hvi || !rgb || npix == 0) IE_FAIL(IE_ERR_INVALID, "ie_hvi_to_rgb_nhwc_f32: bad args");
    hipLaunchKernelGGL(hvi_to_rgb_kernel, dim3(nblk(npix)), dim3(256), 0, (hipStream_t)stream, hvi, rgb, npix, density_k, eps, clamp01);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_rgb_to_hsv_nhwc_f32(const float* x, float* hsv, float* v, size_t npix, float eps, void* stream) {
    if (!x || (!hsv && !v) || npix == 0) IE_FAIL(IE_ERR_INVALID, "ie_rgb_to_hsv_nhwc_f32: bad args");
    hipLaunchKernelGGL(rgb_to_hsv_kernel, dim3(nblk(npix)), dim3(256), 0, (hipStream_t)stream, x, hsv, v, npix, eps);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_hsv_to_rgb_nhwc_f32(const float* hsv, const float* v_new, float* rgb, size_t npix, void* stream) {
    if (!hsv || !rgb || npix == 0) IE_FAIL(IE_ERR_INVALID, "ie_hsv_to_rgb_nhwc_f32: bad args");
    hipLaunchKernelGGL(hsv_to_rgb_kernel, dim3(nblk(npix)), dim3(256), 0, (hipStream_t)stream, hsv, v_new, rgb, npix);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_le_curve_nhwc_f32(const float* x, const float* adjust, const float* bam, float* y, size_t npix, int C, int Ca, int iters,
                                   void* stream) {
    if (!x || !adjust || !y || npix == 0 || C <= 0 || iters <= 0 || (Ca != C && Ca < C * iters))
        IE_FAIL(IE_ERR_INVALID, "ie_le_curve_nhwc_f32: bad args (C=%d Ca=%d iters=%d)", C, Ca, iters);
    hipLaunchKernelGGL(le_curve_c_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, x, adjust, bam, y, npix, C, Ca, iters,
                       Ca == C ? 1 : 0);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_box_filter_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int radius, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || radius < 0) IE_FAIL(IE_ERR_INVALID, "ie_box_filter_nhwc_f32: bad args");
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(box_filter_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, radius, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_guided_filter_nhwc_f32(const float* x, const float* y, float* out, float* ws, int N, int H, int W, int Cx, int Cy, int radius,
                                        float eps, void* stream) {
    if (!x || !y || !out || !ws || N <= 0 || H <= 0 || W <= 0 || (Cx != 1 && Cx != Cy) || Cy <= 0 || radius < 1)
        IE_FAIL(IE_ERR_INVALID, "ie_guided_filter_nhwc_f32: bad args");
    if (H <= 2 * radius + 1 || W <= 2 * radius + 1) IE_FAIL(IE_ERR_INVALID, "ie_guided_filter_nhwc_f32: image smaller than the window");
    const size_t total = (size_t)N * H * W * Cy;
    float* A = ws; float* B = ws + total;
    hipLaunchKernelGGL(gf_coeff_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, A, B, H, W, Cx, Cy, radius, eps, total);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gf_apply_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, A, B, out, H, W, Cx, Cy, radius, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_fast_guided_filter_nhwc_f32(const float* x_lr, const float* y_lr, const float* x_hr, float* out, float* ws, int N, int h, int w,
                                             int H, int W, int Cx, int Cy, int radius, float eps, int clip, float lo, float hi, void* stream) {
    if (!x_lr || !y_lr || !x_hr || !out || !ws || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || (Cx != 1 && Cx != Cy) || Cy <= 0 || radius < 1)
        IE_FAIL(IE_ERR_INVALID, "ie_fast_guided_filter_nhwc_f32: bad args");
    if (h <= 2 * radius + 1 || w <= 2 * radius + 1) IE_FAIL(IE_ERR_INVALID, "ie_fast_guided_filter_nhwc_f32: low-resolution image smaller than the window");
    const size_t lr = (size_t)N * h * w * Cy, total = (size_t)N * H * W * Cy;
    float* A = ws; float* B = ws + lr;
    hipLaunchKernelGGL(gf_coeff_kernel, dim3(nblk(lr)), dim3(256), 0, (hipStream_t)stream, x_lr, y_lr, A, B, h, w, Cx, Cy, radius, eps, lr);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(fgf_apply_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x_hr, A, B, out, h, w, H, W, Cx, Cy, clip, lo, hi, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_resize_bicubic_nhwc_f32(const float* x, float* y, int N, int h, int w, int H, int W, int C, void* stream) {
    if (!x || !y || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_resize_bicubic_nhwc_f32: bad args");
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(resize_bicubic_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, h, w, H, W, C, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_reduce_max_f32(const float* a, size_t n, float* out_dev, float* ws_dev, void* stream) {
    if (!a || !out_dev || !ws_dev || n == 0) IE_FAIL(IE_ERR_INVALID, "ie_reduce_max_f32: bad args");
    size_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(max_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, ws_dev, n);
    IE_LAUNCH_CHECK();
    hipLaunchKernelGGL(max_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws_dev, (int)blocks, out_dev);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_median_blur_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int ksize, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_median_blur_nhwc_f32: bad args");
    const size_t total = (size_t)N * H * W * C;
    hipStream_t st = (hipStream_t)stream;
    switch (ksize) {
    case 3: hipLaunchKernelGGL(median_blur_kernel<3>, dim3(nblk(total)), dim3(256), 0, st, x, y, H, W, C, total); break;
    case 5: hipLaunchKernelGGL(median_blur_kernel<5>, dim3(nblk(total)), dim3(256), 0, st, x, y, H, W, C, total); break;
    case 7: hipLaunchKernelGGL(median_blur_kernel<7>, dim3(nblk(total)), dim3(256), 0, st, x, y, H, W, C, total); break;
    case 9: hipLaunchKernelGGL(median_blur_kernel<9>, dim3(nblk(total)), dim3(256), 0, st, x, y, H, W, C, total); break;
    default: IE_FAIL(IE_ERR_UNSUPPORTED, "ie_median_blur_nhwc_f32: ksize %d (3, 5, 7, 9)", ksize);
    }
    IE_LAUNCH_CHECK();
    return IE_OK;
}
IE_EXPORT int ie_sobel_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, float eps, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) IE_FAIL(IE_ERR_INVALID, "ie_sobel_nhwc_f32: bad args");
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(sobel_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, eps, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
