// Library plumbing + HBM-bound boundary ops (layout conversion, u8 quantise, LE curve).
// All of these are streaming kernels: one pass over the tensor, 16-byte accesses where
// the layout allows, bounded by HBM bandwidth.
#include "ie_common.h"
#include <string.h>
#include <mutex>

// ---- error plumbing ---------------------------------------------------------------------
static thread_local char g_err[512] = "";
void ie_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
IE_EXPORT const char* ie_last_error(void) { return g_err; }
IE_EXPORT int ie_abi_version(void) { return IE_ABI_VERSION; }

// ---- deterministic mode -------------------------------------------------------------------
// Off (default): the conv weight gradients combine their split-K partial sums with fp32 atomics, whose order -- and with
// it the last bits of dW -- changes from run to run.  On: every split writes its partial tile to a slab of a library-owned
// scratch buffer and a second kernel adds the slabs in a fixed order (fp64 accumulate), so identical inputs give identical
// bits.  The scratch grows on demand with hipMalloc (a device synchronisation; do not toggle inside a graph capture).
static int g_det = 0;
static void* g_det_ws[16] = {nullptr};
static size_t g_det_bytes[16] = {0};
IE_EXPORT int ie_set_deterministic(int on) { g_det = on ? 1 : 0; return IE_OK; }
IE_EXPORT int ie_get_deterministic(void) { return g_det; }
int ie_det_workspace(size_t bytes, void** out) {
    int dev = 0;
    IE_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) IE_FAIL(IE_ERR_UNSUPPORTED, "deterministic workspace: device ordinal %d", dev);
    if (bytes > g_det_bytes[dev]) {
        IE_HIP_CHECK(hipDeviceSynchronize());
        if (g_det_ws[dev]) (void)hipFree(g_det_ws[dev]);
        g_det_ws[dev] = nullptr; g_det_bytes[dev] = 0;
        const size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&g_det_ws[dev], want);
        if (e != hipSuccess) { ie_set_error("deterministic workspace: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return IE_ERR_NOMEM; }
        g_det_bytes[dev] = want;
    }
    *out = g_det_ws[dev];
    return IE_OK;
}

// ---- dynamic-LDS grants --------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-(device, function) attribute: a flag per function would skip the grant on the second
// GPU of a process and that launch fails with "invalid value".  One table for the whole library, keyed by the current device ordinal.
int ie_grant_lds(const void* fn, int bytes) {
    struct Grant { const void* fn; int dev; int bytes; };
    static Grant table[512];
    static int count = 0;
    static std::mutex mu;
    int dev = 0;
    IE_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    Grant* slot = nullptr;
    for (int i = 0; i < count; ++i)
        if (table[i].fn == fn && table[i].dev == dev) { slot = &table[i]; break; }
    if (slot && slot->bytes >= bytes) return IE_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) IE_FAIL(IE_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) on device %d: %s", bytes, dev, hipGetErrorString(e));
    if (slot) slot->bytes = bytes;
    else if (count < 512) table[count++] = Grant{fn, dev, bytes};      // a full table only means the attribute is set again next time
    return IE_OK;
}

IE_EXPORT int ie_device_info(int device, char* name, int cap, int* cu, size_t* hbm_bytes) {
    hipDeviceProp_t prop;
    IE_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && cap > 0) { strncpy(name, prop.gcnArchName, cap - 1); name[cap - 1] = 0; }
    if (cu) *cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return IE_OK;
}

namespace {

template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

// NCHW fp32 -> NHWC T (channel pad).  Thread = one pixel; reads are coalesced per plane,
// writes are Cp*sizeof(T) contiguous bytes per thread.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int Cp, size_t plane,
                                    size_t total_px) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    const float* s = src + n * C * plane + pix;
    T* d = dst + i * Cp;
    for (int c = 0; c < Cp; ++c) d[c] = from_f32<T>(c < C ? s[(size_t)c * plane] : 0.f);
}

// The 16-bit image layout every matrix-core model starts from: up to 8 channels (RGB + zeros) = one 16-byte pixel, one store per thread, so
// a store instruction covers 1 KiB of whole lines (four pixels per thread = 16-byte pieces 64 bytes apart per instruction measured 4x slower).
template <typename T>
__global__ void nchw_to_nhwc8_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, size_t plane, size_t total_px) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    const float* s = src + n * C * plane + pix;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = c < C ? s[(size_t)c * plane] : 0.f;
    *(ie_u32x4*)(dst + i * 8) = ie_u32x4{ie_pack2<T>(v[0], v[1]), ie_pack2<T>(v[2], v[3]), ie_pack2<T>(v[4], v[5]), ie_pack2<T>(v[6], v[7])};
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int Cp, size_t plane,
                                    size_t total_px) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    const T* s = src + i * Cp;
    float* d = dst + n * C * plane + pix;
    for (int c = 0; c < C; ++c) d[(size_t)c * plane] = to_f32<T>(s[c]);
}

__global__ void f32_to_u8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int C, size_t plane,
                                 size_t total_px, int mode) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    const float* s = src + n * C * plane + pix;
    unsigned char* d = dst + i * C;
    for (int c = 0; c < C; ++c) {
        const float v = s[(size_t)c * plane];
        if (mode == 0) d[c] = ie_quant_u8(v);
        else d[c] = (unsigned char)fminf(fmaxf(v * 255.f, 0.f), 255.f);
    }
}

__global__ void u8_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int C, size_t plane,
                                 size_t total_px) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    const unsigned char* s = src + i * C;
    float* d = dst + n * C * plane + pix;
    for (int c = 0; c < C; ++c) d[(size_t)c * plane] = (float)s[c] / 255.0f;
}

__global__ void le_curve_kernel(const float* __restrict__ img, const float* __restrict__ adj, int Cp,
                                float* __restrict__ out, size_t plane, size_t total_px, int iters, int shared) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    const float* a = adj + i * Cp;
    for (int c = 0; c < 3; ++c) {
        float y = img[(n * 3 + c) * plane + pix];
        for (int k = 0; k < iters; ++k) y = ie_curve_step(y, a[shared ? c : 3 * k + c]);
        out[(n * 3 + c) * plane + pix] = y;
    }
}

// 8 iterations x 3 maps, not shared (the reference configuration): the pixel's 24 curve values as six 16-byte loads
__global__ void le_curve8_kernel(const float* __restrict__ img, const float* __restrict__ adj, float* __restrict__ out, size_t plane,
                                 size_t total_px) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_px) return;
    const size_t n = i / plane, pix = i % plane;
    float r[24];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const ie_f32x4 v = *(const ie_f32x4*)(adj + i * 24 + j * 4);
        r[4 * j] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float y = img[(n * 3 + c) * plane + pix];
#pragma unroll
        for (int k = 0; k < 8; ++k) y = ie_curve_step(y, r[3 * k + c]);
        out[(n * 3 + c) * plane + pix] = y;
    }
}

// Bilinear resize, NCHW fp32, F.interpolate(mode="bilinear") semantics (what kornia's resize
// inside VisionModel.infer reduces to, core/image/geometry/transform.py:216-223, and
// nn.Upsample(bilinear, align_corners=True), llunetpp.py:182).
//   align_corners = 0: src = max(0, (dst + 0.5) * in/out - 0.5);  1: src = dst * (in-1)/(out-1)
__global__ void resize_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int NC, int Hi, int Wi,
                                       int Ho, int Wo, int align_corners) {
    const size_t total = (size_t)NC * Ho * Wo;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xo = (int)(i % Wo);
    const int yo = (int)((i / Wo) % Ho);
    const size_t nc = i / ((size_t)Wo * Ho);
    float sy, sx;
    if (align_corners) {
        const float ry = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
        const float rx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
        sy = ry * yo; sx = rx * xo;
    } else {
        const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
        sy = fmaxf(ry * (yo + 0.5f) - 0.5f, 0.f);
        sx = fmaxf(rx * (xo + 0.5f) - 0.5f, 0.f);
    }
    const int y0 = min((int)sy, Hi - 1), x0 = min((int)sx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p = src + nc * (size_t)Hi * Wi;
    const float v00 = p[(size_t)y0 * Wi + x0], v01 = p[(size_t)y0 * Wi + x1];
    const float v10 = p[(size_t)y1 * Wi + x0], v11 = p[(size_t)y1 * Wi + x1];
    dst[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

}  // namespace

static inline unsigned grid_for(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

IE_EXPORT int ie_nchw_f32_to_nhwc(const float* src, void* dst, ie_dtype dt, int N, int C, int H, int W, int Cp, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) IE_FAIL(IE_ERR_INVALID, "ie_nchw_f32_to_nhwc: bad args");
    const size_t plane = (size_t)H * W, total = plane * N;
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
    case IE_F32: hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid_for(total, 256), 256, 0, st, src, (float*)dst, C, Cp, plane, total); break;
    case IE_F16:
        if (Cp == 8) hipLaunchKernelGGL(nchw_to_nhwc8_kernel<_Float16>, grid_for(total, 256), 256, 0, st, src, (_Float16*)dst, C, plane, total);
        else hipLaunchKernelGGL(nchw_to_nhwc_kernel<_Float16>, grid_for(total, 256), 256, 0, st, src, (_Float16*)dst, C, Cp, plane, total);
        break;
    case IE_BF16:
        if (Cp == 8) hipLaunchKernelGGL(nchw_to_nhwc8_kernel<__bf16>, grid_for(total, 256), 256, 0, st, src, (__bf16*)dst, C, plane, total);
        else hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, grid_for(total, 256), 256, 0, st, src, (__bf16*)dst, C, Cp, plane, total);
        break;
    default: IE_FAIL(IE_ERR_UNSUPPORTED, "ie_nchw_f32_to_nhwc: dtype %d", (int)dt);
    }
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_nhwc_to_nchw_f32(const void* src, ie_dtype dt, float* dst, int N, int C, int H, int W, int Cp, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) IE_FAIL(IE_ERR_INVALID, "ie_nhwc_to_nchw_f32: bad args");
    const size_t plane = (size_t)H * W, total = plane * N;
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
    case IE_F32: hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid_for(total, 256), 256, 0, st, (const float*)src, dst, C, Cp, plane, total); break;
    case IE_F16: hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, grid_for(total, 256), 256, 0, st, (const _Float16*)src, dst, C, Cp, plane, total); break;
    case IE_BF16: hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, grid_for(total, 256), 256, 0, st, (const __bf16*)src, dst, C, Cp, plane, total); break;
    default: IE_FAIL(IE_ERR_UNSUPPORTED, "ie_nhwc_to_nchw_f32: dtype %d", (int)dt);
    }
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_nchw_f32_to_u8_nhwc(const float* src, uint8_t* dst, int N, int C, int H, int W, int mode, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) IE_FAIL(IE_ERR_INVALID, "ie_nchw_f32_to_u8_nhwc: bad args");
    const size_t plane = (size_t)H * W, total = plane * N;
    hipLaunchKernelGGL(f32_to_u8_kernel, grid_for(total, 256), 256, 0, (hipStream_t)stream, src, dst, C, plane, total, mode);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_u8_nhwc_to_nchw_f32(const uint8_t* src, float* dst, int N, int C, int H, int W, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0) IE_FAIL(IE_ERR_INVALID, "ie_u8_nhwc_to_nchw_f32: bad args");
    const size_t plane = (size_t)H * W, total = plane * N;
    hipLaunchKernelGGL(u8_to_f32_kernel, grid_for(total, 256), 256, 0, (hipStream_t)stream, src, dst, C, plane, total);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_le_curve(const float* img, const float* adjust_nhwc, int Cp, float* enhanced, int N, int H, int W,
                          int iters, void* stream) {
    const int shared = (Cp >= 3 && Cp < 3 * iters) ? 1 : 0;   // one 3-channel map for every iteration (zero_dcepp.py:201-203)
    if (!img || !adjust_nhwc || !enhanced || N <= 0 || H <= 0 || W <= 0 || iters <= 0 || Cp < 3)
        IE_FAIL(IE_ERR_INVALID, "ie_le_curve: bad args");
    const size_t plane = (size_t)H * W, total = plane * N;
    if (iters == 8 && Cp == 24 && !shared)
        hipLaunchKernelGGL(le_curve8_kernel, grid_for(total, 256), 256, 0, (hipStream_t)stream, img, adjust_nhwc, enhanced, plane, total);
    else
        hipLaunchKernelGGL(le_curve_kernel, grid_for(total, 256), 256, 0, (hipStream_t)stream, img, adjust_nhwc, Cp, enhanced,
                           plane, total, iters, shared);
    IE_LAUNCH_CHECK();
    return IE_OK;
}

IE_EXPORT int ie_resize_bilinear_nchw_f32(const float* src, float* dst, int N, int C, int Hi, int Wi, int Ho, int Wo,
                                          int align_corners, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0)
        IE_FAIL(IE_ERR_INVALID, "ie_resize_bilinear_nchw_f32: bad args");
    const size_t total = (size_t)N * C * Ho * Wo;
    hipLaunchKernelGGL(resize_bilinear_kernel, grid_for(total, 256), 256, 0, (hipStream_t)stream, src, dst, N * C, Hi, Wi,
                       Ho, Wo, align_corners);
    IE_LAUNCH_CHECK();
    return IE_OK;
}
