// Shared host/device helpers for the gfx950 image-enhancement library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ie_hip.h"

#define IE_EXPORT extern "C" __attribute__((visibility("default")))

// ---- error plumbing (thread-local message, negative status codes) ----------
void ie_set_error(const char* fmt, ...);
#define IE_FAIL(code, ...) do { ie_set_error(__VA_ARGS__); return (code); } while (0)
#define IE_HIP_CHECK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { \
    ie_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    return IE_ERR_HIP; } } while (0)
#define IE_LAUNCH_CHECK() IE_HIP_CHECK(hipGetLastError())
// Raise a kernel's dynamic-LDS limit to `bytes` on the CURRENT device (once per (device, function); ie_core.hip).  Returns IE_OK or a status.
int ie_grant_lds(const void* fn, int bytes);
#define IE_GRANT_LDS(fn, bytes) do { int s__ = ie_grant_lds((const void*)(fn), (int)(bytes)); if (s__ != IE_OK) return s__; } while (0)

static inline size_t ie_dtype_size(int dt) {
    switch (dt) { case IE_F32: return 4; case IE_F16: case IE_BF16: return 2; case IE_U8: return 1; default: return 0; }
}
static inline int ie_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device vector types -----------------------------------------------------
typedef __attribute__((ext_vector_type(8))) _Float16 ie_half8;
typedef __attribute__((ext_vector_type(8))) __bf16 ie_bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 ie_half4;
typedef __attribute__((ext_vector_type(4))) __bf16 ie_bf16x4;
typedef __attribute__((ext_vector_type(16))) float ie_f32x16;
typedef __attribute__((ext_vector_type(4))) float ie_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int ie_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int ie_u32x2;
typedef __attribute__((ext_vector_type(8))) unsigned short ie_u16x8;

#ifdef __HIPCC__
// Element-type traits for the 16-bit MFMA paths.
template <typename T> struct ie_elem;
template <> struct ie_elem<_Float16> {
    using frag = ie_half8;
    static constexpr int dtype = IE_F16;
    static __device__ __forceinline__ ie_f32x16 mfma32(frag a, frag b, ie_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ ie_f32x4 mfma16(frag a, frag b, ie_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct ie_elem<__bf16> {
    using frag = ie_bf16x8;
    static constexpr int dtype = IE_BF16;
    static __device__ __forceinline__ ie_f32x16 mfma32(frag a, frag b, ie_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ ie_f32x4 mfma16(frag a, frag b, ie_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

// MFMA on raw 16-byte fragments: kernels whose operands mix fp16 and bf16 sources (the Zero-DCE "bf16" schedule
// keeps the precision-critical tensors in fp16) hold every fragment as four dwords and name the
// element type at the instruction.
template <typename T> __device__ __forceinline__ ie_f32x4 ie_mfma16_raw(ie_u32x4 a, ie_u32x4 b, ie_f32x4 c) {
    using frag = typename ie_elem<T>::frag;
    return ie_elem<T>::mfma16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c);
}

// Pack two floats into one dword of 2 x T: one v_cvt_pk_{bf16,f16}_f32 (round-to-nearest-even).
typedef __attribute__((ext_vector_type(2))) float ie_f32x2;
typedef __attribute__((ext_vector_type(2))) short ie_s16x2;
template <typename T> __device__ __forceinline__ unsigned ie_pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) T t2;
    const ie_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, t2));
}
// the two halves of a dword of 2 x T as floats
template <typename T> __device__ __forceinline__ float ie_unpack_lo(unsigned v) {
    typedef __attribute__((ext_vector_type(2))) T t2;
    return (float)__builtin_bit_cast(t2, v)[0];
}
template <typename T> __device__ __forceinline__ float ie_unpack_hi(unsigned v) {
    typedef __attribute__((ext_vector_type(2))) T t2;
    return (float)__builtin_bit_cast(t2, v)[1];
}
// ... followed by ReLU on the packed pair: negative 16-bit floats have the sign bit set, i.e. are
// negative as int16, so ReLU == v_pk_max_i16(x, 0) (-0.0 -> +0.0; 2 instructions per 2 outputs).
template <typename T> __device__ __forceinline__ unsigned ie_pack2_relu(float lo, float hi) {
    ie_s16x2 s = __builtin_bit_cast(ie_s16x2, ie_pack2<T>(lo, hi));
    const ie_s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(s, z));
}

// MODE.FP16_OVFL (hwreg MODE bit 23): a conversion to fp16 that overflows gives +-65504 instead of +-inf for the rest of the wave's life
// (true infinities stay infinite; tools/micro/fp16_ovfl.hip checks v_cvt_pk_f16_f32 and v_cvt_f16_f32 on gfx950).  The Zero-DCE plan's
// "bf16" schedule stores x1 / x2 / x6 / P in fp16 for precision; with this bit an out-of-range activation saturates like a bf16 one would
// merely lose precision, instead of poisoning everything behind it with inf - inf = NaN.  One scalar instruction per wave, no cost per value.
__device__ __forceinline__ void ie_fp16_saturate_on() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }

// `s_waitcnt vmcnt` takes an immediate.  A wave-uniform count that is only known at run time used to go through a 30-way switch -- a
// compare / branch tree of 10-15 scalar branches in EVERY step of the strip kernels.  The steady-state value of a loop is a compile-time
// constant per wave class, so: test the one or two hot constants first and fall back to the caller's switch (prologue steps) otherwise.
template <int K> __device__ __forceinline__ void ie_wait_vm_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }
template <int H0, int H1 = -1, int H2 = -1, typename F> __device__ __forceinline__ void ie_wait_vm_hot(int n, F&& cold) {
    if (n < 0) return;                       // this wave has nothing in flight to wait for
    if (n == H0) { ie_wait_vm_imm<(H0 < 0 ? 0 : (H0 > 63 ? 63 : H0))>(); return; }
    if constexpr (H1 >= 0) { if (n == H1) { ie_wait_vm_imm<(H1 > 63 ? 63 : (H1 < 0 ? 0 : H1))>(); return; } }
    if constexpr (H2 >= 0) { if (n == H2) { ie_wait_vm_imm<(H2 > 63 ? 63 : (H2 < 0 ? 0 : H2))>(); return; } }
    cold(n);
}

// tanh as 1 - 2/(exp(2x)+1) on the hardware exp2/rcp units: absolute error
// <= ~3e-7 (the LE curve consumes r as a multiplier, so absolute error is what
// matters); saturates correctly to +-1.
__device__ __forceinline__ float ie_tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f); // exp(2x)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float ie_sigmoid_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(-x * 1.4426950408889634f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// One LE-curve step in the reference's fp32 association order
// (zero_dce.py:187: y + r * (pow(y,2) - y)); no FMA contraction.
__device__ __forceinline__ float ie_curve_step(float y, float r) {
#pragma clang fp contract(off)
    const float t = y * y;
    const float u = t - y;
    const float v = r * u;
    return y + v;
}
// The same step as two FMAs (y*y - y, then r*u + y): what the 16-bit Zero-DCE kernels use -- their r already carries 16-bit
// rounding (1e-4 .. 1e-3), so the reference's rounding ORDER (<= 1 ulp of fp32 per step) is not what limits them, and the curve's
// VALU work competes with the matrix pipe for issue slots (8 steps x 3 channels per pixel: 96 -> 48 instructions).
__device__ __forceinline__ float ie_curve_step_fma(float y, float r) {
    const float u = __builtin_fmaf(y, y, -y);
    return __builtin_fmaf(r, u, y);
}
// torchvision save_image rule: clamp(x*255+0.5, 0, 255) -> u8 (truncate).
__device__ __forceinline__ unsigned char ie_quant_u8(float v) {
#pragma clang fp contract(off)
    float q = v * 255.0f;
    q = q + 0.5f;
    q = fminf(fmaxf(q, 0.0f), 255.0f);
    return (unsigned char)q;
}
// activation and its derivative expressed in the OUTPUT y (what the backward pass keeps)
__device__ __forceinline__ float ie_act_apply(float v, int act, float slope) {
    switch (act) {
    case IE_ACT_RELU: return fmaxf(v, 0.f);
    case IE_ACT_LRELU: return v > 0.f ? v : v * slope;
    case IE_ACT_TANH: return tanhf(v);
    case IE_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
    }
}
__device__ __forceinline__ float ie_act_grad_from_output(float y, int act, float slope) {
    switch (act) {
    case IE_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case IE_ACT_LRELU: return y > 0.f ? 1.f : slope;
    case IE_ACT_TANH: return 1.f - y * y;
    case IE_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
    }
}
#endif
