// Practical MFMA ceiling of this MI355X: register-only chains of v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16
// (no LDS, no memory), 8 independent accumulators per wave, 1 / 2 / 4 waves per SIMD.  Prints TFLOP/s over a ~10 ms run,
// i.e. under the sustained power / clock state the fused kernels see.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) {      // pseudo-random operands in (-1, 1): all-zero / trivial data runs at a higher clock than real data
        unsigned h = (threadIdx.x * 8 + k + blockIdx.x * 2048) * 2654435761u;
        a[k] = (__bf16)((int)(h >> 8 & 0xffff) * (1.f / 32768.f) - 1.f);
        b[k] = (__bf16)((int)(h >> 12 & 0xffff) * (1.f / 32768.f) - 1.f);
    }
    // operands differ per accumulator and the loop is fully explicit, so the body is exactly N MFMAs + loop control
    // (inline asm with VGPR constraints: the builtin form made the compiler rotate AGPRs through VGPRs every iteration
    // and the 16x16 loop read 35 % low)
    if constexpr (SHAPE == 16) {
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c5) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c6) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c7) : "v"(a), "v"(b));
        }
        const f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
        if (s[0] + s[3] == 12345.678f) out[0] = s[1];
    } else {
        f32x16 c0, c1, c2, c3;
        for (int k = 0; k < 16; ++k) { c0[k] = 0.f; c1[k] = 0.f; c2[k] = 0.f; c3[k] = 0.f; }
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
        }
        const f32x16 s = c0 + c1 + c2 + c3;
        if (s[0] + s[15] == 12345.678f) out[0] = s[1];
    }
}

template <int SHAPE>
static void run(int waves_per_simd, float* d) {
    const int blocks = 256 * waves_per_simd;          // 256 CUs x (256 threads = 4 waves = 1 per SIMD) x waves_per_simd
    const int iters = 40000 / waves_per_simd;
    const double flop_per_iter = SHAPE == 16 ? 8.0 * 16 * 16 * 32 * 2 : 4.0 * 32 * 32 * 16 * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, d, iters);      // warm-up
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 4 * iters * flop_per_iter;
    printf("{\"mfma\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.2f, \"tflops\": %.1f}\n", SHAPE == 16 ? "16x16x32_bf16" : "32x32x16_bf16",
           waves_per_simd, ms / 5, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    for (int w : {1, 2, 4}) run<16>(w, d);
    for (int w : {1, 2, 4}) run<32>(w, d);
    return 0;
}
