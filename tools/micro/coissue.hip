// Do the matrix and the vector pipe of one SIMD overlap ACROSS its two waves on gfx950?  Register-only loops, one 512-thread
// workgroup per CU (waves w and w+4 share a SIMD), cycles per iteration from s_memtime (wave 0 of each block, averaged).
//   mode 0  waves 0-3: 36 MFMAs per iteration, waves 4-7 idle
//   mode 1  waves 0-3 idle, waves 4-7: NV VALU ops per iteration (v_fma_f32 / v_cvt_pk_f16_f32 mix)
//   mode 2  waves 0-3: 36 MFMAs, waves 4-7: NV VALU ops, concurrently (what the two-stage strip kernels rely on)
//   mode 3  every wave: 18 MFMAs with NV/2 VALU ops interleaved in its own stream, 2 or 4 per MFMA
//   mode 4  every wave: 18 MFMAs, then NV/2 VALU ops (bursts, as an epilogue after an MFMA phase)
//   mode 5  every wave: 18 MFMAs (matrix beside matrix)
// with and without an s_barrier per iteration.  Build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA(c) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define VFMA(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(u), "v"(v))
#define VCVT(r, x, y) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y))

template <int NV>
__device__ __forceinline__ void valu_burst(float (&x)[8], unsigned (&r)[4], float u, float v) {
#pragma unroll
    for (int k = 0; k < NV / 12; ++k) {
        VFMA(x[0]); VFMA(x[1]); VFMA(x[2]); VFMA(x[3]); VFMA(x[4]); VFMA(x[5]); VFMA(x[6]); VFMA(x[7]);
        VCVT(r[0], x[0], x[1]); VCVT(r[1], x[2], x[3]); VCVT(r[2], x[4], x[5]); VCVT(r[3], x[6], x[7]);
    }
}

template <int MODE, int NV, bool BARRIER>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        unsigned h = (threadIdx.x * 8 + i + blockIdx.x * 4096) * 2654435761u;
        a[i] = (_Float16)((int)(h >> 8 & 0xffff) * (1.f / 32768.f) - 1.f);
        b[i] = (_Float16)((int)(h >> 12 & 0xffff) * (1.f / 32768.f) - 1.f);
    }
    f32x4 c[12];
    for (int i = 0; i < 12; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x[8];
    unsigned r[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) x[i] = (float)a[i];
    const float u = 0.999f, v = 1e-3f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool roleB = wave >= 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
        if constexpr (MODE == 0 || MODE == 2) {
            if (!roleB) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 12; ++i) MFMA(c[i]);
            } else if (MODE == 2) {
                valu_burst<NV>(x, r, u, v);
            }
        } else if constexpr (MODE == 1) {
            if (roleB) valu_burst<NV>(x, r, u, v);
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                MFMA(c[i % 12]);
#pragma unroll
                for (int j = 0; j < NV / 2 / 18; ++j) { if ((j & 1) == 0) VFMA(x[(2 * i + j) & 7]); else VCVT(r[(i + j) & 3], x[(i + j) & 7], x[(i + j + 1) & 7]); }
            }
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 18; ++i) MFMA(c[i % 12]);
            valu_burst<NV / 2>(x, r, u, v);
        } else {
#pragma unroll
            for (int i = 0; i < 18; ++i) MFMA(c[i % 12]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = c[0];
    for (int i = 1; i < 12; ++i) s += c[i];
    float xs = 0.f;
    for (int i = 0; i < 8; ++i) xs += x[i];
    if (s[0] + s[3] + xs == 12345.678f) out[0] = s[1] + (float)(r[0] ^ r[1] ^ r[2] ^ r[3]);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NV, bool BARRIER>
static void run(const char* what, float* d, unsigned long long* dc) {
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL((k<MODE, NV, BARRIER>), dim3(blocks), dim3(512), 0, 0, d, dc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV, BARRIER>), dim3(blocks), dim3(512), 0, 0, d, dc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    // s_memtime counts at 100 MHz on gfx950: report the wall time per iteration in ns and in cycles of the clock the run held
    printf("mode %d NV %3d barrier %d  %-58s %8.1f ns/iter  (s_memtime ticks/iter %.2f)\n", MODE, NV, (int)BARRIER, what, ms * 1e6 / iters, s / blocks / iters);
}

int main() {
    float* d; unsigned long long* dc;
    hipMalloc(&d, 4096); hipMalloc(&dc, 256 * 8);
    run<0, 72, false>("A: 36 MFMA, B idle", d, dc);
    run<1, 72, false>("A idle, B: 72 VALU", d, dc);
    run<1, 144, false>("A idle, B: 144 VALU", d, dc);
    run<2, 72, false>("A: 36 MFMA || B: 72 VALU", d, dc);
    run<2, 144, false>("A: 36 MFMA || B: 144 VALU", d, dc);
    run<2, 72, true>("A: 36 MFMA || B: 72 VALU", d, dc);
    run<2, 144, true>("A: 36 MFMA || B: 144 VALU", d, dc);
    run<5, 72, false>("both: 18 MFMA", d, dc);
    run<3, 72, false>("both: 18 MFMA with 2 VALU after each", d, dc);
    run<3, 144, false>("both: 18 MFMA with 4 VALU after each", d, dc);
    run<4, 72, false>("both: 18 MFMA then 36 VALU", d, dc);
    run<4, 144, false>("both: 18 MFMA then 72 VALU", d, dc);
    run<4, 72, true>("both: 18 MFMA then 36 VALU", d, dc);
    run<4, 144, true>("both: 18 MFMA then 72 VALU", d, dc);
    return 0;
}
