// Store-pattern microbenchmark (MI355X): what HBM write rate do the epilogue stores of the Zero-DCE strip kernels get?
//   pattern 0: every wave instruction writes 1 KiB contiguous (lane * 16 B)                       -- the device's plain-store rate
//   pattern 1: the strip kernels' WIDE epilogue: a wave instruction writes 32 pixels x 32 contiguous bytes (half of each
//              pixel's 64-byte NHWC line); the other half comes from another wave of the same workgroup
//   pattern 2: pattern 1 with both halves written by the SAME wave in two back-to-back instructions
//   pattern 3: 64 B per pixel written as 4 lanes x 16 B by one instruction (16 pixels x 64 B = 1 KiB contiguous) = pattern 0 layout
// Each variant also with nontemporal stores (nt = 1).   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int PAT, int NT>
__global__ __launch_bounds__(256) void k(char* dst, size_t npix_per_wg, int iters_unused) {
    // workgroup = 4 waves; it owns npix_per_wg pixels of 64 B (a "row strip"), walked 128 pixels per step like the conv kernels
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    char* base = dst + (size_t)blockIdx.x * npix_per_wg * 64;
    const u32x4 v = {(unsigned)t, 1u, 2u, 3u};
    const int p16 = lane & 15, kq = lane >> 4, ph = wave & 1, ch = wave >> 1;
    for (size_t p0 = 0; p0 < npix_per_wg; p0 += 128) {
        char* row = base + p0 * 64;
        if (PAT == 0 || PAT == 3) {
            // 128 px x 64 B = 8 KiB per step = 8 instructions of 1 KiB: 2 per wave
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4* q = (u32x4*)(row + (size_t)(2 * wave + j) * 1024 + lane * 16);
                if (NT) __builtin_nontemporal_store(v, q); else *q = v;
            }
        } else if (PAT == 1) {
#pragma unroll
            for (int nt = 0; nt < 4; nt += 2) {
                const int q = 64 * ph + 16 * (nt + (kq & 1)) + p16;
                u32x4* a = (u32x4*)(row + (size_t)q * 64 + (16 * ch + 4 * (kq & 2)) * 2);
                if (NT) __builtin_nontemporal_store(v, a); else *a = v;
            }
        } else {   // PAT 2: wave (ph, ch) writes BOTH halves of the pixels of column quarter (2*ph + ch): 32 px x 64 B in two instructions
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q = 32 * (2 * ph + ch) + 16 * (kq & 1) + p16;
                u32x4* a = (u32x4*)(row + (size_t)q * 64 + 32 * h + 8 * (kq & 2));
                if (NT) __builtin_nontemporal_store(v, a); else *a = v;
            }
        }
    }
}

template <int PAT, int NT> float run(char* d, size_t bytes, int wgs) {
    const size_t npix = bytes / 64 / wgs / 128 * 128;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<PAT, NT>), dim3(wgs), dim3(256), 0, 0, d, npix, 0);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<PAT, NT>), dim3(wgs), dim3(256), 0, 0, d, npix, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double gbs = (double)npix * 64 * wgs * reps / (ms * 1e-3) / 1e9;
    printf("pattern %d nt %d wgs %5d: %8.1f GB/s\n", PAT, NT, wgs, gbs);
    return (float)gbs;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    char* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int wgs : {512, 1024, 2048, 8192}) {
        run<0, 0>(d, bytes, wgs); run<0, 1>(d, bytes, wgs);
        run<1, 0>(d, bytes, wgs); run<1, 1>(d, bytes, wgs);
        run<2, 0>(d, bytes, wgs); run<2, 1>(d, bytes, wgs);
    }
    hipFree(d);
    return 0;
}
