// What pace do the strip kernels' MFMA streams reach when the two waves of a SIMD stream TOGETHER, with the LDS fragment reads the real kernels carry?
// One 512-thread workgroup per CU (waves w and w + 4 share a SIMD, as in K345 / K67X); every wave runs the kernels' own row stream: 12 ds_read_b128
// fragments of a 32-channel row image (XOR-swizzled pixel lines) -> 36 v_mfma_f32_16x16x32_f16 into three rotating accumulator sets, two rows per step
// (72 MFMAs per wave and step).  Reported: wall ns per step and per MFMA of a SIMD, and the s_memtime span of a stage-A and a stage-B wave.
//   pair 0  waves 0-3 stream, waves 4-7 idle            pair 1  all stream, no barrier
//   pair 2  all stream, one s_barrier per step          pair 3  as 2, waves 4-7 half a step behind (their barrier sits between their two rows)
//   form 0  the kernels' pinned form, fragments read AHEAD = 3 ahead       form 1  AHEAD = 6        form 2  all 12 fragments of a row read up front
//   form 3  register-only (no LDS reads: the same fragment registers every time)        forms 4-8: see mfma_row / mfma_row_acc / mfma_row_both
//   chores  0 none   2 as 1 with stage A's tail moved to the head of its next step   1 stage A ends its step with a latency-bound tail (LDS read -> pack -> 4 ds_write_b128 -> 6 LDS-DMA pieces), stage B opens its step with
//           a 96-operation transcendental burst -- the shape of K67X
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value [-DUSE_BF16] [-DRELU_DATA] mfma_pair.hip -o mfma_pair     (RELU_DATA: half of the LDS values zero, as behind a ReLU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifdef USE_BF16
typedef __bf16 elem_t;
#define MFMA_NAME "v_mfma_f32_16x16x32_bf16"
#define MFMA_BUILTIN __builtin_amdgcn_mfma_f32_16x16x32_bf16
#else
typedef _Float16 elem_t;
#define MFMA_NAME "v_mfma_f32_16x16x32_f16"
#define MFMA_BUILTIN __builtin_amdgcn_mfma_f32_16x16x32_f16
#endif
typedef __attribute__((ext_vector_type(8))) elem_t f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr;
#define SB() __builtin_amdgcn_sched_barrier(0)
constexpr int PXB = 64, ROWH_B = 130 * PXB, NR = 6;

template <int FORM>
__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (FORM == 3) {      // (loop-invariant operands: keep the compiler from hoisting / merging the register-only stream)
        f32x4 d;
        asm volatile(MFMA_NAME " %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    } else {
        return MFMA_BUILTIN(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
}

template <int FORM>
__device__ __forceinline__ void mfma_row(const char* slot, const int (&base)[3], const u32x4* wv, f32x4 (&accN)[4], f32x4 (&accM)[4], f32x4 (&accO)[4], bool first, const f32x4& init,
                                         u32x4 (&keep)[12]) {
    // forms 4-6: as form 0 (3 ahead) with the fragment registers kept away from the MFMAs that just read them: LAG = how many groups a fragment's registers stay
    // reserved after its last MFMA (an empty asm use), so that the ds_read of a group does not target the B operand of the MFMAs in flight; form 5 / 6: the read
    // sits between the group's second and third MFMA
    constexpr int AHEAD = FORM == 1 ? 6 : FORM == 2 ? 12 : 3;
    constexpr int LAG = FORM == 4 ? 2 : FORM == 6 ? 2 : 0;
    constexpr bool MID = FORM == 5 || FORM == 6;
    u32x4 bx[12];
    const char* bk[3] = {slot + base[0], slot + base[1], slot + base[2]};
    auto rd = [&](int g) __attribute__((always_inline)) {
        if constexpr (FORM == 3) bx[g] = keep[g];
        else bx[g] = *(const u32x4*)(bk[g >> 2] + 1024 * (g & 3));
    };
#pragma unroll
    for (int g = 0; g < AHEAD && g < 12; ++g) rd(g);
    SB();
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int kx = g >> 2, j = g & 3;
        accN[j] = mfma<FORM>(wv[0 * 3 + kx], bx[g], (first && kx == 0) ? init : accN[j]);
        if (!MID && g + AHEAD < 12) rd(g + AHEAD);
        SB();
        accM[j] = mfma<FORM>(wv[1 * 3 + kx], bx[g], accM[j]);
        if (MID && g + AHEAD < 12) rd(g + AHEAD);
        SB();
        accO[j] = mfma<FORM>(wv[2 * 3 + kx], bx[g], accO[j]);
        if constexpr (LAG > 0) { if (g >= LAG) asm volatile("" ::"v"(bx[g - LAG])); }
        SB();
    }
}

// form 8: a wave that owns BOTH 16-cout fragments of its layer over a quarter of the strip (2 pixel tiles instead of 4): every fragment feeds 6 MFMAs instead of 3, so a
// row's 36 MFMAs need 6 ds_read_b128 instead of 12 (the price in a real kernel: 18 more weight fragments = 72 VGPRs).  Same accumulators (12), same MFMA count.
__device__ __forceinline__ void mfma_row_both(const char* slot, const int (&base)[3], const u32x4* wv, f32x4 (&accN)[4], f32x4 (&accM)[4], f32x4 (&accO)[4], bool first,
                                              const f32x4& init) {
    u32x4 bx[6];
    const char* bk[3] = {slot + base[0], slot + base[1], slot + base[2]};
    auto rd = [&](int g) __attribute__((always_inline)) { bx[g] = *(const u32x4*)(bk[g >> 1] + 1024 * (g & 1)); };
    rd(0); rd(1);
    SB();
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int kx = g >> 1, j = g & 1;       // accumulator j: cout fragment 0, j + 2: cout fragment 1
        accN[j] = mfma<0>(wv[0 * 3 + kx], bx[g], (first && kx == 0) ? init : accN[j]);
        if (g + 2 < 6) rd(g + 2);
        SB();
        accM[j] = mfma<0>(wv[1 * 3 + kx], bx[g], accM[j]); SB();
        accO[j] = mfma<0>(wv[2 * 3 + kx], bx[g], accO[j]); SB();
        accN[j + 2] = mfma<0>(wv[9 + 0 * 3 + kx], bx[g], (first && kx == 0) ? init : accN[j + 2]); SB();
        accM[j + 2] = mfma<0>(wv[9 + 1 * 3 + kx], bx[g], accM[j + 2]); SB();
        accO[j + 2] = mfma<0>(wv[9 + 2 * 3 + kx], bx[g], accO[j + 2]); SB();
    }
}

// form 7: the same stream with the fragments in AccVGPRs (ds_read_b128 a[..] / MFMA B operand from a[..]; inline asm, counted lgkmcnt by hand): does the LDS return
// path cost less when it does not write the architectural VGPR banks the MFMAs in flight read their A operand and accumulators from?
__device__ __forceinline__ void mfma_row_acc(const char* slot, const int (&base)[3], const u32x4* wv, f32x4 (&accN)[4], f32x4 (&accM)[4], f32x4 (&accO)[4], bool first,
                                             const f32x4& init) {
    u32x4 bx[12];
    const unsigned ad[3] = {(unsigned)(size_t)(slot + base[0]), (unsigned)(size_t)(slot + base[1]), (unsigned)(size_t)(slot + base[2])};
#define RD(g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(bx[g]) : "v"(ad[(g) >> 2]), "n"(1024 * ((g) & 3)))
#define MF(d, a, b, c) asm volatile(MFMA_NAME " %0, %1, %2, %3" : "=v"(d) : "v"(a), "a"(b), "v"(c))
    RD(0); RD(1); RD(2);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int kx = g >> 2, j = g & 3;
        if (g <= 9) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (g == 10) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (first && kx == 0) MF(accN[j], wv[0 * 3 + kx], bx[g], init); else MF(accN[j], wv[0 * 3 + kx], bx[g], accN[j]);
        if (g + 3 < 12) { switch (g + 3) { case 3: RD(3); break; case 4: RD(4); break; case 5: RD(5); break; case 6: RD(6); break; case 7: RD(7); break; case 8: RD(8); break;
                                           case 9: RD(9); break; case 10: RD(10); break; default: RD(11); break; } }
        MF(accM[j], wv[1 * 3 + kx], bx[g], accM[j]);
        MF(accO[j], wv[2 * 3 + kx], bx[g], accO[j]);
    }
#undef RD
#undef MF
}

template <int PAIR, int FORM, int CHORES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float* out, unsigned long long* cyc, const float* src, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[NR * ROWH_B + 8192];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;
    const int stg = wave >> 2, w4 = wave & 3, hf = w4 >> 1;
    for (int i = tid; i < (NR * ROWH_B + 8192) / 4; i += 512) {
        unsigned h = (unsigned)(i + blockIdx.x * 7919) * 2654435761u;
        const elem_t a = (elem_t)((int)(h >> 8 & 0xffff) * (1.f / 32768.f) - 1.f), b = (elem_t)((int)(h >> 12 & 0xffff) * (1.f / 32768.f) - 1.f);
#ifdef RELU_DATA
        ((unsigned*)smem)[i] = ((float)a > 0.f ? (unsigned)__builtin_bit_cast(unsigned short, a) : 0u) | (((float)b > 0.f ? (unsigned)__builtin_bit_cast(unsigned short, b) : 0u) << 16);
#else
        ((unsigned*)smem)[i] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
#endif
    }
    u32x4 wv[18], keep[12];
    for (int f = 0; f < 18; ++f)
        for (int e = 0; e < 4; ++e) {
            unsigned h = (unsigned)((wave * 18 + f) * 256 + lane * 4 + e) * 2246822519u;
            const elem_t a = (elem_t)(((int)(h >> 8 & 0xffff) * (1.f / 32768.f) - 1.f) * 0.05f), b = (elem_t)(((int)(h >> 12 & 0xffff) * (1.f / 32768.f) - 1.f) * 0.05f);
            wv[f][e] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
        }
    int base[3];
    for (int kx = 0; kx < 3; ++kx) {
        const int P = 64 * hf + p16 + kx;
        base[kx] = P * PXB + ((kq ^ ((P >> 1) & 3)) << 4);
    }
    __syncthreads();
    for (int g = 0; g < 12; ++g) keep[g] = *(const u32x4*)(smem + base[g >> 2] + 1024 * (g & 3));
    const f32x4 bias = {0.01f, 0.02f, 0.03f, 0.04f};
    f32x4 a0[4], a1[4], a2[4];
    for (int j = 0; j < 4; ++j) { a0[j] = bias; a1[j] = bias; a2[j] = bias; }
    float cv[4] = {0.1f, 0.2f, 0.3f, 0.4f}, sink[4] = {0.f, 0.f, 0.f, 0.f};
    const bool active = PAIR != 0 || stg == 0;
    const auto rsS = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (active) {
        auto chores_a = [&](int t, const f32x4 (&acc)[4]) __attribute__((always_inline)) {
            if constexpr (CHORES >= 1) {
                // a staged row leaves (LDS read -> use), a finished row is packed and written, the next rows are requested
                const u32x4 v = *(const u32x4*)(smem + NR * ROWH_B + lane * 16);
                unsigned pk[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pk[2 * j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(acc[j][0], acc[j][1]));
                    pk[2 * j + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(acc[j][2], acc[j][3]));
                }
                char* d = smem + NR * ROWH_B + 1024 + w4 * 1024 + lane * 16;
                *(u32x4*)d = u32x4{pk[0] ^ v[0], pk[1], pk[2], pk[3]};
                *(u32x4*)(d + 4096 - 1024 * w4) = u32x4{pk[4] ^ v[1], pk[5], pk[6], pk[7]};
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lds_ptr)(smem + NR * ROWH_B + 1024 * (q & 3) + 4096), 16, (int)(((t * 6 + q) & 63) * 1024 + lane * 16), 0, 0, 0);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
        };
        auto chores_b = [&](const f32x4 (&acc)[4]) __attribute__((always_inline)) {
            if constexpr (CHORES >= 1) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float e = __builtin_amdgcn_exp2f(acc[j][kk]);
                        e = __builtin_amdgcn_rcpf(e + 1.0f);
                        e = __builtin_fmaf(e, -2.0f, 1.0f);
                        const float u = __builtin_fmaf(cv[j], cv[j], -cv[j]);
                        cv[j] = __builtin_fmaf(e, u, cv[j]);
                    }
                SB();
            }
        };
#define ROW(r, w, n, m, o, f) do { if constexpr (FORM == 8) mfma_row_both(r, base, wv, n, m, o, f, bias); else if constexpr (FORM == 7) mfma_row_acc(r, base, wv, n, m, o, f, bias);       /* (one row's nine weight fragments for both rows: the unified file splits 128 + 128 here) */ else mfma_row<FORM>(r, base, w, n, m, o, f, bias, keep); } while (0)
        auto step = [&](int t, f32x4 (&aN)[4], f32x4 (&aM)[4], f32x4 (&aO)[4], f32x4 (&pN)[4], f32x4 (&pM)[4], f32x4 (&pO)[4]) __attribute__((always_inline)) {
            const char* r0 = smem + ((2 * t) % NR) * ROWH_B;
            const char* r1 = smem + ((2 * t + 1) % NR) * ROWH_B;
            if (PAIR == 3 && stg == 1) {       // half a step behind: second row of the previous step, barrier, first row of this one
                ROW(r1, wv + 9, pN, pM, pO, false);      // (p* = the previous step's sets)
#pragma unroll
                for (int j = 0; j < 4; ++j) sink[j] += pO[j][j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                chores_b(pO);
                ROW(r0, wv, aN, aM, aO, true);
                return;
            } else {
                if (PAIR >= 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                if (stg == 1) chores_b(aO);
                if (CHORES == 2 && stg == 0) chores_a(t, aN);       // chores 2: stage A's hand-off work at the HEAD of its step (on the row finished a step earlier), under stage B's stream
                ROW(r0, wv, aN, aM, aO, true);
                ROW(r1, wv + 9, aN, aM, aO, false);
                if (CHORES == 1 && stg == 0) chores_a(t, aO);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sink[j] += aO[j][j];      // the finished row is consumed (keeps its MFMAs alive)
        };
        for (int t = 0; t < iters; t += 3) {
            step(t, a0, a1, a2, a1, a2, a0);
            step(t + 1, a2, a0, a1, a0, a1, a2);
            step(t + 2, a1, a2, a0, a2, a0, a1);
        }
    } else if (PAIR == 0) {
        // idle partner: leaves at once
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 s = a0[0];
    for (int j = 0; j < 4; ++j) s += a0[j] + a1[j] + a2[j];
    if (s[0] + s[1] + s[2] + s[3] + cv[0] + cv[1] + cv[2] + cv[3] + sink[0] + sink[1] + sink[2] + sink[3] == 12345.678f) out[0] = s[1];
    if (lane == 0 && (wave == 0 || wave == 4)) cyc[blockIdx.x * 2 + (wave >> 2)] = t1 - t0;
}

template <int PAIR, int FORM, int CHORES>
static void run(const char* what, float* d, unsigned long long* dc, const float* src) {
    const int iters = 30000, blocks = 256;
    hipLaunchKernelGGL((k<PAIR, FORM, CHORES>), dim3(blocks), dim3(512), 0, 0, d, dc, src, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<PAIR, FORM, CHORES>), dim3(blocks), dim3(512), 0, 0, d, dc, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512];
    hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int i = 0; i < blocks; ++i) { sa += (double)h[2 * i]; sb += (double)h[2 * i + 1]; }
    const double per_simd = PAIR == 0 ? 72.0 : 144.0;
    const double ns = ms * 1e6 / iters;
    // s_memtime ticks at the clock the shader engines hold (it moves with DVFS: 1.6 - 1.9 per ns under these loops), so the spans are shader cycles
    const double ca = sa / blocks / iters, cb = sb / blocks / iters, cmax = ca > cb ? ca : cb;
    printf("pair %d form %d chores %d  %-60s %7.1f ns/step %6.2f ns/MFMA | cycles/step A %6.0f B %6.0f = %5.2f cycles/MFMA at %.2f GHz\n", PAIR, FORM, CHORES, what, ns,
           ns / per_simd, ca, cb, cmax / per_simd, cmax / ns);
    fflush(stdout);
    if (getenv("PROBE")) {       // socket power and shader clock while this stream runs (60 more launches in flight behind a read-only rocm-smi poll)
        for (int r = 0; r < 60; ++r) hipLaunchKernelGGL((k<PAIR, FORM, CHORES>), dim3(blocks), dim3(512), 0, 0, d, dc, src, iters);
        if (system("sleep 1; rocm-smi -d 0 --showpower --showclocks 2>/dev/null | grep -E 'Socket|sclk' | sed -E 's/\\s+/ /g' | tr '\\n' ' '; echo")) {}
        hipDeviceSynchronize();
    }
}

int main() {
    float* d; unsigned long long* dc; float* src;
    hipMalloc(&d, 4096); hipMalloc(&dc, 512 * 8); hipMalloc(&src, 1 << 20);
    hipMemset(src, 0x3c, 1 << 20);
    for (int rep = 0; rep < (getenv("PROBE") ? 1 : 2); ++rep) {
    printf("-- pass %d\n", rep);
    run<0, 3, 0>("one wave per SIMD, register-only", d, dc, src);
    run<1, 3, 0>("two waves per SIMD, register-only, no barrier", d, dc, src);
    run<2, 3, 0>("two waves per SIMD, register-only, barrier per step", d, dc, src);
    run<0, 0, 0>("one wave per SIMD, LDS fragments 3 ahead", d, dc, src);
    run<1, 0, 0>("two waves, LDS fragments 3 ahead, no barrier", d, dc, src);
    run<2, 0, 0>("two waves, LDS fragments 3 ahead, barrier per step", d, dc, src);
    run<3, 0, 0>("two waves, 3 ahead, barrier, waves 4-7 half a step behind", d, dc, src);
    run<0, 1, 0>("one wave per SIMD, LDS fragments 6 ahead", d, dc, src);
    run<2, 1, 0>("two waves, 6 ahead, barrier per step", d, dc, src);
    run<3, 1, 0>("two waves, 6 ahead, barrier, half a step behind", d, dc, src);
    run<0, 2, 0>("one wave per SIMD, a row's 12 fragments up front", d, dc, src);
    run<2, 2, 0>("two waves, 12 up front, barrier per step", d, dc, src);
    run<3, 2, 0>("two waves, 12 up front, barrier, half a step behind", d, dc, src);
    run<0, 4, 0>("one wave, 3 ahead, fragment registers rest 2 groups", d, dc, src);
    run<2, 4, 0>("two waves, 3 ahead, registers rest 2 groups, barrier", d, dc, src);
    run<0, 5, 0>("one wave, 3 ahead, read between 2nd and 3rd MFMA", d, dc, src);
    run<2, 5, 0>("two waves, read between 2nd and 3rd MFMA, barrier", d, dc, src);
    run<0, 6, 0>("one wave, registers rest 2 groups + mid read", d, dc, src);
    run<2, 6, 0>("two waves, registers rest 2 groups + mid read, barrier", d, dc, src);
    run<0, 7, 0>("one wave, 3 ahead, fragments in AccVGPRs", d, dc, src);
    run<2, 7, 0>("two waves, fragments in AccVGPRs, barrier", d, dc, src);
    run<0, 8, 0>("one wave, both cout fragments: a fragment feeds 6 MFMAs", d, dc, src);
    run<2, 8, 0>("two waves, both cout fragments (6 reads per row), barrier", d, dc, src);
    run<2, 8, 1>("K67X shape with both cout fragments per wave", d, dc, src);
    run<2, 0, 2>("K67X shape, stage A's chores at the HEAD of its step", d, dc, src);
    run<2, 8, 2>("both cout fragments per wave, stage A's chores at the head", d, dc, src);
    run<2, 0, 1>("K67X shape: A tail chores, B head burst; barrier per step", d, dc, src);
    run<3, 0, 1>("K67X shape, waves 4-7 half a step behind", d, dc, src);
    run<2, 3, 1>("K67X shape, register-only fragments", d, dc, src);
    }
    return 0;
}
