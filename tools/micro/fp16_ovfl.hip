// Does MODE.FP16_OVFL (bit 23) make v_cvt_pk_f16_f32 / v_cvt_f16_f32 saturate to +-65504 instead of +-inf on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fp16_ovfl.hip -o /tmp/fp16_ovfl && /tmp/fp16_ovfl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
__global__ void k(const float* in, unsigned* out, int on) {
    if (on) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // hwreg(HW_REG_MODE, 23, 1) = 1
    const f2 v = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
    out[threadIdx.x] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
    out[64 + threadIdx.x] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)in[2 * threadIdx.x]);
}
int main() {
    float h[8] = {1e6f, -1e6f, 65504.f, 65520.f, 70000.f, 1.0f, __builtin_inff(), -__builtin_inff()};
    float* d; unsigned* o; unsigned r[128];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int on = 0; on < 2; ++on) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, on);
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d: pk %08x %08x %08x %08x | scalar %04x %04x %04x %04x  (7c00 = inf, 7bff = 65504)\n", on, r[0], r[1], r[2], r[3], r[64], r[65], r[66], r[67]);
    }
    return 0;
}
