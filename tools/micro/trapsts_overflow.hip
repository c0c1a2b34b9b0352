// Does the wave's sticky exception state (TRAPSTS.EXCP, read with s_getreg_b32) record an f16 overflow in the conversions the
// split-precision form uses (v_cvt_pkrtz_f16_f32, v_fma_mixlo_f16)?  It would be a range guard with no cost in the sample loop.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int mode) {
    const float x = in[threadIdx.x];
    unsigned before, after, r = 0;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(before));
    if (mode == 0) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
    else if (mode == 1) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(r) : "v"(x));
    else if (mode == 2) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(x));
    else if (mode == 3) asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
    else if (mode == 4) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r) : "v"(x * 1e30f));
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after) : "v"(r));
    unsigned mode_reg;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode_reg));
    if (threadIdx.x == 0) { out[0] = before; out[1] = after; out[2] = r; out[3] = mode_reg; }
}
int main() {
    float* in; unsigned* out;
    (void)hipMalloc(&in, 256); (void)hipMalloc(&out, 64);
    const char* names[] = {"v_cvt_pkrtz_f16_f32", "v_fma_mixlo_f16", "v_cvt_f16_f32", "v_exp_f32", "v_mul_f32 (f32 overflow)"};
    for (int big = 0; big < 2; ++big)
        for (int mode = 0; mode < 5; ++mode) {
            float h[64];
            for (int i = 0; i < 64; ++i) h[i] = big ? 1.0e6f : 1.5f;
            if (mode == 3) for (int i = 0; i < 64; ++i) h[i] = big ? 200.f : 1.5f;
            (void)hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
            k<<<1, 64>>>(in, out, mode);
            unsigned o[4];
            (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
            printf("%-28s input %-8s TRAPSTS.EXCP before %03x after %03x  result %08x  MODE %08x\n", names[mode], big ? "huge" : "small", o[0], o[1], o[2], o[3]);
        }
    return 0;
}
