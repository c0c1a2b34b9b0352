// Is a VGPR written by an MFMA protected when its consumer is an `asm()` statement?  Three kernels compute out = 2 * (A x B)[0]:
// `visible` adds in C++ (LLVM's hazard recogniser sees a VALU reading the MFMA's destination and pads with s_nop), `opaque`
// adds in inline assembly (to LLVM an INLINEASM node is not a VALU instruction: no padding).  gfx940/gfx950 do not interlock
// this dependency in hardware.  tools/isa_mfma_hazards.py flags `opaque` statically; run on the GPU this prints how many of the
// opaque kernel's results differ from the visible kernel's (stale accumulator contents read before the MFMA has written).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_asm_hazard mfma_asm_hazard.hip && ./mfma_asm_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// MODE 0: visible consumer; 1: opaque (asm) consumer; 2: opaque consumer behind an explicit `s_nop 11` (the fix when an asm
// statement must follow an MFMA: 12 wait states, what LLVM pads a visible consumer of this 8-pass MFMA with)
template <int MODE>
__global__ void __launch_bounds__(512) k(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float* __restrict__ out, int reps) {
    const int lane = threadIdx.x;
    h8 va, vb;
    for (int j = 0; j < 8; ++j) { va[j] = a[(blockIdx.x * 64 + lane) * 8 + j]; vb[j] = b[(blockIdx.x * 64 + lane) * 8 + j]; }
    float sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = (float)(r + 1);                 // a known "stale" value
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
        float d;
        if (MODE == 1) asm volatile("v_add_f32 %0, %1, %1" : "=v"(d) : "v"(acc[0]));
        else if (MODE == 2) asm volatile("s_nop 11\n\tv_add_f32 %0, %1, %1" : "=v"(d) : "v"(acc[0]));
        else d = acc[0] + acc[0];
        sum += d;
    }
    out[blockIdx.x * 64 + lane] = sum;
}

int main() {
    const int blocks = 1024, n = blocks * 64;
    std::vector<_Float16> ha(n * 8), hb(n * 8);
    for (int i = 0; i < n * 8; ++i) { ha[i] = (_Float16)(((i * 2654435761u) >> 20 & 255) / 64.f - 2.f); hb[i] = (_Float16)(((i * 40503u) >> 12 & 255) / 64.f - 2.f); }
    _Float16 *a, *b; float *o0, *o1, *o2;
    (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMalloc(&o0, n * 4); (void)hipMalloc(&o1, n * 4); (void)hipMalloc(&o2, n * 4);
    (void)hipMemcpy(a, ha.data(), n * 16, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), n * 16, hipMemcpyHostToDevice);
    k<0><<<blocks, 64>>>(a, b, o0, 8);
    k<1><<<blocks, 64>>>(a, b, o1, 8);
    k<2><<<blocks, 64>>>(a, b, o2, 8);
    std::vector<float> r0(n), r1(n), r2(n);
    (void)hipMemcpy(r0.data(), o0, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(r1.data(), o1, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(r2.data(), o2, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0; double worst = 0;
    for (int i = 0; i < n; ++i) { if (r0[i] != r1[i]) { ++bad; double e = fabs((double)r0[i] - r1[i]); if (e > worst) worst = e; } bad2 += r0[i] != r2[i]; }
    printf("opaque consumer: %d of %d results differ from the compiler-visible consumer's (max |diff| %.4g); behind s_nop 11: %d differ\n", bad, n, worst, bad2);
    return 0;
}
