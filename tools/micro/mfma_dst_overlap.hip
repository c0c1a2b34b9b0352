// Does v_mfma_f32_32x32x2_f32 tolerate a destination that contains its B (or A) register?  LLVM emits
//   v_mfma_f32_32x32x2_f32 v[0:15], v16, v0, 0
// for chains that start from the constant 0 (the reference-order form's mfma_tile_bias): 40 places in its kernels.
// Measured on gfx950: YES -- "0 of 1024 results differ" for either operand (the sources are latched before the first write-back).
// (Suspected while the reference-order form gave garbage; the cause was elsewhere: interleave16's note in gpnerf_kernels.hip.)
// Each lane computes D = A x B with A = 1 in every lane, B = lane-dependent, once with separate registers and once with
// the destination placed over B / over A.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float a = 1.0f + 0.25f * (lane & 31), b = 2.0f + (lane >> 5) + 0.5f * (lane & 31);
    f32x16 good = {0};
    good = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, good, 0, 0, 0);
    float r_b[16], r_a[16];
    // destination v[0:15] over B (v0); A in v16
    asm volatile("v_mov_b32 v16, %16\n\tv_mov_b32 v0, %17\n\ts_nop 4\n\t"
                 "v_mfma_f32_32x32x2_f32 v[0:15], v16, v0, 0\n\ts_nop 15\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v2\n\tv_mov_b32 %3, v3\n\tv_mov_b32 %4, v4\n\tv_mov_b32 %5, v5\n\tv_mov_b32 %6, v6\n\tv_mov_b32 %7, v7\n\t"
                 "v_mov_b32 %8, v8\n\tv_mov_b32 %9, v9\n\tv_mov_b32 %10, v10\n\tv_mov_b32 %11, v11\n\tv_mov_b32 %12, v12\n\tv_mov_b32 %13, v13\n\tv_mov_b32 %14, v14\n\tv_mov_b32 %15, v15"
                 : "=&v"(r_b[0]), "=&v"(r_b[1]), "=&v"(r_b[2]), "=&v"(r_b[3]), "=&v"(r_b[4]), "=&v"(r_b[5]), "=&v"(r_b[6]), "=&v"(r_b[7]),
                   "=&v"(r_b[8]), "=&v"(r_b[9]), "=&v"(r_b[10]), "=&v"(r_b[11]), "=&v"(r_b[12]), "=&v"(r_b[13]), "=&v"(r_b[14]), "=&v"(r_b[15])
                 : "v"(a), "v"(b)
                 : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16");
    // destination v[0:15] over A (v0); B in v16
    asm volatile("v_mov_b32 v16, %17\n\tv_mov_b32 v0, %16\n\ts_nop 4\n\t"
                 "v_mfma_f32_32x32x2_f32 v[0:15], v0, v16, 0\n\ts_nop 15\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v2\n\tv_mov_b32 %3, v3\n\tv_mov_b32 %4, v4\n\tv_mov_b32 %5, v5\n\tv_mov_b32 %6, v6\n\tv_mov_b32 %7, v7\n\t"
                 "v_mov_b32 %8, v8\n\tv_mov_b32 %9, v9\n\tv_mov_b32 %10, v10\n\tv_mov_b32 %11, v11\n\tv_mov_b32 %12, v12\n\tv_mov_b32 %13, v13\n\tv_mov_b32 %14, v14\n\tv_mov_b32 %15, v15"
                 : "=&v"(r_a[0]), "=&v"(r_a[1]), "=&v"(r_a[2]), "=&v"(r_a[3]), "=&v"(r_a[4]), "=&v"(r_a[5]), "=&v"(r_a[6]), "=&v"(r_a[7]),
                   "=&v"(r_a[8]), "=&v"(r_a[9]), "=&v"(r_a[10]), "=&v"(r_a[11]), "=&v"(r_a[12]), "=&v"(r_a[13]), "=&v"(r_a[14]), "=&v"(r_a[15])
                 : "v"(a), "v"(b)
                 : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16");
    for (int r = 0; r < 16; ++r) {
        out[(0 * 16 + r) * 64 + lane] = good[r];
        out[(1 * 16 + r) * 64 + lane] = r_b[r];
        out[(2 * 16 + r) * 64 + lane] = r_a[r];
    }
}
int main() {
    float* d; static float h[3 * 16 * 64];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad_b = 0, bad_a = 0;
    for (int i = 0; i < 16 * 64; ++i) { bad_b += h[16 * 64 + i] != h[i]; bad_a += h[2 * 16 * 64 + i] != h[i]; }
    printf("destination over B: %d of 1024 results differ; destination over A: %d of 1024 differ\n", bad_b, bad_a);
    return 0;
}
