// Does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs (needed for an f32 = hi + lo split in f16)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0; b[j] = (_Float16)0; }
    a[0] = (_Float16)a_val; b[0] = (_Float16)b_val;      // k = 0 (lanes < 32) and k = 8 (lanes >= 32)
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
    float* d; (void)hipMalloc(&d, 64);
    const float tests[][2] = {{1.0f, 1.0f}, {3e-6f, 1.0f}, {1.0f, 3e-6f}, {2.4e-5f, 0.5f}, {6.2e-5f, 1.0f}, {1e-7f, 1.0f}};
    for (auto& t : tests) {
        k<<<1, 64>>>(t[0], t[1], d);
        float h[3]; (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g b=%g : f16(a)=%g f16(b)=%g  mfma sum over the 2 lane halves = %g (expect %g)\n", t[0], t[1], h[1], h[2], h[0], 2.0 * h[1] * h[2]);
    }
    return 0;
}
