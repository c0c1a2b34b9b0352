// Cycles per v_mfma_f32_32x32x16_f16 as the 3x3 convolution issues it: one wave per SIMD, one accumulator chain (and, beside it,
// three chains; two waves per SIMD).  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(512) rate(float* out, int iters, long long* cyc) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k % CHAINS], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS>
void run(const char* name, int threads, int grid) {
    float* out; long long* cyc; hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<CHAINS><<<grid, threads>>>(out, iters, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); rate<CHAINS><<<grid, threads>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 27.0 * iters;
    printf("%-44s %7.1f ns per MFMA per wave (wall), s_memtime ticks per MFMA %.2f (100 MHz ticks -> x24 = cycles at 2.4 GHz: %.1f)\n", name, ms * 1e6 / n, c / n, c / n * 24);
}

int main() {
    run<1>("1 wave/SIMD, 1 chain, 192 workgroups", 256, 192);
    run<3>("1 wave/SIMD, 3 chains, 192 workgroups", 256, 192);
    run<1>("2 waves/SIMD, 1 chain each, 192 workgroups", 512, 192);
    run<1>("1 wave/SIMD, 1 chain, 8 workgroups", 256, 8);
    return 0;
}
