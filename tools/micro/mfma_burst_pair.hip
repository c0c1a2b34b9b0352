// Micro-benchmark: W waves per SIMD, each looping [RUN dependent v_mfma_f32_32x32x2_f32] -> [BURST VALU] (the shape of
// one MLP layer + its ELU).  How well do two such waves share the matrix pipe, and does a start offset change it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int RUN, int BURST, int PRIO>
__global__ void __launch_bounds__(512, 2) k(const float* g, float* out, int iters, int offset_sleeps, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a = g[lane], b = g[lane + 64];
    float v[8] = {a, b, a, b, a, b, a, b};
    f32x16 acc = {0};
    if (wave >= 4) for (int i = 0; i < offset_sleeps; ++i) __builtin_amdgcn_s_sleep(16);   // ~1k cycles each
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < RUN; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        asm volatile("" : "+v"(acc));
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < BURST; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(a), "v"(b));
        asm volatile("" : "+v"(acc));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RUN, int BURST, int PRIO = 0>
void run(const float* g, float* out, long long* cyc, int threads, int sleeps) {
    const int iters = 400, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) { k<RUN, BURST, PRIO><<<blocks, threads>>>(g, out, iters, sleeps, cyc); (void)hipDeviceSynchronize(); }
    std::vector<long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double m = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) m += h[b * 8 + w];
    m /= blocks * waves * (double)iters;
    const double ideal1 = RUN * 64.0 + BURST * 4.8;
    double first = 0, second = 0;
    for (int b = 0; b < blocks; ++b) { for (int w = 0; w < 4; ++w) first += h[b * 8 + w]; if (waves > 4) for (int w = 4; w < 8; ++w) second += h[b * 8 + w]; }
    printf("   waves 0-3: %.0f   waves 4-7: %.0f cycles per iteration\n", first / (blocks * 4.0 * iters), second / (blocks * 4.0 * iters));
    printf("prio %d run %3d burst %3d  waves/SIMD %d  offset %2d k-cycles: %8.0f cycles per wave-iteration (MFMA alone %d, 1-wave serial %.0f) pipe busy %.0f%%\n",
           PRIO, RUN, BURST, waves / 4, sleeps, m, RUN * 64, ideal1, 100.0 * (waves / 4) * RUN * 64 / m);
}

int main() {
    float *g, *out; long long* cyc;
    (void)hipMalloc(&g, 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 64);
    (void)hipMemset(g, 0, 4096);
    run<32, 128>(g, out, cyc, 256, 0);
    run<32, 128>(g, out, cyc, 512, 0);
    run<32, 128>(g, out, cyc, 512, 1);
    run<32, 128>(g, out, cyc, 512, 3);
    run<16, 64>(g, out, cyc, 512, 0);
    run<16, 64>(g, out, cyc, 512, 1);
    run<64, 128>(g, out, cyc, 512, 0);
    run<64, 128>(g, out, cyc, 512, 2);
    run<32, 128, 1>(g, out, cyc, 512, 0);
    run<32, 128, 2>(g, out, cyc, 512, 0);
    run<32, 128, 1>(g, out, cyc, 512, 1);
    run<32, 128, 2>(g, out, cyc, 512, 1);
    return 0;
}
