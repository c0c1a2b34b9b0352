// Timeline of two [RUN MFMAs -> BURST VALU] waves sharing a SIMD: s_memtime at run start / run end / burst end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RUN = 32, BURST = 128, ITERS = 24;

__global__ void __launch_bounds__(512, 2) k(const float* g, float* out, int sleeps, long long* tr) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a = g[lane], b = g[lane + 64];
    float v[8] = {a, b, a, b, a, b, a, b};
    f32x16 acc = {0};
    long long t[ITERS][3];
    if (wave >= 4) for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(16);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        t[it][0] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < RUN; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        asm volatile("" : "+v"(acc));
        t[it][1] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < BURST; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(a), "v"(b));
        asm volatile("" : "+v"(acc));
        t[it][2] = __builtin_amdgcn_s_memtime();
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0 && blockIdx.x == 0)
        for (int it = 0; it < ITERS; ++it)
            for (int j = 0; j < 3; ++j) tr[(wave * ITERS + it) * 3 + j] = t[it][j];
}

int main() {
    float *g, *out; long long* tr;
    (void)hipMalloc(&g, 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&tr, 8 * ITERS * 3 * 8);
    (void)hipMemset(g, 0, 4096);
    for (int sleeps = 0; sleeps <= 1; ++sleeps) {
        k<<<256, 512>>>(g, out, sleeps, tr); (void)hipDeviceSynchronize();
        k<<<256, 512>>>(g, out, sleeps, tr); (void)hipDeviceSynchronize();
        std::vector<long long> h(8 * ITERS * 3);
        (void)hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
        long long t0 = h[0];
        printf("offset sleeps %d   (wave 0 and wave 4 share SIMD 0?)  columns: run start, run end, burst end; run len, burst len\n", sleeps);
        for (int it = 8; it < 16; ++it)
            for (int w : {0, 4}) {
                long long* p = &h[(w * ITERS + it) * 3];
                printf("  wave %d it %2d: %7lld %7lld %7lld   run %5lld  burst %5lld\n", w, it, p[0] - t0, p[1] - t0, p[2] - t0, p[1] - p[0], p[2] - p[1]);
            }
    }
    return 0;
}
