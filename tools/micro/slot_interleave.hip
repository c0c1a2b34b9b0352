// Micro-benchmark: can ONE wave per SIMD keep the fp32 MFMA pipe busy while it also issues VALU work and loads,
// when the two are alternated slot by slot in program order (4 MFMAs + NV VALU + NL loads, sched_barrier between slots)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, bool FINE>
__global__ void __launch_bounds__(256, 1) k(const float* __restrict__ g, float* out, int iters, long long* cyc) {
    __shared__ float lds[8192];
    int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = g[i];
    __syncthreads();
    f32x16 acc = {0};
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    float bop[4] = {g[lane], g[lane + 64], g[lane + 128], g[lane + 192]};
    f32x4 ld[2] = {{0,0,0,0},{0,0,0,0}};
    const float* gp = g + (blockIdx.x * 256 + threadIdx.x) * 4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm("" : "+v"(lane));
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + (s * 64 + lane) * 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bop[m], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(bop[0]), "v"(bop[1]));
                if (m < NL) ld[m & 1] = *reinterpret_cast<const f32x4*>(gp + (((it * 16 + s) * 4 + m) * 4096) % (1 << 22));
                if (FINE) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NL, bool FINE>
void run(const float* g, float* out, long long* cyc, const char* name) {
    const int iters = 200, blocks = 256;
    k<NV, NL, FINE><<<blocks, 256>>>(g, out, 2, cyc);
    (void)hipDeviceSynchronize();
    k<NV, NL, FINE><<<blocks, 256>>>(g, out, iters, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    printf("%-28s cycles per slot (4 MFMA 32x32x2 = 256 min): %.1f\n", name, m / (iters * 16));
}

int main() {
    float *g, *out; long long* cyc;
    (void)hipMalloc(&g, (1 << 22) * 4 + (1<<24)); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(g, 0, (1 << 22) * 4 + (1<<24));
    run<0, 0, true>(g, out, cyc, "MFMA only");
    run<16, 0, false>(g, out, cyc, "+16 VALU  (slot barrier)");
    run<16, 0, true>(g, out, cyc, "+16 VALU  (per-MFMA barrier)");
    run<32, 0, false>(g, out, cyc, "+32 VALU  (slot barrier)");
    run<32, 0, true>(g, out, cyc, "+32 VALU  (per-MFMA barrier)");
    run<48, 0, true>(g, out, cyc, "+48 VALU  (per-MFMA barrier)");
    run<64, 0, true>(g, out, cyc, "+64 VALU  (per-MFMA barrier)");
    run<32, 2, true>(g, out, cyc, "+32 VALU +2 loads (fine)");
    run<32, 4, true>(g, out, cyc, "+32 VALU +4 loads (fine)");
    return 0;
}
