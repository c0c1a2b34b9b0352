// Micro-benchmark: one wave per SIMD; v_mfma_f32_32x32x2_f32 alternating over NCH independent accumulators with NV
// independent VALU instructions after every MFMA.  Does VALU hide in the MFMA shadow when consecutive MFMAs are NOT a
// dependent chain (the dependent-chain case is slot_interleave.hip: +8 cycles per VALU)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH, int NV, bool TRANS>
__global__ void __launch_bounds__(256, 1) k(const float* __restrict__ g, float* out, int iters, long long* cyc) {
    __shared__ float lds[8192];
    int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = g[i];
    __syncthreads();
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x16{0};
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    float bop[4] = {g[lane], g[lane + 64], g[lane + 128], g[lane + 192]};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm("" : "+v"(lane));
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + (s * 64 + lane) * 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[(s * 4 + m) % NCH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bop[m], acc[(s * 4 + m) % NCH], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    if (TRANS && (j & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(bop[0]), "v"(bop[1]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int c = 0; c < NCH; ++c)
        for (int i = 0; i < 16; ++i) r += acc[c][i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH, int NV, bool TRANS>
void run(const float* g, float* out, long long* cyc) {
    const int iters = 200, blocks = 256;
    k<NCH, NV, TRANS><<<blocks, 256>>>(g, out, 2, cyc);
    (void)hipDeviceSynchronize();
    k<NCH, NV, TRANS><<<blocks, 256>>>(g, out, iters, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    printf("chains %d  VALU/MFMA %2d %s : %.1f ticks per MFMA\n", NCH, NV, TRANS ? "(1 in 4 v_exp)" : "              ", m / (iters * 64));
}

int main() {
    float *g, *out; long long* cyc;
    (void)hipMalloc(&g, 1 << 20); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(g, 0, 1 << 20);
    run<1, 0, false>(g, out, cyc); run<1, 2, false>(g, out, cyc); run<1, 4, false>(g, out, cyc); run<1, 8, false>(g, out, cyc);
    run<2, 0, false>(g, out, cyc); run<2, 2, false>(g, out, cyc); run<2, 4, false>(g, out, cyc); run<2, 8, false>(g, out, cyc);
    run<2, 12, false>(g, out, cyc); run<2, 16, false>(g, out, cyc);
    run<4, 4, false>(g, out, cyc); run<4, 8, false>(g, out, cyc);
    run<2, 4, true>(g, out, cyc); run<2, 8, true>(g, out, cyc); run<1, 4, true>(g, out, cyc);
    return 0;
}
