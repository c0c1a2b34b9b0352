// Does v_mfma_f32_32x32x2_f32 share an execution resource with the f32 VALU?  One MFMA-only wave and one
// VALU-only wave per SIMD (512-thread block, 1 block/CU): time the MFMA wave alone and beside the VALU wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
// MODE 0: MFMA waves only, 1: + v_fma_f32 partner, 2: + v_exp_f32 partner, 3: + v_pk_fma_f32 partner,
//      4: v_fma_f32 waves only (no MFMA), 5: v_pk_fma_f32 waves only, 6: two MFMA waves per SIMD
template <int MODE>
__global__ void __launch_bounds__(512, 2) k(const float* g, float* out, int iters, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a = g[lane], b = g[lane + 64];
    float r = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    if ((wave < 4 && MODE < 4) || MODE == 6) {   // MFMA waves
        f32x16 acc = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += acc[i];
    } else if ((wave >= 4 || MODE == 7 || MODE == 8) && MODE > 0 && MODE != 6) {
        float v[8] = {a, b, a, b, a, b, a, b};
        f32x2 p[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
        const f32x2 pa = {a, b}, pb = {b, a};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 1024; ++i) {
                if (MODE == 1 || MODE == 4 || MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(a), "v"(b));
                else if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 3]) : "v"(pa), "v"(pb));
            }
        }
        for (int i = 0; i < 8; ++i) r += v[i];
        for (int i = 0; i < 4; ++i) r += p[i][0] + p[i][1];
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const float* g, float* out, long long* cyc, const char* name) {
    const int iters = 200, blocks = 256;
    k<MODE><<<blocks, 512>>>(g, out, iters, cyc);
    (void)hipDeviceSynchronize();
    k<MODE><<<blocks, 512>>>(g, out, iters, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < blocks; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) v += h[b * 8 + w]; }
    m /= blocks * 4; v /= blocks * 4;
    printf("%-34s MFMA wave: %.1f cycles per MFMA (64 = alone at full rate); VALU wave: %.2f cycles per VALU\n", name,
           m / (iters * 64.0), v / (iters * 1024.0));
}

int main() {
    float *g, *out; long long* cyc;
    (void)hipMalloc(&g, 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 64);
    (void)hipMemset(g, 0, 4096);
    run<0>(g, out, cyc, "f32 MFMA waves alone");
    run<1>(g, out, cyc, "f32 MFMA + v_fma_f32 partner");
    run<2>(g, out, cyc, "f32 MFMA + v_exp_f32 partner");
    run<3>(g, out, cyc, "f32 MFMA + v_pk_fma_f32 partner");
    run<4>(g, out, cyc, "v_fma_f32 waves alone");
    run<5>(g, out, cyc, "v_pk_fma_f32 waves alone");
    run<6>(g, out, cyc, "two f32 MFMA waves per SIMD");
    run<7>(g, out, cyc, "two v_fma_f32 waves per SIMD");
    run<8>(g, out, cyc, "two v_pk_fma_f32 waves per SIMD");
    return 0;
}
