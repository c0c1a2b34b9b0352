// What can run beside v_mfma_f32_32x32x16_f16 on one SIMD?  (split-precision form of the fused kernel)
//   A: waves 0-3 stream a dependent f16 MFMA chain, waves 4-7 (their SIMD partners) stream one kind of vector instruction
//   B: ONE wave per SIMD: every MFMA is followed by N instructions of one kind (in-wave interleave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// KIND 0 none, 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_exp_f32, 4 v_cvt_pkrtz_f16_f32, 5 v_fma_mixlo_f16, 6 v_med3_f32, 7 v_mov_b32,
//      8 ds_read_b128, 9 v_pk_mul_f32, 10 v_add_u32
template <int KIND>
__device__ __forceinline__ void one(float (&v)[8], f32x2 (&p)[4], float a, float b, int i, const float* lds) {
    const f32x2 pa = {a, b};
    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(a), "v"(b));
    else if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 3]) : "v"(pa));
    else if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
    else if (KIND == 4) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(a));
    else if (KIND == 5) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v[i & 7]) : "v"(a), "v"(b));
    else if (KIND == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(a), "v"(b));
    else if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i & 7]) : "v"(a));
    else if (KIND == 8) { typedef float f4 __attribute__((ext_vector_type(4))); f4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((int)(threadIdx.x & 63) * 16)); }
    else if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 3]) : "v"(pa));
    else if (KIND == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(a));
}

template <int KIND, int NPER, bool SOLO>       // SOLO: in-wave interleave, 4 waves per block
__global__ void __launch_bounds__(512) k(const float* g, float* out, int iters, long long* cyc) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = g[lane];
    __syncthreads();
    float a = g[lane], b = g[lane + 64];
    float v[8] = {a, b, a, b, a, b, a, b};
    f32x2 p[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)a; hb[i] = (_Float16)b; }
    float r = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (SOLO) {
        f32x16 acc = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
                asm volatile("" : "+v"(acc));
#pragma unroll
                for (int j = 0; j < NPER; ++j) one<KIND>(v, p, a, b, i * NPER + j, lds);
            }
        }
        for (int i = 0; i < 16; ++i) r += acc[i];
    } else if (wave < 4) {
        f32x16 acc = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += acc[i];
    } else if (KIND) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 512; ++i) one<KIND>(v, p, a, b, i, lds);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) r += v[i];
    for (int i = 0; i < 4; ++i) r += p[i][0] + p[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float *g, *out; static long long* cyc;
template <int KIND>
void pair(const char* name) {
    const int iters = 100, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) { k<KIND, 0, false><<<blocks, 512>>>(g, out, iters, cyc); (void)hipDeviceSynchronize(); }
    std::vector<long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < blocks; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) v += h[b * 8 + w]; }
    m /= blocks * 4; v /= blocks * 4;
    printf("pair  %-22s MFMA wave %.1f cyc/MFMA (total %.0f)   partner %.2f cyc/instr (total %.0f)\n", name, m / (iters * 64.0), m, v / (iters * 512.0), v);
}
template <int KIND, int NPER>
void solo(const char* name) {
    const int iters = 100, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) { k<KIND, NPER, true><<<blocks, 256>>>(g, out, iters, cyc); (void)hipDeviceSynchronize(); }
    std::vector<long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
    double m = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) m += h[b * 8 + w];
    m /= blocks * 4;
    printf("solo  %-22s x%d per MFMA: %.1f cyc per MFMA+fillers\n", name, NPER, m / (iters * 32.0));
}
#define SOLOS(K, name) solo<K, 2>(name); solo<K, 4>(name); solo<K, 6>(name); solo<K, 8>(name);
int main() {
    (void)hipMalloc(&g, 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 64);
    (void)hipMemset(g, 0, 4096);
    pair<0>("(none)"); pair<1>("v_fma_f32"); pair<2>("v_pk_fma_f32"); pair<3>("v_exp_f32"); pair<4>("v_cvt_pkrtz_f16_f32");
    pair<5>("v_fma_mixlo_f16"); pair<6>("v_med3_f32"); pair<7>("v_mov_b32"); pair<8>("ds_read_b128"); pair<9>("v_pk_mul_f32"); pair<10>("v_add_u32");
    solo<0, 0>("(none)");
    SOLOS(1, "v_fma_f32") SOLOS(2, "v_pk_fma_f32") SOLOS(3, "v_exp_f32") SOLOS(4, "v_cvt_pkrtz") SOLOS(5, "v_fma_mixlo_f16") SOLOS(6, "v_med3_f32")
    SOLOS(7, "v_mov_b32") SOLOS(8, "ds_read_b128") SOLOS(10, "v_add_u32")
    return 0;
}
