// Reproducer attempt for what broke the split form's first range guard: a (never taken) branch between an MFMA chain's
// consumers and the next chain's operand conversions.  Both kernels compute the same thing; B has `if (__any(max >= limit))`
// between the layers, A does not.  Prints whether A == B and whether each is the same in 20 runs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_rtz(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ unsigned lo_pair(unsigned w, float x0, float x1) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w), "v"(x1));
    return r;
}
struct Frag { h8 hi, lo; };
__device__ __forceinline__ Frag make_frag(const float* v) {
    u32x4 H, L;
    for (int p = 0; p < 4; ++p) { H[p] = pk_rtz(v[2 * p], v[2 * p + 1]); L[p] = lo_pair(H[p], v[2 * p], v[2 * p + 1]); }
    Frag f; f.hi = __builtin_bit_cast(h8, H); f.lo = __builtin_bit_cast(h8, L);
    return f;
}
template <bool BRANCH>
__global__ void __launch_bounds__(512) k(const float* in, const unsigned* w, float* out, int layers, float limit, unsigned* flag) {
    __shared__ unsigned lw[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lw[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float t[16];
    for (int r = 0; r < 16; ++r) t[r] = in[(blockIdx.x * 512 + threadIdx.x) * 16 + r];
    for (int l = 0; l < layers; ++l) {
        if (BRANCH) {
            float m = fmaxf(fabsf(t[0]), fabsf(t[1]));
            for (int i = 2; i < 16; i += 2) m = fmaxf(fmaxf(m, fabsf(t[i])), fabsf(t[i + 1]));
            if (__any(!(m < limit))) flag[0] = 1u;
        }
        Frag f0 = make_frag(t), f1 = make_frag(t + 8);
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.01f * r;
        for (int s = 0; s < 2; ++s) {
            const u32x4 ah = *reinterpret_cast<const u32x4*>(lw + ((l & 1) * 2 + s) * 512 + lane * 4);
            const u32x4 al = *reinterpret_cast<const u32x4*>(lw + ((l & 1) * 2 + s) * 512 + 256 + lane * 4);
            const Frag& b = s ? f1 : f0;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, al), b.hi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ah), b.lo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ah), b.hi, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) t[r] = __builtin_amdgcn_fmed3f(acc[r], __builtin_amdgcn_exp2f(acc[r]) - 1.f, 0.f);
    }
    for (int r = 0; r < 16; ++r) out[(blockIdx.x * 512 + threadIdx.x) * 16 + r] = t[r];
}
int main() {
    const int blocks = 256, n = blocks * 512 * 16;
    std::vector<float> hin(n); std::vector<unsigned> hw(4096);
    for (int i = 0; i < n; ++i) hin[i] = ((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
    for (int i = 0; i < 4096; ++i) { _Float16 a = (_Float16)((((i * 40503u) >> 4) & 1023) / 8192.f - 0.0625f), b = (_Float16)((((i * 30011u) >> 3) & 1023) / 8192.f - 0.0625f);
        unsigned short ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2); hw[i] = ua | (ub << 16); }
    float *in, *oa, *ob; unsigned *w, *flag;
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&oa, n * 4); (void)hipMalloc(&ob, n * 4); (void)hipMalloc(&w, 4096 * 4); (void)hipMalloc(&flag, 4);
    (void)hipMemcpy(in, hin.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(w, hw.data(), 4096 * 4, hipMemcpyHostToDevice);
    std::vector<float> ra(n), rb(n), r0a, r0b;
    int diff_ab = 0, unstable_a = 0, unstable_b = 0;
    for (int run = 0; run < 20; ++run) {
        k<false><<<blocks, 512>>>(in, w, oa, 24, 65504.f, flag);
        k<true><<<blocks, 512>>>(in, w, ob, 24, 65504.f, flag);
        (void)hipMemcpy(ra.data(), oa, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(rb.data(), ob, n * 4, hipMemcpyDeviceToHost);
        if (run == 0) { r0a = ra; r0b = rb; }
        diff_ab += memcmp(ra.data(), rb.data(), n * 4) != 0;
        unstable_a += memcmp(ra.data(), r0a.data(), n * 4) != 0;
        unstable_b += memcmp(rb.data(), r0b.data(), n * 4) != 0;
    }
    printf("20 runs: branch-free != branchy in %d runs; branch-free changed from its first run in %d, branchy in %d\n", diff_ab, unstable_a, unstable_b);
    return 0;
}
