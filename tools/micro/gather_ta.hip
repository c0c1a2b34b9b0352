// Micro-benchmark: cost of one wave-wide global_load_dwordx4 as a function of how its 64 lanes' 16-byte pieces
// are arranged (table resident in L2): every lane its own 128-B line / quads of lanes sharing 64 contiguous
// bytes / 8 lanes sharing a 128-B line.  8 waves per CU, 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int GROUP>   // lanes per contiguous group (1, 2, 4, 8)
__global__ void __launch_bounds__(512) k(const float* __restrict__ tab, unsigned nlines, float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    // GROUP == 0: the render kernel's pattern -- lanes l and l+32 read the two 64-B halves of the same line
    unsigned h = (GROUP == 0 ? (blockIdx.x * 512 + (threadIdx.x & ~32u)) : (blockIdx.x * 512 + threadIdx.x) / (GROUP ? GROUP : 1)) * 2654435761u;
    f32x4 acc = {0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            h = h * 1664525u + 1013904223u;
            const unsigned line = (h >> 8) % nlines;                         // 128-B line
            const unsigned off = GROUP == 0 ? line * 32 + (lane >> 5) * 16 + (j & 3) * 4
                                            : line * 32 + (lane % (GROUP ? GROUP : 1)) * 4 + ((GROUP == 1) ? ((h >> 4) & 7) * 4 : 0);
            const f32x4 v = *reinterpret_cast<const f32x4*>(tab + off);
            acc += v;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int GROUP>
void run(const float* tab, unsigned nlines, float* out, long long* cyc, const char* name) {
    const int iters = 200, blocks = 256;
    k<GROUP><<<blocks, 512>>>(tab, nlines, out, 2, cyc);
    (void)hipDeviceSynchronize();
    k<GROUP><<<blocks, 512>>>(tab, nlines, out, iters, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    printf("%-44s %.1f cycles per wave-instruction per CU (8 waves share the CU)\n", name, m / (iters * 16.0 * 8));
}

int main() {
    float *tab, *out; long long* cyc;
    const unsigned nlines = 1u << 15;   // 4 MiB table
    (void)hipMalloc(&tab, (size_t)nlines * 128 + 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(tab, 0, (size_t)nlines * 128 + 4096);
    run<0>(tab, nlines, out, cyc, "lanes l, l+32 share a line (render kernel)");
    run<1>(tab, nlines, out, cyc, "each lane its own line (16 B of 128)");
    run<2>(tab, nlines, out, cyc, "2 lanes share 32 contiguous bytes");
    run<4>(tab, nlines, out, cyc, "4 lanes share 64 contiguous bytes");
    run<8>(tab, nlines, out, cyc, "8 lanes share a 128-B line");
    return 0;
}
