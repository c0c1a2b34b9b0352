// What v_permlane32_swap_b32 does on gfx950 (the reference-order form's channel re-interleave rests on it):
// (a, b) -> ([a.lo, b.lo], [a.hi, b.hi]) with lo = lanes 0..31, hi = lanes 32..63.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const unsigned a = 100 + lane, b = 200 + lane;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r0: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
    printf("r1: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[64 + 31], h[64 + 32], h[64 + 63]);
    const bool ok = h[0] == 100 && h[31] == 131 && h[32] == 200 && h[63] == 231 && h[64] == 132 && h[95] == 163 && h[96] == 232 && h[127] == 263;
    printf(ok ? "as assumed: r0 = [a.lo, b.lo], r1 = [a.hi, b.hi]\n" : "NOT as assumed\n");
    return ok ? 0 : 1;
}
