// Are the inline-asm PRODUCERS of this code base safe in front of their consumers?  LLVM's hazard recogniser pads, for instructions it
// can see: (a) a VALU write of an MFMA's A / B operand register (no wait state needed on gfx950 per its tables, but unverifiable for
// an asm() producer), (b) a partial write -- v_fma_mixhi_f16 writes only the HIGH half of its destination -- followed at once by
// a reader (the dst-sel forwarding hazard of gfx940+: one wait state), (c) v_permlane32_swap followed at once by a reader.  An asm()
// statement is opaque to it, and the kernels emit exactly these three as inline assembly (lo_pair, interleave16).  Each kernel
// runs producer -> consumer inside ONE asm block (no compiler scheduling in between) with 0 or 4 wait states between them and the
// host compares: a difference = the hardware does not interlock that pair.
//   hipcc -O3 --offload-arch=gfx950 -o asm_producer_hazards asm_producer_hazards.hip && ./asm_producer_hazards
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define NOPS0 ""
#define NOPS1 "s_nop 0\n\t"
#define NOPS2 "s_nop 1\n\t"
#define NOPS3 "s_nop 2\n\t"
#define NOPS4 "s_nop 3\n\t"

// CASE 0: mixlo + mixhi -> v_mov reads the pair          CASE 1: mixlo + mixhi -> MFMA reads the register as part of its B operand
// CASE 2: permlane32_swap -> v_mov reads both results     CASE 3: permlane32_swap -> fp32 MFMA reads a result as its B operand
template <int CASE, int WAIT>
__global__ void __launch_bounds__(64) k(const float* __restrict__ x, unsigned* __restrict__ out, int reps) {
    const int lane = threadIdx.x, i = blockIdx.x * 64 + lane;
    float x0 = x[2 * i], x1 = x[2 * i + 1];
    unsigned acc_out = 0u;
    for (int r = 0; r < reps; ++r) {
        x0 += 0.125f; x1 -= 0.25f;
        const unsigned w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
        if constexpr (CASE == 0) {
            unsigned lo, got;
            if constexpr (WAIT < 4)
                asm volatile("v_mov_b32 %0, 0x7e007e00\n\ts_nop 4\n\tv_fma_mixlo_f16 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" NOPS0 "v_mov_b32 %1, %0"
                             : "=&v"(lo), "=&v"(got) : "v"(w), "v"(x0), "v"(x1));
            else
                asm volatile("v_mov_b32 %0, 0x7e007e00\n\ts_nop 4\n\tv_fma_mixlo_f16 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" NOPS4 "v_mov_b32 %1, %0"
                             : "=&v"(lo), "=&v"(got) : "v"(w), "v"(x0), "v"(x1));
            acc_out ^= got * 2654435761u + (unsigned)r;
        } else if constexpr (CASE == 1) {
            // B operand = v[100:103] = {w, w, w, lo}: lo (v103) is written by the asm pair directly in front of the MFMA that reads it
            const u32x4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};      // f16 ones
            f32x16 acc;
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#define CASE1(NOPS) \
            asm volatile("v_mov_b32 v100, %2\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, 0x7e007e00\n\ts_nop 4\n\t" \
                         "v_fma_mixlo_f16 v103, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" \
                         "v_fma_mixhi_f16 v103, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" NOPS \
                         "v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n\ts_nop 15" \
                         : "+v"(acc) : "v"(a), "v"(w), "v"(x0), "v"(x1) : "v100", "v101", "v102", "v103")
            if constexpr (WAIT == 0) CASE1(NOPS0); else if constexpr (WAIT == 1) CASE1(NOPS1); else if constexpr (WAIT == 2) CASE1(NOPS2);
            else if constexpr (WAIT == 3) CASE1(NOPS3); else CASE1(NOPS4);
            acc_out ^= __builtin_bit_cast(unsigned, acc[0]) + __builtin_bit_cast(unsigned, acc[5]) * 31u;
        } else if constexpr (CASE == 2) {
            float a = x0, b = x1;
            unsigned g0, g1;
            if constexpr (WAIT < 4)
                asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\t" NOPS0 "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1" : "+v"(a), "+v"(b), "=&v"(g0), "=&v"(g1));
            else
                asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\t" NOPS4 "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1" : "+v"(a), "+v"(b), "=&v"(g0), "=&v"(g1));
            acc_out ^= g0 * 2654435761u + g1 * 40503u + (unsigned)r;
        } else if constexpr (CASE == 3) {
            float a = x0, b = x1;
            f32x16 acc;
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            const float one = 1.f;
#define CASE3(NOPS) \
            asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %1, %2\n\t" NOPS "v_mfma_f32_32x32x2_f32 %0, %3, %1, %0\n\ts_nop 15\n\ts_nop 7" \
                         : "+v"(acc), "+v"(a), "+v"(b) : "v"(one))
            if constexpr (WAIT == 0) CASE3(NOPS0); else if constexpr (WAIT == 1) CASE3(NOPS1); else if constexpr (WAIT == 2) CASE3(NOPS2);
            else if constexpr (WAIT == 3) CASE3(NOPS3); else CASE3(NOPS4);
            acc_out ^= __builtin_bit_cast(unsigned, acc[0]) + __builtin_bit_cast(unsigned, acc[3]) * 31u;
        } else if constexpr (CASE == 4) {
            // a plain full-register VALU write (v_add_f32) of the fp32 MFMA's B operand
            float a = 0.f;
            f32x16 acc;
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            const float one = 1.f;
#define CASE4(NOPS) \
            asm volatile("v_mov_b32 %1, 0x7fc00000\n\ts_nop 4\n\tv_add_f32 %1, %3, %4\n\t" NOPS "v_mfma_f32_32x32x2_f32 %0, %2, %1, %0\n\ts_nop 15\n\ts_nop 7" \
                         : "+v"(acc), "=&v"(a) : "v"(one), "v"(x0), "v"(x1))
            if constexpr (WAIT == 0) CASE4(NOPS0); else if constexpr (WAIT == 1) CASE4(NOPS1); else if constexpr (WAIT == 2) CASE4(NOPS2);
            else if constexpr (WAIT == 3) CASE4(NOPS3); else CASE4(NOPS4);
            acc_out ^= __builtin_bit_cast(unsigned, acc[0]) + __builtin_bit_cast(unsigned, acc[3]) * 31u;
        } else {
            // a plain full-register VALU write (v_mov_b32) of one register of the f16 MFMA's B operand
            const u32x4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
            f32x16 acc;
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#define CASE5(NOPS) \
            asm volatile("v_mov_b32 v100, %2\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, 0x7e007e00\n\ts_nop 4\n\t" \
                         "v_mov_b32 v103, %2\n\t" NOPS "v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n\ts_nop 15" \
                         : "+v"(acc) : "v"(a), "v"(w) : "v100", "v101", "v102", "v103")
            if constexpr (WAIT == 0) CASE5(NOPS0); else if constexpr (WAIT == 1) CASE5(NOPS1); else if constexpr (WAIT == 2) CASE5(NOPS2);
            else if constexpr (WAIT == 3) CASE5(NOPS3); else CASE5(NOPS4);
            acc_out ^= __builtin_bit_cast(unsigned, acc[0]) + __builtin_bit_cast(unsigned, acc[5]) * 31u;
        }
    }
    out[i] = acc_out;
}

template <int CASE, int WAIT>
int differs(const float* dx, int blocks, const std::vector<unsigned>& want) {
    const int n = blocks * 64;
    unsigned* o;
    (void)hipMalloc(&o, n * 4);
    k<CASE, WAIT><<<blocks, 64>>>(dx, o, 16);
    std::vector<unsigned> r(n);
    (void)hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(o);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += r[i] != want[i];
    return bad;
}
template <int CASE>
int run(const float* dx, int blocks, const char* what) {
    const int n = blocks * 64;
    unsigned* o;
    (void)hipMalloc(&o, n * 4);
    k<CASE, 4><<<blocks, 64>>>(dx, o, 16);
    std::vector<unsigned> want(n);
    (void)hipMemcpy(want.data(), o, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(o);
    const int b0 = differs<CASE, 0>(dx, blocks, want), b1 = differs<CASE, 1>(dx, blocks, want), b2 = differs<CASE, 2>(dx, blocks, want),
              b3 = differs<CASE, 3>(dx, blocks, want);
    printf("%-76s lanes (of %d) that differ from 4 wait states, with 0 / 1 / 2 / 3: %d / %d / %d / %d%s\n", what, n, b0, b1, b2, b3,
           (b0 | b1 | b2 | b3) ? "   <-- NOT interlocked" : "");
    return b0 + b1 + b2 + b3;
}

int main() {
    const int blocks = 2048, n = blocks * 64;
    std::vector<float> hx(2 * n);
    for (int i = 0; i < 2 * n; ++i) hx[i] = ((i * 2654435761u) >> 12 & 4095) / 512.f - 4.f;
    float* dx;
    (void)hipMalloc(&dx, 2 * n * 4);
    (void)hipMemcpy(dx, hx.data(), 2 * n * 4, hipMemcpyHostToDevice);
    int bad = 0;
    bad += run<0>(dx, blocks, "v_fma_mixlo_f16 + v_fma_mixhi_f16 -> v_mov_b32 of the pair:");
    bad += run<1>(dx, blocks, "v_fma_mixlo_f16 + v_fma_mixhi_f16 -> v_mfma_f32_32x32x16_f16 B operand:");
    bad += run<2>(dx, blocks, "v_permlane32_swap_b32 -> v_mov_b32 of both results:");
    bad += run<3>(dx, blocks, "v_permlane32_swap_b32 -> v_mfma_f32_32x32x2_f32 B operand:");
    bad += run<4>(dx, blocks, "v_add_f32 (plain VALU write) -> v_mfma_f32_32x32x2_f32 B operand:");
    bad += run<5>(dx, blocks, "v_mov_b32 (plain VALU write) -> v_mfma_f32_32x32x16_f16 B operand:");
    printf(bad ? "at least one producer -> consumer pair is not interlocked by the hardware\n" : "all pairs give the same bits with and without wait states\n");
    return 0;
}
