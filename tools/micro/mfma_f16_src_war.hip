// Does a VALU write to an A- or B-operand register of v_mfma_f32_32x32x16_f16 (four VGPRs each on gfx950) issued right BEHIND the
// MFMA disturb it (a write-after-read hazard the compiler does not pad: LLVM's tables have one for SrcC only)?  Suspected while
// one build of the split-precision form with the deferred colour branch gave wrong colours (gpnerf_kernels.hip SPLIT_DEFERS): in
// that build the compiler itself had placed `v_mul_f32 v90, ...` directly behind `v_mfma_f32_32x32x16_f16 .., v[90:93], ..`.
// For each of the eight operand registers and 0 .. 3 independent instructions in between: D = A x B with the register overwritten
// behind the MFMA, against the undisturbed product.  Two back-to-back MFMAs variant too (the second's operands overwritten while
// the first still occupies the pipe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define COPY_OUT "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v2\n\tv_mov_b32 %3, v3\n\tv_mov_b32 %4, v4\n\tv_mov_b32 %5, v5\n\tv_mov_b32 %6, v6\n\tv_mov_b32 %7, v7\n\t" \
                 "v_mov_b32 %8, v8\n\tv_mov_b32 %9, v9\n\tv_mov_b32 %10, v10\n\tv_mov_b32 %11, v11\n\tv_mov_b32 %12, v12\n\tv_mov_b32 %13, v13\n\tv_mov_b32 %14, v14\n\tv_mov_b32 %15, v15"
#define LOAD_IN "v_mov_b32 v40, %16\n\tv_mov_b32 v41, %17\n\tv_mov_b32 v42, %18\n\tv_mov_b32 v43, %19\n\tv_mov_b32 v44, %20\n\tv_mov_b32 v45, %21\n\tv_mov_b32 v46, %22\n\tv_mov_b32 v47, %23\n\ts_nop 7\n\t"
#define OUTS(r) "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), \
                "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
#define INS "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(junk)
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", \
             "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", \
             "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48"

// one MFMA, then GAP independent v_mov's into a scratch register, then the overwrite of register REG
#define CASE(REG, GAPTXT, slot)                                                                                          \
    {                                                                                                                    \
        float r[16];                                                                                                     \
        asm volatile(LOAD_IN "v_mfma_f32_32x32x16_f16 v[0:15], v[40:43], v[44:47], 0\n\t" GAPTXT                         \
                     "v_mov_b32 " REG ", %24\n\ts_nop 15\n\ts_nop 15\n\t" COPY_OUT                                       \
                     : OUTS(r) : INS : CLOB);                                                                            \
        for (int i = 0; i < 16; ++i) out[((slot) * 16 + i) * 64 + lane] = r[i];                                          \
    }
// two MFMAs back to back (independent accumulators); the SECOND one's operand register overwritten right behind it
#define CASE2(REG, GAPTXT, slot)                                                                                         \
    {                                                                                                                    \
        float r[16];                                                                                                     \
        asm volatile(LOAD_IN "v_mfma_f32_32x32x16_f16 v[20:35], v[44:47], v[40:43], 0\n\t"                               \
                     "v_mfma_f32_32x32x16_f16 v[0:15], v[40:43], v[44:47], 0\n\t" GAPTXT                                 \
                     "v_mov_b32 " REG ", %24\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t" COPY_OUT                           \
                     : OUTS(r) : INS : CLOB);                                                                            \
        for (int i = 0; i < 16; ++i) out[((slot) * 16 + i) * 64 + lane] = r[i];                                          \
    }
#define G0 ""
#define G1 "v_mov_b32 v48, %24\n\t"
#define G2 "v_mov_b32 v48, %24\n\tv_mov_b32 v48, %24\n\t"
#define G3 "v_mov_b32 v48, %24\n\tv_mov_b32 v48, %24\n\tv_mov_b32 v48, %24\n\t"

__global__ void k(float* out) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.25f * ((lane * 7 + j * 3) % 13) - 1.f); b[j] = (_Float16)(0.5f * ((lane * 5 + j) % 11) - 2.f); }
    const u32x4 A = __builtin_bit_cast(u32x4, a), B = __builtin_bit_cast(u32x4, b);
    const unsigned junk = 0x7bff7bffu;                      // two halfs of 65504
    f32x16 good = {0};
    good = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, good, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[i * 64 + lane] = good[i];
    // slots 1 .. 32: single MFMA; register index (0..7 = v40..v47) x gap (0..3)
    CASE("v40", G0, 1)  CASE("v41", G0, 2)  CASE("v42", G0, 3)  CASE("v43", G0, 4)  CASE("v44", G0, 5)  CASE("v45", G0, 6)  CASE("v46", G0, 7)  CASE("v47", G0, 8)
    CASE("v40", G1, 9)  CASE("v41", G1, 10) CASE("v42", G1, 11) CASE("v43", G1, 12) CASE("v44", G1, 13) CASE("v45", G1, 14) CASE("v46", G1, 15) CASE("v47", G1, 16)
    CASE("v40", G2, 17) CASE("v41", G2, 18) CASE("v42", G2, 19) CASE("v43", G2, 20) CASE("v44", G2, 21) CASE("v45", G2, 22) CASE("v46", G2, 23) CASE("v47", G2, 24)
    CASE("v40", G3, 25) CASE("v41", G3, 26) CASE("v42", G3, 27) CASE("v43", G3, 28) CASE("v44", G3, 29) CASE("v45", G3, 30) CASE("v46", G3, 31) CASE("v47", G3, 32)
    // slots 33 .. 48: behind a second, back-to-back MFMA; gap 0 and 1
    CASE2("v40", G0, 33) CASE2("v41", G0, 34) CASE2("v42", G0, 35) CASE2("v43", G0, 36) CASE2("v44", G0, 37) CASE2("v45", G0, 38) CASE2("v46", G0, 39) CASE2("v47", G0, 40)
    CASE2("v40", G1, 41) CASE2("v41", G1, 42) CASE2("v42", G1, 43) CASE2("v43", G1, 44) CASE2("v44", G1, 45) CASE2("v45", G1, 46) CASE2("v46", G1, 47) CASE2("v47", G1, 48)
}
int main() {
    const int slots = 49;
    float* d; static float h[49 * 16 * 64];
    (void)hipMalloc(&d, sizeof(h));
    (void)hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    int total = 0;
    for (int s = 1; s < slots; ++s) {
        int bad = 0;
        for (int i = 0; i < 16 * 64; ++i) bad += memcmp(&h[s * 16 * 64 + i], &h[i], 4) != 0;
        total += bad;
        const int idx = (s - 1) % 8, gap = s <= 32 ? (s - 1) / 8 : (s - 33) / 8;
        printf("%s %c[%d] overwritten %d instruction(s) behind the MFMA: %4d of 1024 results differ\n", s <= 32 ? "single      " : "back-to-back",
               idx < 4 ? 'A' : 'B', idx & 3, gap, bad);
    }
    printf("total differing: %d\n", total);
    return 0;
}
