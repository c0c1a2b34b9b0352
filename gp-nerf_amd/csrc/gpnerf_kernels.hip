// gpnerf_kernels.hip -- gfx950 (MI355X) kernels and C ABI of GP-NeRF's per-ray render path.
//
// One wavefront renders a tile of 32 rays: ray = lane & 31, the two lane-halves split the K
// dimension of every dense layer (v_mfma_f32_32x32x2_f32 takes k = lane >> 5).  The wave walks the
// samples of its rays front to back; per step it gathers the features of 32 samples (one per ray),
// runs the whole density/colour MLP on the matrix cores with activations kept in registers
// (see head_layout.h), and folds the result into per-lane compositing state -- so the
// transmittance "scan" of the reference's cumprod is a plain per-lane running product and
// early ray termination is one __all() per step.
//
// Reference semantics (paths relative to the reference root) are cited at each step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#include "../../include/gpnerf_hip.h"
#include "head_layout.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define DEV __device__ __forceinline__

namespace {

constexpr int NV = GPNERF_VIEWS;
constexpr int RAYS_PER_WAVE = 32;
constexpr float LOG2E = 1.44269504088896340736f;

// The lab's hook points (per-phase cycle stamps, per-wavefront time stamps, launcher experiment knobs): EMPTY in the product --
// this resolves to csrc/nodiag/gpnerf_diag.h on the product's include path; csrc/diag/Makefile builds the diagnostic libraries
// with csrc/diag/gpnerf_diag.h instead (never loaded by the product).
#include "gpnerf_diag.h"

DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }       // v_exp_f32
// nn.ELU(alpha=1) = x > 0 ? x : exp(x) - 1.  exp(x) - 1 >= x everywhere and has x's sign, so the select is the median
// of (x, exp(x) - 1, 0): one v_med3_f32 instead of a compare and a conditional move.
// The dense layers run in a scaled domain: every MFMA accumulator holds x' = log2(e) * (W h + b) and every activation
// h' = log2(e) * ELU(x) = med3(x', (2^x' - 1) * log2(e), 0) -- v_exp_f32 on the accumulator as it is, one FMA, one
// median.  A layer fed by scaled activations needs W h' + log2(e) b, i.e. its weights unchanged and its bias scaled; the
// packers (pack_scale) scale the biases, the weight columns that multiply raw inputs, and the two VALU tails by ln 2.
DEV float elus(float xs) { return __builtin_amdgcn_fmed3f(xs, fmaf(__builtin_amdgcn_exp2f(xs), LOG2E, -LOG2E), 0.f); }
// N accumulator registers at once, stage by stage: the v_exp_f32 results are not consumed back to back (a transcendental
// feeding the next instruction costs a wait state) and the FMAs pair up as v_pk_fma_f32
// The split form leaves the ordering pins off: its f16 MFMAs hold the vector issue port for 8 of their 32 cycles, so what
// the scheduler threads between them is (mostly) hidden -- measured 7.32 -> 7.17 ms on the headline frame; the fp32 form
// (PIN = true) loses time the same way, its MFMAs run on the vector ALUs.
constexpr bool SPIN = false;
template <int N, bool PIN = true>
DEV void elus_n(f32x16& a, float* out) {
    if constexpr (PIN) asm volatile("" : "+v"(a));          // not before the MFMA groups issued so far (see the note at the end)
    float e[N];
#pragma unroll
    for (int r = 0; r < N; ++r) e[r] = __builtin_amdgcn_exp2f(a[r]);
#pragma unroll
    for (int r = 0; r < N; r += 2) {
        const f32x2 m = __builtin_elementwise_fma(f32x2{e[r], e[r + 1]}, f32x2{LOG2E, LOG2E}, f32x2{-LOG2E, -LOG2E});
        e[r] = m[0]; e[r + 1] = m[1];
    }
#pragma unroll
    for (int r = 0; r < N; ++r) out[r] = __builtin_amdgcn_fmed3f(a[r], e[r], 0.f);
    // All N results exist before anything that follows in program order: left alone, the scheduler threads these VALU
    // instructions between the next layer's MFMAs, and every MFMA -> VALU -> MFMA switch inside one wave idles the matrix
    // pipe for ~20 cycles (tools/micro/mfma_chains.hip).  One burst per layer, then an uninterrupted MFMA run.
    static_assert(N == 8 || N == 16, "");
    if constexpr (!PIN) return;
    asm volatile("" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]), "+v"(out[4]), "+v"(out[5]), "+v"(out[6]), "+v"(out[7]));
    if constexpr (N == 16)
        asm volatile("" : "+v"(out[8]), "+v"(out[9]), "+v"(out[10]), "+v"(out[11]), "+v"(out[12]), "+v"(out[13]), "+v"(out[14]), "+v"(out[15]));
}

// ---------------------------------------------------------------------------------------------
// dense layers on v_mfma_f32_32x32x2_f32
// ---------------------------------------------------------------------------------------------
// One 32-row output tile: acc += W_tile * B.  `w` points at the tile's LDS image
// [NT/4][64 lanes][4 k-steps] (+ [64][2] tail when NT % 4 == 2); b[t] is this lane's B value of k-step t.
template <int NT>
DEV void mfma_tile(const float* __restrict__ w, int lane, const float (&b)[NT], f32x16& acc) {
    constexpr int NG = NT / 4;
    // Software-pipelined weight reads: group g+1's ds_read_b128 is issued before group g's four MFMAs, so its LDS
    // latency hides behind 256 cycles of matrix work.  (Left to itself the compiler reuses one register quad: read,
    // wait ~100 cycles, 4 MFMAs, read, wait ...)  The empty asm orders the read ahead of the MFMAs -- it "modifies" the
    // offset the read used (so the read cannot sink below it) and the accumulator (so the MFMAs cannot rise above it) --
    // without consuming the loaded value, i.e. without a wait.
    typedef const __attribute__((address_space(3))) float* lds_ptr;      // 32-bit LDS address: one VGPR to launder
    lds_ptr p = (lds_ptr)w + lane * 4;
    f32x4 a = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(p);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        f32x4 an = a;
        f32x2 at = {0.f, 0.f};
        if (g + 1 < NG) an = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(p + (g + 1) * 256);
        else if constexpr (NT % 4 == 2) at = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((lds_ptr)w + NG * 256 + lane * 2);
        asm volatile("" : "+v"(p), "+v"(acc));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[4 * g + 3], acc, 0, 0, 0);
        a = an;
        if constexpr (NT % 4 == 2) {
            if (g + 1 == NG) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(at[0], b[4 * NG + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(at[1], b[4 * NG + 1], acc, 0, 0, 0);
            }
        }
    }
}

// The same, but the sum starts from another tile's value WITHOUT copying it first: the first MFMA reads `cin` as its C operand
// and writes a fresh accumulator (the per-view layer starts from the shared [mean,var] tile three times over).
template <int NT>
DEV f32x16 mfma_tile_from(const float* __restrict__ w, int lane, const float (&b)[NT], const f32x16& cin) {
    static_assert(NT % 4 == 2 || NT % 4 == 0, "");
    typedef const __attribute__((address_space(3))) float* lds_ptr;
    lds_ptr p = (lds_ptr)w + lane * 4;
    f32x4 a = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(p);
    constexpr int NG = NT / 4;
    f32x4 an = a;
    f32x2 at = {0.f, 0.f};
    if (1 < NG) an = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(p + 256);
    else if constexpr (NT % 4 == 2) at = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((lds_ptr)w + NG * 256 + lane * 2);
    asm volatile("" : "+v"(p));
    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], cin, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
    a = an;
#pragma unroll
    for (int g = 1; g < NG; ++g) {
        an = a;
        if (g + 1 < NG) an = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(p + (g + 1) * 256);
        else if constexpr (NT % 4 == 2) at = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((lds_ptr)w + NG * 256 + lane * 2);
        asm volatile("" : "+v"(p), "+v"(acc));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[4 * g + 3], acc, 0, 0, 0);
        a = an;
    }
    if constexpr (NT % 4 == 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(at[0], b[4 * NG + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(at[1], b[4 * NG + 1], acc, 0, 0, 0);
    }
    return acc;
}

// The reference-order chain: as mfma_tile_from, and the tile's 16 biases (this half's) are fetched from LDS while the chain's
// LAST group of MFMAs runs -- they are added after the chain (sgemm's beta = 1 step), and read at their first use they
// cost an exposed LDS round trip per tile.
template <int NT>
DEV f32x16 mfma_tile_bias(const float* __restrict__ w, int lane, const float (&b)[NT], const f32x16& cin,
                          const float* __restrict__ bias, f32x4 (&bq)[4]) {
    static_assert(NT % 4 == 2 || NT % 4 == 0, "");
    typedef const __attribute__((address_space(3))) float* lds_ptr;
    typedef const __attribute__((address_space(3))) f32x4* lds_ptr4;
    lds_ptr p = (lds_ptr)w + lane * 4;
    lds_ptr bp = (lds_ptr)bias;
    f32x4 a = *reinterpret_cast<lds_ptr4>(p);
    constexpr int NG = NT / 4;
    f32x4 an = a;
    f32x2 at = {0.f, 0.f};
    f32x16 acc = cin;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        an = a;
        if (g + 1 < NG) an = *reinterpret_cast<lds_ptr4>(p + (g + 1) * 256);
        else {
            if constexpr (NT % 4 == 2) at = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((lds_ptr)w + NG * 256 + lane * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<lds_ptr4>(bp + 4 * q);
            asm volatile("" : "+v"(bp));
        }
        if (g == 0) asm volatile("" : "+v"(p));
        else asm volatile("" : "+v"(p), "+v"(acc));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[4 * g + 3], acc, 0, 0, 0);
        a = an;
    }
    if constexpr (NT % 4 == 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(at[0], b[4 * NG + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(at[1], b[4 * NG + 1], acc, 0, 0, 0);
    }
    return acc;
}

// accumulator initialised with the bias of tile m of layer L (image: [half][16 regs])
template <int L>
DEV f32x16 bias_tile(const float* __restrict__ lds, int m, int half) {
    const float* p = lds + gpl::b_off(L) + m * 32 + half * 16;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
        acc[4 * q + 0] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
    }
    return acc;
}

template <int L>
DEV const float* wtile(const float* lds, int m) { return lds + gpl::w_off(L) + m * gpl::NT[L] * 64; }

// sigma feature: Linear(128,64)+ELU on the volume features (trainhead.py:39-40,58), 32 samples.
//   fv[64] : this half's 16 channels of each of the 4 volume levels (k-step t: level t>>4, channel 16h+(t&15))
//   sf[32] : the 64 output features as two accumulator tiles (B-operand order of the density layer)
DEV void geo_eval(const float* __restrict__ lds, int lane, const float (&fv)[64], float (&sf)[32]) {
    // The weights are loop-invariant across samples; without this the compiler hoists every LDS weight
    // read out of the sample loop and spills ~140 KB of them to scratch.  Launder the lane id so the
    // reads depend on a value the optimiser cannot see through.
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    f32x16 g0 = bias_tile<gpl::GEO>(lds, 0, half), g1 = bias_tile<gpl::GEO>(lds, 1, half);
    mfma_tile<64>(wtile<gpl::GEO>(lds, 0), lane, fv, g0);
    mfma_tile<64>(wtile<gpl::GEO>(lds, 1), lane, fv, g1);
    elus_n<16>(g0, sf);
    elus_n<16>(g1, sf + 16);
}

// The rest of NeRFHead.forward (libs/nerfheads/trainhead.py:118-145,159-163) for 32 samples.
//   x[v][18]: this half's slots of [rgb(3), feat(32)] of view v (gpl::idx35)
//   nvalid : number of valid views of this lane's sample
// returns sigma and rgb (identical in both halves of a ray).
// cross-view mean / variance (fused_mean_variance, trainhead.py:20-24): all 3 views, unmasked
DEV void mean_var(const float (&x)[NV][18], float (&mv)[36]) {
#pragma unroll
    for (int t = 0; t < 18; ++t) {
        const float m = ((x[0][t] + x[1][t]) + x[2][t]) * (1.f / 3.f);
        const float a = x[0][t] - m, b = x[1][t] - m, c = x[2][t] - m;
        mv[t] = m;
        mv[18 + t] = ((a * a + b * b) + c * c) * (1.f / 3.f);
    }
}

// density branch alone (the deferred-colour sample loop, render_tile)
DEV void mlp_density(const float* __restrict__ lds, int lane, const float (&sf)[32], const float (&mv)[36], float nvalid, float& sigma) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    float d1in[68];
#pragma unroll
    for (int r = 0; r < 32; ++r) d1in[r] = sf[r];
#pragma unroll
    for (int t = 0; t < 36; ++t) d1in[32 + t] = mv[t];

    // ---- density branch 134 -> 64 -> 32 -> 16 -> 1 (trainhead.py:102-110,133-137) ----
    {
        f32x16 a0 = bias_tile<gpl::D1>(lds, 0, half), a1 = bias_tile<gpl::D1>(lds, 1, half);
        mfma_tile<68>(wtile<gpl::D1>(lds, 0), lane, d1in, a0);
        mfma_tile<68>(wtile<gpl::D1>(lds, 1), lane, d1in, a1);
        float h1[32];
        elus_n<16>(a0, h1);
        elus_n<16>(a1, h1 + 16);
        f32x16 a2 = bias_tile<gpl::D2>(lds, 0, half);
        mfma_tile<32>(wtile<gpl::D2>(lds, 0), lane, h1, a2);
        float h2[16];
        elus_n<16>(a2, h2);
        f32x16 a3 = bias_tile<gpl::D3>(lds, 0, half);
        mfma_tile<16>(wtile<gpl::D3>(lds, 0), lane, h2, a3);
        // 16 -> 1 on the VALU: this half holds features ft(r,h), r < 8
        const float* w4 = lds + gpl::D4_W + half * 8;
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) part = fmaf(w4[r], elus(a3[r]), part);
        float s = part + __shfl_xor(part, 32) + lds[gpl::D4_B];
        s = fmaxf(s, 0.f);                              // nn.ReLU
        sigma = (nvalid < 1.f) ? 0.f : s;               // masked_fill(num_valid_obs < 1, 0)
    }
}

// colour branch alone
DEV void mlp_colour(const float* __restrict__ lds, int lane, const float (&x)[NV][18], const float (&mv)[36], float (&rgb)[3], Stamps& st) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    // ---- colour branch (trainhead.py:85-100,131,139-143) ----
    // base_fc layer 1 on [mean, var, x_v]: the [mean, var] part is view-independent, computed once
    f32x16 s0 = bias_tile<gpl::BS>(lds, 0, half), s1 = bias_tile<gpl::BS>(lds, 1, half);
    mfma_tile<36>(wtile<gpl::BS>(lds, 0), lane, mv, s0);
    mfma_tile<36>(wtile<gpl::BS>(lds, 1), lane, mv, s1);
    float y[48];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        // the three views share these weights; keep the compiler from holding ~100 VGPRs of them across views
        asm volatile("" : "+v"(lane));
        f32x16 a0 = mfma_tile_from<18>(wtile<gpl::BV>(lds, 0), lane, x[v], s0);
        f32x16 a1 = mfma_tile_from<18>(wtile<gpl::BV>(lds, 1), lane, x[v], s1);
        float h1[32];
        elus_n<16>(a0, h1);
        elus_n<16>(a1, h1 + 16);
        f32x16 a2 = bias_tile<gpl::B2>(lds, 0, half);
        mfma_tile<32>(wtile<gpl::B2>(lds, 0), lane, h1, a2);
        float xb[16];
        elus_n<16>(a2, xb);
        f32x16 t1 = bias_tile<gpl::V1>(lds, 0, half);
        mfma_tile<16>(wtile<gpl::V1>(lds, 0), lane, xb, t1);                               // weights carry the 1 / num_views
        float u1[16];
        elus_n<16>(t1, u1);
        f32x16 t2 = bias_tile<gpl::V2>(lds, 0, half);
        mfma_tile<16>(wtile<gpl::V2>(lds, 0), lane, u1, t2);
        elus_n<16>(t2, y + 16 * v);
#pragma unroll
        for (int r = 0; r < 16; ++r) y[16 * v + r] += xb[r];                                // x = x + x_vis
    }
    STAMP(st, 4);
    {
        f32x16 c1 = bias_tile<gpl::R1>(lds, 0, half);
        mfma_tile<48>(wtile<gpl::R1>(lds, 0), lane, y, c1);
        float h1[16];
        elus_n<16>(c1, h1);
        f32x16 c2 = bias_tile<gpl::R2>(lds, 0, half);
        mfma_tile<16>(wtile<gpl::R2>(lds, 0), lane, h1, c2);
        float e[8];
        elus_n<8>(c2, e);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float* w3 = lds + gpl::R3_W + o * 16 + half * 8;
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) part = fmaf(w3[r], e[r], part);
            const float s = part + __shfl_xor(part, 32) + lds[gpl::R3_B + o];
            rgb[o] = 1.f / (1.f + fast_exp(-s));        // .sigmoid()
        }
    }
    STAMP(st, 5);
}

DEV void mlp_eval(const float* __restrict__ lds, int lane, const float (&sf)[32], const float (&x)[NV][18],
                  float nvalid, float& sigma, float (&rgb)[3], Stamps& st) {
    float mv[36];
    mean_var(x, mv);
    mlp_density(lds, lane, sf, mv, nvalid, sigma);
    STAMP(st, 3);
    mlp_colour(lds, lane, x, mv, rgb, st);
}

// ---------------------------------------------------------------------------------------------
// reference-order MLP core (FORM_F32: the fp32 form without folded volumes; head_layout.h gpr, gpnerf_pack_head_ref)
// ---------------------------------------------------------------------------------------------
// On trained parameters (head weights x 1.5 ... x 3) every layer amplifies what the layers before it rounded differently from
// the reference, so a kernel that is "fp32 with another summation order" sits 5-10 x further from the reference's maps than
// an order-faithful one -- and EACH deviation alone (bias first, k in register order, the log2(e)-scaled domain, x * (1/3) for
// x / 3, FMA-accumulated trilinear taps, folded levels) costs most of that, because one early layer out of step decorrelates
// everything behind it (oracle/kernel_order.inc reproduces the round-4 kernel's distances to two digits on the CPU and
// switches the deviations one at a time; profiles/r05/a_kernel_order_attribution.txt).  This form therefore follows the
// reference's order everywhere it matters: each dense layer is sgemm's chain -- k ascending from ZERO, one FMA per k -- then
// the bias, then ELU on the unscaled value; mean / variance and vis_fc's input divide by 3 with IEEE rounding; the trilinear
// taps multiply, then add.  What is left: exp through v_exp_f32 (ELU's negative branch as 2^(x log2 e) - 1, sigmoid, alpha),
// and the two small output layers as two half sums -- measured harmless (same file).

// x / 3, correctly rounded (torch.mean divides): q = RN(x / 3) for every finite x whose quotient is normal -- c = RN(1/3), one
// Newton step on the residual, which fma(-3, q0, x) delivers exactly (checked against IEEE division on 2^32 samples by
// tests/test_kernel_order.py's C twin).  Three full-rate instructions against ~10 for v_rcp + refinement + v_div_fixup.
DEV f32x2 div3(f32x2 x) {
    const f32x2 c = {1.f / 3.f, 1.f / 3.f};
    const f32x2 q = x * c;
    const f32x2 r = __builtin_elementwise_fma(f32x2{-3.f, -3.f}, q, x);
    return __builtin_elementwise_fma(r, c, q);
}

// bias, then nn.ELU on N accumulator registers: x = s + b;  x > 0 ? x : 2^(x log2 e) - 1  as med3(x, e - 1, 0)
// (e - 1 is exact by Sterbenz for e in [1/2, 1]; where expm1 would return -|tiny| this returns 0: 6e-8 absolute at most).
// `b`: this half's 16 biases of the tile (mfma_tile_bias fetched them).  Staged like elus_n: all the adds, then the exps, then the medians.
template <int N>
DEV void elur_n(f32x16& a, const f32x4 (&b)[4], float* out) {
    asm volatile("" : "+v"(a));
    float x[N], e[N];
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const f32x4 bv = b[q];
        const f32x2 s0 = f32x2{a[4 * q], a[4 * q + 1]} + f32x2{bv[0], bv[1]};
        const f32x2 s1 = f32x2{a[4 * q + 2], a[4 * q + 3]} + f32x2{bv[2], bv[3]};
        x[4 * q] = s0[0]; x[4 * q + 1] = s0[1]; x[4 * q + 2] = s1[0]; x[4 * q + 3] = s1[1];
    }
#pragma unroll
    for (int r = 0; r < N; r += 2) {
        const f32x2 t = f32x2{x[r], x[r + 1]} * f32x2{LOG2E, LOG2E};
        e[r] = t[0]; e[r + 1] = t[1];
    }
#pragma unroll
    for (int r = 0; r < N; ++r) e[r] = __builtin_amdgcn_exp2f(e[r]);
#pragma unroll
    for (int r = 0; r < N; r += 2) {
        const f32x2 m = f32x2{e[r], e[r + 1]} - f32x2{1.f, 1.f};
        e[r] = m[0]; e[r + 1] = m[1];
    }
#pragma unroll
    for (int r = 0; r < N; ++r) out[r] = __builtin_amdgcn_fmed3f(x[r], e[r], 0.f);
    static_assert(N == 8 || N == 16, "");
    asm volatile("" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]), "+v"(out[4]), "+v"(out[5]), "+v"(out[6]), "+v"(out[7]));
    if constexpr (N == 16)
        asm volatile("" : "+v"(out[8]), "+v"(out[9]), "+v"(out[10]), "+v"(out[11]), "+v"(out[12]), "+v"(out[13]), "+v"(out[14]), "+v"(out[15]));
}

template <int L>
DEV const float* bias_ptr(const float* __restrict__ lds, int m, int half) { return lds + gpl::b_off(L) + m * 32 + half * 16; }

DEV f32x16 zero_tile() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// 16 gathered channels per half (register c: channel c in half 0, 16 + c in half 1) -> k order: out[i] = channels (2i, 2i+1),
// out[8 + i] = (16 + 2i, 17 + 2i).  v_permlane32_swap exchanges the upper half of its first operand with the lower half of its
// second: ([a.lo, a.hi], [b.lo, b.hi]) -> ([a.lo, b.lo], [a.hi, b.hi]) (tools/micro/permlane32_swap.hip).
// Written as inline assembly: with __builtin_amdgcn_permlane32_swap this compiler (ROCm 7.2 clang) emitted the instruction but
// wired every user of the SECOND result to the first (the sigma layer then saw channels 0..15 twice; found with a register
// dump against the oracle's stage vectors).  The s_nop covers the one hazard LLVM pads for this instruction (a VALU write of an
// operand needs two wait states before the swap reads it; GCNHazardRecognizer::checkPermlaneHazards) -- inside an asm statement
// nobody else would.
DEV void interleave16(const float* __restrict__ in, float* __restrict__ out) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = in[2 * i], b = in[2 * i + 1];
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        out[i] = a;
        out[8 + i] = b;
    }
}

// sigma feature layer, reference order.  fv[64]: k-step t of the layer = level t >> 4, channels 2 (t & 15) + half
DEV void geo_eval_ref(const float* __restrict__ lds, int lane, const float (&fv)[64], float (&sf)[32]) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    f32x4 bq_g0[4], bq_g1[4];
    f32x16 g0 = mfma_tile_bias<64>(wtile<gpl::GEO>(lds, 0), lane, fv, zero_tile(), bias_ptr<gpl::GEO>(lds, 0, half), bq_g0);
    f32x16 g1 = mfma_tile_bias<64>(wtile<gpl::GEO>(lds, 1), lane, fv, zero_tile(), bias_ptr<gpl::GEO>(lds, 1, half), bq_g1);
    elur_n<16>(g0, bq_g0, sf);
    elur_n<16>(g1, bq_g1, sf + 16);
}

// the same layer on all-zero volume features: ELU(0 + bias), no matrix work
DEV void geo_bias_ref(const float* __restrict__ lds, int lane, float (&sf)[32]) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    f32x16 z0 = zero_tile(), z1 = zero_tile();
    f32x4 b0[4], b1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        b0[q] = *reinterpret_cast<const f32x4*>(bias_ptr<gpl::GEO>(lds, 0, half) + 4 * q);
        b1[q] = *reinterpret_cast<const f32x4*>(bias_ptr<gpl::GEO>(lds, 1, half) + 4 * q);
    }
    elur_n<16>(z0, b0, sf);
    elur_n<16>(z1, b1, sf + 16);
}

// The rest of NeRFHead.forward in the reference's order.  x[v][18]: slot t of half h = element gpr::ref35(t, h) of view v's
// [r, g, b, f0..f31] (slot 1 of half 1 is the zero pad).
// skip_colour: the caller does not need rgb where the density is zero (no `raw` output): when the density of ALL 32 samples of the
// step is exactly 0 (nn.ReLU; masked_fill) every weight alpha * T of the step is 0 and 0 * rgb adds exactly 0 to every map for any
// finite rgb, so the colour branch -- 434 of the step's 748 MFMAs -- is not evaluated.  Same bits; a trained model's empty space.
// fused_mean_variance (trainhead.py:20-24): x.mean(-2) = sum / 3, then mean((x - mean)^2) = sum / 3
DEV void mean_var_ref(const float (&x)[NV][18], float (&mv)[36]) {
#pragma unroll
    for (int t = 0; t < 18; t += 2) {
        const f32x2 x0 = {x[0][t], x[0][t + 1]}, x1 = {x[1][t], x[1][t + 1]}, x2 = {x[2][t], x[2][t + 1]};
        const f32x2 m = div3((x0 + x1) + x2);
        const f32x2 a = x0 - m, b = x1 - m, c = x2 - m;
        const f32x2 v = div3((a * a + b * b) + c * c);
        mv[t] = m[0]; mv[t + 1] = m[1];
        mv[18 + t] = v[0]; mv[18 + t + 1] = v[1];
    }
}

// density branch alone (the deferred-colour sample loop, render_tile): sigma of this lane's sample
DEV void mlp_density_ref(const float* __restrict__ lds, int lane, const float (&sf)[32], const float (&mv)[36], float nvalid, float& sigma) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    float d1in[68];
#pragma unroll
    for (int r = 0; r < 32; ++r) d1in[r] = sf[r];
#pragma unroll
    for (int t = 0; t < 36; ++t) d1in[32 + t] = mv[t];

    // ---- density branch 134 -> 64 -> 32 -> 16 -> 1 (trainhead.py:102-110,133-137) ----
    {
        f32x4 bq_a0[4];
        f32x16 a0 = mfma_tile_bias<68>(wtile<gpl::D1>(lds, 0), lane, d1in, zero_tile(), bias_ptr<gpl::D1>(lds, 0, half), bq_a0);
        f32x4 bq_a1[4];
        f32x16 a1 = mfma_tile_bias<68>(wtile<gpl::D1>(lds, 1), lane, d1in, zero_tile(), bias_ptr<gpl::D1>(lds, 1, half), bq_a1);
        float h1[32];
        elur_n<16>(a0, bq_a0, h1);
        elur_n<16>(a1, bq_a1, h1 + 16);
        f32x4 bq_a2[4];
        f32x16 a2 = mfma_tile_bias<32>(wtile<gpl::D2>(lds, 0), lane, h1, zero_tile(), bias_ptr<gpl::D2>(lds, 0, half), bq_a2);
        float h2[16];
        elur_n<16>(a2, bq_a2, h2);
        f32x4 bq_a3[4];
        f32x16 a3 = mfma_tile_bias<16>(wtile<gpl::D3>(lds, 0), lane, h2, zero_tile(), bias_ptr<gpl::D3>(lds, 0, half), bq_a3);
        float h3[8];
        elur_n<8>(a3, bq_a3, h3);
        // 16 -> 1 on the VALU: this half holds features 2r + half, r < 8; two half sums, then the bias
        const float* w4 = lds + gpl::D4_W + half * 8;
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) part = fmaf(w4[r], h3[r], part);
        float s = part + __shfl_xor(part, 32) + lds[gpl::D4_B];
        s = fmaxf(s, 0.f);                              // nn.ReLU
        sigma = (nvalid < 1.f) ? 0.f : s;               // masked_fill(num_valid_obs < 1, 0)
    }
}

// colour branch alone: rgb of this lane's sample from the three views' 35-vectors (and their mean / variance)
DEV void mlp_colour_ref(const float* __restrict__ lds, int lane, const float (&x)[NV][18], const float (&mv)[36], float (&rgb)[3], Stamps& st) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    // ---- colour branch (trainhead.py:85-100,131,139-143) ----
    // base_fc.0 on [mean, var, x_v]: k = 0..69 ([mean, var]) is the same for the three views, computed once; each view's chain
    // goes on from there over its own 35 columns, then the bias
    f32x16 s0 = mfma_tile_from<36>(wtile<gpl::BS>(lds, 0), lane, mv, zero_tile());
    f32x16 s1 = mfma_tile_from<36>(wtile<gpl::BS>(lds, 1), lane, mv, zero_tile());
    float y[48];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        asm volatile("" : "+v"(lane));
        f32x4 bq_a0[4];
        f32x16 a0 = mfma_tile_bias<18>(wtile<gpl::BV>(lds, 0), lane, x[v], s0, bias_ptr<gpl::BS>(lds, 0, half), bq_a0);
        f32x4 bq_a1[4];
        f32x16 a1 = mfma_tile_bias<18>(wtile<gpl::BV>(lds, 1), lane, x[v], s1, bias_ptr<gpl::BS>(lds, 1, half), bq_a1);
        float h1[32];
        elur_n<16>(a0, bq_a0, h1);
        elur_n<16>(a1, bq_a1, h1 + 16);
        f32x4 bq_a2[4];
        f32x16 a2 = mfma_tile_bias<32>(wtile<gpl::B2>(lds, 0), lane, h1, zero_tile(), bias_ptr<gpl::B2>(lds, 0, half), bq_a2);
        float xb[16], xs[16];
        elur_n<16>(a2, bq_a2, xb);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                                                   // x * 1.0 / num_views (:140)
            const f32x2 q = div3(f32x2{xb[r], xb[r + 1]});
            xs[r] = q[0]; xs[r + 1] = q[1];
        }
        f32x4 bq_t1[4];
        f32x16 t1 = mfma_tile_bias<16>(wtile<gpl::V1>(lds, 0), lane, xs, zero_tile(), bias_ptr<gpl::V1>(lds, 0, half), bq_t1);
        float u1[16];
        elur_n<16>(t1, bq_t1, u1);
        f32x4 bq_t2[4];
        f32x16 t2 = mfma_tile_bias<16>(wtile<gpl::V2>(lds, 0), lane, u1, zero_tile(), bias_ptr<gpl::V2>(lds, 0, half), bq_t2);
        elur_n<16>(t2, bq_t2, y + 16 * v);
#pragma unroll
        for (int r = 0; r < 16; ++r) y[16 * v + r] = xb[r] + y[16 * v + r];                 // x = x + x_vis
    }
    STAMP(st, 4);
    {
        f32x4 bq_c1[4];
        f32x16 c1 = mfma_tile_bias<48>(wtile<gpl::R1>(lds, 0), lane, y, zero_tile(), bias_ptr<gpl::R1>(lds, 0, half), bq_c1);
        float h1[16];
        elur_n<16>(c1, bq_c1, h1);
        f32x4 bq_c2[4];
        f32x16 c2 = mfma_tile_bias<16>(wtile<gpl::R2>(lds, 0), lane, h1, zero_tile(), bias_ptr<gpl::R2>(lds, 0, half), bq_c2);
        float e[8];
        elur_n<8>(c2, bq_c2, e);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float* w3 = lds + gpl::R3_W + o * 16 + half * 8;
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) part = fmaf(w3[r], e[r], part);
            const float s = part + __shfl_xor(part, 32) + lds[gpl::R3_B + o];
            rgb[o] = 1.f / (1.f + fast_exp(-s));        // .sigmoid()
        }
    }
    STAMP(st, 5);
}

DEV void mlp_eval_ref(const float* __restrict__ lds, int lane, const float (&sf)[32], const float (&x)[NV][18],
                      float nvalid, float& sigma, float (&rgb)[3], Stamps& st, const bool skip_colour = false) {
    float mv[36];
    mean_var_ref(x, mv);
    mlp_density_ref(lds, lane, sf, mv, nvalid, sigma);
    if (skip_colour && __all(sigma == 0.f)) { rgb[0] = 0.f; rgb[1] = 0.f; rgb[2] = 0.f; return; }
    STAMP(st, 3);
    mlp_colour_ref(lds, lane, x, mv, rgb, st);
}

// ---------------------------------------------------------------------------------------------
// split-precision MLP core (GPNERF_FLAG_SPLIT_F16): v_mfma_f32_32x32x16_f16 on hi/lo f16 halves, see head_layout.h
// ---------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

struct Frag { h8 hi, lo; };        // one 16-deep k-step of the B operand: 8 values of this lane half, as f16 hi + lo

DEV unsigned pk_rtz(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }

// lo pair of (x0, x1) given their packed hi halves w: f16(x0 - w.lo) in the low half, f16(x1 - w.hi) in the high half.
// v_fma_mixlo_f16 / v_fma_mixhi_f16 read the f16 half of w directly, subtract in f32 and write the rounded f16 result
// into one half of the destination -- no conversion back to f32, no separate pack.
DEV unsigned lo_pair(unsigned w, float x0, float x1) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w), "v"(x1));
    return r;
}

// Range guard of the split form (GPNERF_FLAG_SPLIT_GUARD): every value that becomes an MFMA operand passes through guard_n().
// It costs one v_max3_f32 per two values into a short-lived maximum, one compare and one LDS store per group; nothing lives
// across the sample loop: a lane that sees a value at or beyond the f16 range (or a NaN) writes to its wave's LDS slot (indexed
// by the hardware wave slot, HW_ID[5:0]), the others to a dummy word; render_tile() reads the slot back at the end of the tile.  A flagged tile is rendered again by
// the fp32 form (gpnerf_render_fused).
struct NoGuard {};
struct Guard {};
constexpr float F16_RANGE = 65504.f;
constexpr int GUARD_LDS_SLOTS = 64;            // followed by as many dummy words
extern __shared__ __attribute__((aligned(16))) float g_dyn_lds[];       // the kernel's dynamic LDS (same base as its `lds`)
DEV unsigned wave_slot() {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 6)" : "=s"(id));
    return id;
}
DEV unsigned* guard_slot() { return reinterpret_cast<unsigned*>(g_dyn_lds) + gph::BLOB_WORDS + wave_slot(); }
template <int N> DEV void guard_n(NoGuard&, const float*) {}
template <int N> DEV void guard_n(Guard&, const float* v) {
    static_assert(N % 2 == 0, "");
    float m = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
    for (int i = 2; i < N; i += 2) m = fmaxf(fmaxf(m, fabsf(v[i])), fabsf(v[i + 1]));
    // Branch-free on purpose: lanes in range write a dummy word.  With `if (__any(...))` here -- a branch in the middle of the
    // layer chain, never taken -- round 2's build of this kernel gave results 2e-5 off and different from run to run.  What IS
    // known (round 3): gfx950 does not interlock a consumer of an MFMA's destination registers inside the MFMA's shadow, LLVM
    // pads only the consumers it can see, and this file's inline-asm conversions (lo_pair) are opaque to it -- demonstrated on
    // hardware (tools/micro/mfma_asm_hazard.hip: 65 535 of 65 536 results stale) and gated by a static check of the shipped ISA
    // (tools/isa_mfma_hazards.py, tests/test_abi.py).  What is NOT known: whether that was the failing pair in the round-2
    // schedule -- that source variant was never committed, and the branch form re-created in round 3 (a diagnostic build, git
    // history) was deterministic and checker-clean.  The mechanism class is established, the instance is not.
    unsigned* const sl = guard_slot();
    (m < F16_RANGE ? sl + GUARD_LDS_SLOTS : sl)[0] = 1u;
}

// A VGPR written by a VALU instruction must not be read as an MFMA source operand in the very next issue slot: gfx950 does not
// interlock that pair (tools/micro/asm_producer_hazards.hip: v_fma_mixhi_f16 directly in front of the v_mfma_f32_32x32x16_f16 that
// reads its result hands the MFMA the register's PREVIOUS contents in 129 032 of 131 072 lanes; one wait state cures it).  LLVM
// pads the producers it can see; lo_pair's two instructions are inline assembly, invisible to it, and whether one lands directly
// in front of its MFMA is the scheduler's accident -- round 5's "deferred colour branch of the split form comes out 10-30 % wrong
// in one build" (DESIGN.md 4.1).  So the pairs that become an MFMA operand pass through ONE more opaque statement that carries
// the wait state itself: every reader of the operand depends on it, it depends on every lo_pair, hence >= 1 wait state between the
// last conversion and the MFMA whatever the schedule.  tools/isa_mfma_hazards.py checks the built code (tests/test_abi.py).
DEV void settle_operand(u32x4& lo) { asm("s_nop 0" : "+v"(lo)); }

// x = hi + lo with hi = f16(x) toward zero (never overflows to inf), lo = f16(x - hi): ~22 significant bits
template <class G>
DEV Frag make_frag(const float* v, G& g) {
    u32x4 H, Lo;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned w = pk_rtz(v[2 * p], v[2 * p + 1]);
        H[p] = w;
        Lo[p] = lo_pair(w, v[2 * p], v[2 * p + 1]);
    }
    if constexpr (SPIN) asm volatile("" : "+v"(H), "+v"(Lo));      // complete before the MFMA run that follows (no VALU between MFMAs)
    settle_operand(Lo);
    Frag f;
    f.hi = __builtin_bit_cast(h8, H);
    f.lo = __builtin_bit_cast(h8, Lo);
    return f;
}

template <class G>
DEV Frag make_frag2(float a, float b, G& g) {              // the rgb k-step: two live slots, six zero pads
    const unsigned w = pk_rtz(a, b);
    u32x4 H = {w, 0u, 0u, 0u}, Lo = {lo_pair(w, a, b), 0u, 0u, 0u};
    settle_operand(Lo);
    Frag f;
    f.hi = __builtin_bit_cast(h8, H);
    f.lo = __builtin_bit_cast(h8, Lo);
    return f;
}

template <int L>
DEV const unsigned* wstep(const unsigned* lw, int m, int s) { return lw + gph::w_off(L) + (m * gph::NS[L] + s) * gph::STEP_WORDS; }

template <int L>
DEV f32x16 bias_tile_s(const unsigned* lw, int m, int half) {
    const float* p = reinterpret_cast<const float*>(lw) + gph::b_off(L) + m * 32 + half * 16;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
        acc[4 * q + 0] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
    }
    return acc;
}

// acc += W_tile[:, 16 k] . B  as  Wlo.Bhi + Whi.Blo + Whi.Bhi, NSTEP k-steps in one uninterrupted MFMA run: the next
// step's hi/lo weight reads are issued before this step's three MFMAs (same ordering device as mfma_tile)
template <int L, int NSTEP>
DEV void mfma_steps(const unsigned* lw, int m, int s0, int lane, const Frag* b, f32x16& acc) {
    typedef const __attribute__((address_space(3))) unsigned* lds_ptr;
    typedef const __attribute__((address_space(3))) u32x4* lds_vec;
    lds_ptr p = (lds_ptr)wstep<L>(lw, m, s0) + lane * 4;
    u32x4 ah = *reinterpret_cast<lds_vec>(p), al = *reinterpret_cast<lds_vec>(p + 256);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        u32x4 nh = ah, nl = al;
        if (s + 1 < NSTEP) {
            nh = *reinterpret_cast<lds_vec>(p + (s + 1) * gph::STEP_WORDS);
            nl = *reinterpret_cast<lds_vec>(p + (s + 1) * gph::STEP_WORDS + 256);
        }
        asm volatile("" : "+v"(p), "+v"(acc));
        const h8 wh = __builtin_bit_cast(h8, ah), wl = __builtin_bit_cast(h8, al);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, b[s].hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b[s].lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b[s].hi, acc, 0, 0, 0);
        ah = nh; al = nl;
    }
}

// ELU of an accumulator tile -> its two B-operand k-steps (and optionally the fp32 values)
template <class G>
DEV void tile_frags(G& g, f32x16& a, Frag* out2, float* keep = nullptr) {
    float t[16];
    elus_n<16, SPIN>(a, t);
#pragma unroll
    for (int r = 0; r < 16; ++r) { if (keep) keep[r] = t[r]; }
    guard_n<16>(g, t);
    out2[0] = make_frag(t, g);
    out2[1] = make_frag(t + 8, g);
}

template <class G>
DEV void geo_eval_s(G& g, const unsigned* __restrict__ lw, int lane, const float (&fv)[64], Frag (&sff)[4]) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    Frag in[8];
    guard_n<64>(g, fv);
#pragma unroll
    for (int s = 0; s < 8; ++s) in[s] = make_frag(fv + 8 * s, g);
    f32x16 g0 = bias_tile_s<gpl::GEO>(lw, 0, half), g1 = bias_tile_s<gpl::GEO>(lw, 1, half);
    mfma_steps<gpl::GEO, 8>(lw, 0, 0, lane, in, g0);
    mfma_steps<gpl::GEO, 8>(lw, 1, 0, lane, in, g1);
    tile_frags(g, g0, sff);
    tile_frags(g, g1, sff + 2);
}

// cross-view mean / variance (trainhead.py:20-24) and their k-steps: mean(2 + rgb), var(2 + rgb)
template <class G>
DEV void mean_var_s(G& g, const float (&x)[NV][18], Frag (&mvf)[6]) {
    float mv[36];
#pragma unroll
    for (int t = 0; t < 18; ++t) {
        const float m = ((x[0][t] + x[1][t]) + x[2][t]) * (1.f / 3.f);
        const float a = x[0][t] - m, b = x[1][t] - m, c = x[2][t] - m;
        mv[t] = m;
        mv[18 + t] = ((a * a + b * b) + c * c) * (1.f / 3.f);
    }
    guard_n<36>(g, mv);
    mvf[0] = make_frag(mv, g); mvf[1] = make_frag(mv + 8, g); mvf[2] = make_frag2(mv[16], mv[17], g);
    mvf[3] = make_frag(mv + 18, g); mvf[4] = make_frag(mv + 26, g); mvf[5] = make_frag2(mv[34], mv[35], g);
}

// density branch alone (the deferred-colour sample loop, render_tile)
template <class G>
DEV void mlp_density_s(G& g, const unsigned* __restrict__ lw, int lane, const Frag (&sff)[4], const Frag (&mvf)[6], float nvalid, float& sigma) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    const float* lf = reinterpret_cast<const float*>(lw);
    // density branch (trainhead.py:102-110,133-137)
    {
        f32x16 a0 = bias_tile_s<gpl::D1>(lw, 0, half), a1 = bias_tile_s<gpl::D1>(lw, 1, half);
        mfma_steps<gpl::D1, 4>(lw, 0, 0, lane, sff, a0);
        mfma_steps<gpl::D1, 6>(lw, 0, 4, lane, mvf, a0);
        mfma_steps<gpl::D1, 4>(lw, 1, 0, lane, sff, a1);
        mfma_steps<gpl::D1, 6>(lw, 1, 4, lane, mvf, a1);
        Frag h1[4];
        tile_frags(g, a0, h1);
        tile_frags(g, a1, h1 + 2);
        f32x16 a2 = bias_tile_s<gpl::D2>(lw, 0, half);
        mfma_steps<gpl::D2, 4>(lw, 0, 0, lane, h1, a2);
        Frag h2[2];
        tile_frags(g, a2, h2);
        f32x16 a3 = bias_tile_s<gpl::D3>(lw, 0, half);
        mfma_steps<gpl::D3, 2>(lw, 0, 0, lane, h2, a3);
        const float* w4 = lf + gph::D4_W + half * 8;
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) part = fmaf(w4[r], elus(a3[r]), part);
        float s = part + __shfl_xor(part, 32) + lf[gph::D4_B];
        s = fmaxf(s, 0.f);
        sigma = (nvalid < 1.f) ? 0.f : s;
    }
}

// colour branch alone
template <class G>
DEV void mlp_colour_s(G& g, const unsigned* __restrict__ lw, int lane, const float (&x)[NV][18], const Frag (&mvf)[6], float (&rgb)[3], Stamps& st) {
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5;
    const float* lf = reinterpret_cast<const float*>(lw);
    // colour branch (trainhead.py:85-100,131,139-143)
    f32x16 s0 = bias_tile_s<gpl::BS>(lw, 0, half), s1 = bias_tile_s<gpl::BS>(lw, 1, half);
    mfma_steps<gpl::BS, 6>(lw, 0, 0, lane, mvf, s0);
    mfma_steps<gpl::BS, 6>(lw, 1, 0, lane, mvf, s1);
    Frag yf[6];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        asm volatile("" : "+v"(lane));
        Frag xf[3];
        guard_n<18>(g, x[v]);
        xf[0] = make_frag(x[v], g); xf[1] = make_frag(x[v] + 8, g); xf[2] = make_frag2(x[v][16], x[v][17], g);
        f32x16 a0 = s0, a1 = s1;
        mfma_steps<gpl::BV, 3>(lw, 0, 0, lane, xf, a0);
        mfma_steps<gpl::BV, 3>(lw, 1, 0, lane, xf, a1);
        Frag h1[4];
        tile_frags(g, a0, h1);
        tile_frags(g, a1, h1 + 2);
        f32x16 a2 = bias_tile_s<gpl::B2>(lw, 0, half);
        mfma_steps<gpl::B2, 4>(lw, 0, 0, lane, h1, a2);
        float xb[16];
        Frag xs[2];
        tile_frags(g, a2, xs, xb);                            // x * 1.0 / num_views rides on vis_fc.0's packed weights
        f32x16 t1 = bias_tile_s<gpl::V1>(lw, 0, half);
        mfma_steps<gpl::V1, 2>(lw, 0, 0, lane, xs, t1);
        Frag u1[2];
        tile_frags(g, t1, u1);
        f32x16 t2 = bias_tile_s<gpl::V2>(lw, 0, half);
        mfma_steps<gpl::V2, 2>(lw, 0, 0, lane, u1, t2);
        float y[16];
        elus_n<16, SPIN>(t2, y);
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] += xb[r];
        guard_n<16>(g, y);
        yf[2 * v] = make_frag(y, g);
        yf[2 * v + 1] = make_frag(y + 8, g);
    }
    STAMP(st, 4);
    {
        f32x16 c1 = bias_tile_s<gpl::R1>(lw, 0, half);
        mfma_steps<gpl::R1, 6>(lw, 0, 0, lane, yf, c1);
        Frag h1[2];
        tile_frags(g, c1, h1);
        f32x16 c2 = bias_tile_s<gpl::R2>(lw, 0, half);
        mfma_steps<gpl::R2, 2>(lw, 0, 0, lane, h1, c2);
        float e[8];
        elus_n<8>(c2, e);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float* w3 = lf + gph::R3_W + o * 16 + half * 8;
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) part = fmaf(w3[r], e[r], part);
            const float s = part + __shfl_xor(part, 32) + lf[gph::R3_B + o];
            rgb[o] = 1.f / (1.f + fast_exp(-s));
        }
    }
    STAMP(st, 5);
}

template <class G>
DEV void mlp_eval_s(G& g, const unsigned* __restrict__ lw, int lane, const Frag (&sff)[4], const float (&x)[NV][18], float nvalid,
                    float& sigma, float (&rgb)[3], Stamps& st) {
    Frag mvf[6];
    mean_var_s(g, x, mvf);
    mlp_density_s(g, lw, lane, sff, mvf, nvalid, sigma);
    STAMP(st, 3);
    mlp_colour_s(g, lw, lane, x, mvf, rgb, st);
}

// ---------------------------------------------------------------------------------------------
// feature gathers (channels-last sources)
// ---------------------------------------------------------------------------------------------
struct Axis {
    unsigned i0, i1;   // clamped tap indices
    float w0, w1;      // weights, zeroed for out-of-range taps (padding_mode='zeros')
};

// F.grid_sample coordinate handling, align_corners=True: index = ((g + 1) / 2) * (size - 1)
// The continuous index is first clamped to [-1, size]: every tap of an index outside that range is out of bounds anyway
// (zero weight), -1 and size themselves give the same taps and weights as before the clamp, and NaN / +-inf land on a bound
// (v_med3_f32 returns the smallest operand when one is NaN), so the float -> int conversion below is always in range and the
// bounds tests can be two unsigned integer compares instead of four float compares.  Indices and weights are unchanged.
DEV Axis axis_taps(float g, int size) {
    const float sm1 = (float)(size - 1);
    const float ix = __builtin_amdgcn_fmed3f(((g + 1.f) * 0.5f) * sm1, -1.f, (float)size);
    const float f0 = floorf(ix);
    const float t = ix - f0;
    const int j0 = (int)f0, j1 = j0 + 1;
    const bool v0 = (unsigned)j0 < (unsigned)size;
    const bool v1 = (unsigned)j1 < (unsigned)size;
    Axis a;
    a.i0 = v0 ? (unsigned)j0 : 0u;
    a.i1 = v1 ? (unsigned)j1 : 0u;
    a.w0 = v0 ? 1.f - t : 0.f;
    a.w1 = v1 ? t : 0.f;
    return a;
}

// Tap addresses are 32-bit BYTE offsets from a uniform base (to_framek() guarantees every tensor is below 4 GiB and every
// index factor below 2^24): the products are v_mad_u32_u24 (full rate; v_mul_lo_u32 is quarter rate) and the loads take
// the base from SGPRs plus a 32-bit VGPR offset, so no 64-bit address arithmetic runs on the VALU.
DEV unsigned mad24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }
DEV const float* at_byte(const float* base, unsigned byte_off) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// 16 channels of one tap: 8 v_pk_fma_f32 (the tap weight is broadcast by op_sel)
DEV void fma16(const float* __restrict__ p, float w, float* f) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    const f32x2 w2 = {w, w};
    const f32x2 src[8] = {{a[0], a[1]}, {a[2], a[3]}, {b[0], b[1]}, {b[2], b[3]}, {c[0], c[1]}, {c[2], c[3]}, {d[0], d[1]}, {d[2], d[3]}};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x2 r = __builtin_elementwise_fma(src[i], w2, f32x2{f[2 * i], f[2 * i + 1]});
        f[2 * i] = r[0]; f[2 * i + 1] = r[1];
    }
}

// the same, multiply THEN add (two roundings): ATen's scalar 3-D grid_sample kernel accumulates `out += val * weight` without
// contraction, and the reference-order form follows it -- one ulp of a volume feature is 3 x the oracle's distance on the
// trained-like fixtures once the head has amplified it (oracle/kernel_order.inc TRIFMA)
DEV void muladd16(const float* __restrict__ p, float w, float* f) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    const f32x2 w2 = {w, w};
    const f32x2 src[8] = {{a[0], a[1]}, {a[2], a[3]}, {b[0], b[1]}, {b[2], b[3]}, {c[0], c[1]}, {c[2], c[3]}, {d[0], d[1]}, {d[2], d[3]}};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x2 r = f32x2{f[2 * i], f[2 * i + 1]} + src[i] * w2;
        f[2 * i] = r[0]; f[2 * i + 1] = r[1];
    }
}

// trilinear sample of one level [D][H][W][32] at normalised (gx,gy,gz) -> this half's 16 channels
// (libs/nerfheads/networks/SparseConvNet.py:113-116).  UNFUSED: the reference-order form's multiply-then-add taps.
template <bool UNFUSED = false>
DEV void gather_volume(const float* __restrict__ vol, int D, int H, int W, float gx, float gy, float gz, int half,
                       float* f) {
    const Axis ax = axis_taps(gx, W), ay = axis_taps(gy, H), az = axis_taps(gz, D);
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = 0.f;
    const unsigned zi[2] = {az.i0, az.i1}, yi[2] = {ay.i0, ay.i1};
    const float zw[2] = {az.w0, az.w1}, yw[2] = {ay.w0, ay.w1}, xw[2] = {ax.w0, ax.w1};
    const unsigned row_bytes = (unsigned)W * 128u;                              // one x-row: W voxels x 32 channels x 4 B
    const unsigned xb[2] = {ax.i0 * 128u + (unsigned)half * 64u, ax.i1 * 128u + (unsigned)half * 64u};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned rowb = __umul24(mad24(zi[a], (unsigned)H, yi[b]), row_bytes);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if constexpr (UNFUSED) muladd16(at_byte(vol, rowb + xb[e]), (xw[e] * yw[b]) * zw[a], f);
                else fma16(at_byte(vol, rowb + xb[e]), (xw[e] * yw[b]) * zw[a], f);
            }
        }
}

// The same sample with all eight taps' 32 loads issued before the first is used (128 registers: for the sample loops that
// have them free at the top of a step -- the deferred-colour form): one round trip per level.  Same arithmetic, same order.
template <bool UNFUSED>
DEV void gather_volume_batched(const float* __restrict__ vol, int D, int H, int W, float gx, float gy, float gz, int half, float* f) {
    const Axis ax = axis_taps(gx, W), ay = axis_taps(gy, H), az = axis_taps(gz, D);
    const unsigned zi[2] = {az.i0, az.i1}, yi[2] = {ay.i0, ay.i1};
    const float zw[2] = {az.w0, az.w1}, yw[2] = {ay.w0, ay.w1}, xw[2] = {ax.w0, ax.w1};
    const unsigned row_bytes = (unsigned)W * 128u;
    const unsigned xb[2] = {ax.i0 * 128u + (unsigned)half * 64u, ax.i1 * 128u + (unsigned)half * 64u};
    f32x4 q[8][4];
    float wt[8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned rowb = __umul24(mad24(zi[a], (unsigned)H, yi[b]), row_bytes);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4* p = reinterpret_cast<const f32x4*>(at_byte(vol, rowb + xb[e]));
#pragma unroll
                for (int i = 0; i < 4; ++i) q[4 * a + 2 * b + e][i] = p[i];
                wt[4 * a + 2 * b + e] = (xw[e] * yw[b]) * zw[a];
            }
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const f32x2 w2 = {wt[t], wt[t]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 r0, r1;
            if constexpr (UNFUSED) {           // multiply, then add (muladd16)
                r0 = f32x2{f[4 * i], f[4 * i + 1]} + f32x2{q[t][i][0], q[t][i][1]} * w2;
                r1 = f32x2{f[4 * i + 2], f[4 * i + 3]} + f32x2{q[t][i][2], q[t][i][3]} * w2;
            } else {                           // fma16
                r0 = __builtin_elementwise_fma(f32x2{q[t][i][0], q[t][i][1]}, w2, f32x2{f[4 * i], f[4 * i + 1]});
                r1 = __builtin_elementwise_fma(f32x2{q[t][i][2], q[t][i][3]}, w2, f32x2{f[4 * i + 2], f[4 * i + 3]});
            }
            f[4 * i] = r0[0]; f[4 * i + 1] = r0[1]; f[4 * i + 2] = r1[0]; f[4 * i + 3] = r1[1];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The batched gather in two halves, so that the sample loop can put matrix work between a level's loads and their use: the
// deferred-colour step of the reference-order form runs level l's 32 MFMAs of the sigma feature layer while level l + 1's 32
// loads are in flight (render_tile).  vol_finish accumulates as muladd16 does: multiply, then add, taps in the same order.
struct VolTaps {
    f32x4 q[8][4];
    float wt[8];
};
DEV void vol_issue(const float* __restrict__ vol, int D, int H, int W, float gx, float gy, float gz, int half, VolTaps& t) {
    const Axis ax = axis_taps(gx, W), ay = axis_taps(gy, H), az = axis_taps(gz, D);
    const unsigned zi[2] = {az.i0, az.i1}, yi[2] = {ay.i0, ay.i1};
    const float zw[2] = {az.w0, az.w1}, yw[2] = {ay.w0, ay.w1}, xw[2] = {ax.w0, ax.w1};
    const unsigned row_bytes = (unsigned)W * 128u;
    const unsigned xb[2] = {ax.i0 * 128u + (unsigned)half * 64u, ax.i1 * 128u + (unsigned)half * 64u};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned rowb = __umul24(mad24(zi[a], (unsigned)H, yi[b]), row_bytes);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4* p = reinterpret_cast<const f32x4*>(at_byte(vol, rowb + xb[e]));
#pragma unroll
                for (int i = 0; i < 4; ++i) t.q[4 * a + 2 * b + e][i] = p[i];
                t.wt[4 * a + 2 * b + e] = (xw[e] * yw[b]) * zw[a];
            }
        }
}
DEV void vol_finish(const VolTaps& t, float* f) {
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x2 w2 = {t.wt[k], t.wt[k]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 r0 = f32x2{f[4 * i], f[4 * i + 1]} + f32x2{t.q[k][i][0], t.q[k][i][1]} * w2;
            const f32x2 r1 = f32x2{f[4 * i + 2], f[4 * i + 3]} + f32x2{t.q[k][i][2], t.q[k][i][3]} * w2;
            f[4 * i] = r0[0]; f[4 * i + 1] = r0[1]; f[4 * i + 2] = r1[0]; f[4 * i + 3] = r1[1];
        }
    }
}

// The sigma feature layer is linear in the volume features and trilinear sampling is linear in the voxels, so
//   W (sum_t w_t v_t) = sum_t w_t (W v_t):
// gpnerf_fold_volumes applies out_geometry_fc's 32 columns of level l to every voxel of level l once per frame (64 values per
// voxel, laid out [half][tile][16] = the accumulator registers of the two output tiles), and the sample loop interpolates the
// layer's PRE-ACTIVATION directly: 32 values per lane and tap instead of 16, and none of the layer's 128 MFMAs per step.
// 32 values of one tap, already in registers: 16 v_pk_fma_f32
DEV void fma32(const f32x4 (&q)[8], float w, f32x16& g0, f32x16& g1) {
    const f32x2 w2 = {w, w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 a = q[i], b = q[4 + i];
        const f32x2 r0 = __builtin_elementwise_fma(f32x2{a[0], a[1]}, w2, f32x2{g0[4 * i], g0[4 * i + 1]});
        const f32x2 r1 = __builtin_elementwise_fma(f32x2{a[2], a[3]}, w2, f32x2{g0[4 * i + 2], g0[4 * i + 3]});
        const f32x2 r2 = __builtin_elementwise_fma(f32x2{b[0], b[1]}, w2, f32x2{g1[4 * i], g1[4 * i + 1]});
        const f32x2 r3 = __builtin_elementwise_fma(f32x2{b[2], b[3]}, w2, f32x2{g1[4 * i + 2], g1[4 * i + 3]});
        g0[4 * i] = r0[0]; g0[4 * i + 1] = r0[1]; g0[4 * i + 2] = r1[0]; g0[4 * i + 3] = r1[1];
        g1[4 * i] = r2[0]; g1[4 * i + 1] = r2[1]; g1[4 * i + 2] = r3[0]; g1[4 * i + 3] = r3[1];
    }
}
// trilinear sample of one folded level [D][H][W][2 halves][2 tiles][16] -> added to this half's two accumulator tiles.
// A tap is 128 bytes per lane = 32 registers, and left to itself the compiler walks the taps one at a time, a full L2 round trip
// each (32 per step: the kernel then waits more than it computes).  The taps go four at a time: 32 loads in flight, then their FMAs.
constexpr int FOLD_BATCH = 4;
// Levels FOLD_FROM .. 3 are folded.  Measured on the 512x512x64 frame (every variant with the batched view gather): none 13.73 ms,
// the coarsest 13.86, the two coarse levels 13.22, three 13.43, all four 13.59 (+0.17 ms and 400 MB to fold the 1.4 M voxels of
// level 0).  A folded tap is twice the bytes per lane, and what the matrix pipe saves the gathers' round trips take back -- except
// on the coarse levels, whose tables (6 MB folded) stay in the L2s and whose 32 rays share a handful of voxels.
constexpr int FOLD_FROM = GPNERF_FOLD_FIRST_LEVEL;
DEV void gather_folded(const float* __restrict__ vol, int D, int H, int W, float gx, float gy, float gz, int half,
                       f32x16& g0, f32x16& g1) {
    const Axis ax = axis_taps(gx, W), ay = axis_taps(gy, H), az = axis_taps(gz, D);
    const unsigned zi[2] = {az.i0, az.i1}, yi[2] = {ay.i0, ay.i1};
    const float zw[2] = {az.w0, az.w1}, yw[2] = {ay.w0, ay.w1}, xw[2] = {ax.w0, ax.w1};
    const unsigned row_bytes = (unsigned)W * 256u;                              // one x-row: W voxels x 64 values x 4 B
    const unsigned xb[2] = {ax.i0 * 256u + (unsigned)half * 128u, ax.i1 * 256u + (unsigned)half * 128u};
    unsigned off[8];
    float wt[8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned rowb = __umul24(mad24(zi[a], (unsigned)H, yi[b]), row_bytes);
#pragma unroll
            for (int e = 0; e < 2; ++e) { off[4 * a + 2 * b + e] = rowb + xb[e]; wt[4 * a + 2 * b + e] = (xw[e] * yw[b]) * zw[a]; }
        }
#pragma unroll
    for (int t0 = 0; t0 < 8; t0 += FOLD_BATCH) {
        f32x4 q[FOLD_BATCH][8];
#pragma unroll
        for (int t = 0; t < FOLD_BATCH; ++t) {
            const f32x4* p = reinterpret_cast<const f32x4*>(at_byte(vol, off[t0 + t]));
#pragma unroll
            for (int i = 0; i < 8; ++i) q[t][i] = p[i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < FOLD_BATCH; ++t) fma32(q[t], wt[t0 + t], g0, g1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// F.grid_sample of the single-channel occupancy volume (demo_render.py:274-279), same coordinates as the features
DEV float sample_occupancy(const float* __restrict__ occ, int D, int H, int W, float gx, float gy, float gz) {
    const Axis ax = axis_taps(gx, W), ay = axis_taps(gy, H), az = axis_taps(gz, D);
    const unsigned zi[2] = {az.i0, az.i1}, yi[2] = {ay.i0, ay.i1}, xi[2] = {ax.i0, ax.i1};
    const float zw[2] = {az.w0, az.w1}, yw[2] = {ay.w0, ay.w1}, xw[2] = {ax.w0, ax.w1};
    float v = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                v = fmaf(*at_byte(occ, (__umul24(mad24(zi[a], (unsigned)H, yi[b]), (unsigned)W) + xi[e]) * 4u), (xw[e] * yw[b]) * zw[a], v);
    return v;
}

struct ViewSample {
    float rgb[3];
    float valid;
};

// Only the view's validity test of Projector.compute (front of the camera and inside the image: gather_view's own expressions,
// so the same decision to the bit): what a step of exactly opaque rays still owes the outputs (ray_mask counts views).
template <class MP>
DEV float view_valid(MP M, int ih, int iw, float px, float py, float pz, bool neg) {
    const float hx = fmaf(M[2], pz, fmaf(M[1], py, M[0] * px)) + M[3];
    const float hy = fmaf(M[6], pz, fmaf(M[5], py, M[4] * px)) + M[7];
    const float hz = fmaf(M[10], pz, fmaf(M[9], py, M[8] * px)) + M[11];
    float u = hx / hz, w = hy / hz;
    u = fminf(fmaxf(u, -1e6f), 1e6f);
    w = fminf(fmaxf(w, -1e6f), 1e6f);
    const bool front = neg ? (hz < 0.f) : (hz > 0.f);
    const float wm1 = (float)iw - 1.f, hm1 = (float)ih - 1.f;
    const bool inb = (u <= wm1) && (u >= 0.f) && (w <= hm1) && (w >= 0.f);
    return (front && inb) ? 1.f : 0.f;
}

// Projector.compute for one view (libs/renders/BaseRender.py:301-324,296-299,283-294,352-362):
// project p, bilinear RGB from imgs[v] (NHWC4) and 16 feature channels from featmaps[v] (NHWC32).
template <bool BATCH = false, class MP>
DEV ViewSample gather_view(MP M, const float* __restrict__ img, int ih, int iw,
                           const float* __restrict__ fm, int fh, int fw, float px, float py, float pz, bool neg,
                           int half, float* f) {
    // (K4 P4) bmm [p, 1] (BaseRender.py:314): on the reference's CPU path an sgemm whose micro-kernel accumulates over k = 0..3 with
    // FMAs, k ascending, the last term (x 1) a plain add -- checked bit for bit against torch.bmm; with this order the pixel
    // coordinates, the in-bounds masks and the gathered view features are the reference's own bits (round 3 summed left to
    // right with separate multiplies and adds: one ulp of a pixel coordinate, 1.7e-5 in a trained-like feature)
    const float hx = fmaf(M[2], pz, fmaf(M[1], py, M[0] * px)) + M[3];
    const float hy = fmaf(M[6], pz, fmaf(M[5], py, M[4] * px)) + M[7];
    const float hz = fmaf(M[10], pz, fmaf(M[9], py, M[8] * px)) + M[11];
    float u = hx / hz, w = hy / hz;
    u = fminf(fmaxf(u, -1e6f), 1e6f);        // torch.clamp; a NaN lands out of bounds here as it does there
    w = fminf(fmaxf(w, -1e6f), 1e6f);
    const bool front = neg ? (hz < 0.f) : (hz > 0.f);
    const float wm1 = (float)iw - 1.f, hm1 = (float)ih - 1.f;
    const bool inb = (u <= wm1) && (u >= 0.f) && (w <= hm1) && (w >= 0.f);
    const float nx = 2.f * u / wm1 - 1.f, ny = 2.f * w / hm1 - 1.f;
    ViewSample s;
    s.valid = (front && inb) ? 1.f : 0.f;
    // All eight taps of the view (4 image texels, 4 x 64 B of the feature map) are loaded before the first is used: 20 loads in
    // flight and one round trip per view, where the compiler on its own walks the taps one at a time (BATCH: the folded form,
    // which has no matrix work between its two gather phases to cover them).
    if constexpr (BATCH) {
        const Axis ix = axis_taps(nx, iw), iy = axis_taps(ny, ih);
        const unsigned ir0 = __umul24(iy.i0, (unsigned)iw * 16u), ir1 = __umul24(iy.i1, (unsigned)iw * 16u);
        const unsigned ix0 = ix.i0 * 16u, ix1 = ix.i1 * 16u;
        const Axis ax = axis_taps(nx, fw), ay = axis_taps(ny, fh);
        const unsigned r0 = __umul24(ay.i0, (unsigned)fw * 128u), r1 = __umul24(ay.i1, (unsigned)fw * 128u);
        const unsigned x0 = ax.i0 * 128u + (unsigned)half * 64u, x1 = ax.i1 * 128u + (unsigned)half * 64u;
        const f32x4 nw = *reinterpret_cast<const f32x4*>(at_byte(img, ir0 + ix0));
        const f32x4 ne = *reinterpret_cast<const f32x4*>(at_byte(img, ir0 + ix1));
        const f32x4 sw = *reinterpret_cast<const f32x4*>(at_byte(img, ir1 + ix0));
        const f32x4 se = *reinterpret_cast<const f32x4*>(at_byte(img, ir1 + ix1));
        const unsigned fo[4] = {r0 + x0, r0 + x1, r1 + x0, r1 + x1};
        const float fwt[4] = {ax.w0 * ay.w0, ax.w1 * ay.w0, ax.w0 * ay.w1, ax.w1 * ay.w1};
        f32x4 q[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4* p = reinterpret_cast<const f32x4*>(at_byte(fm, fo[t]));
#pragma unroll
            for (int i = 0; i < 4; ++i) q[t][i] = p[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float wnw = ix.w0 * iy.w0, wne = ix.w1 * iy.w0, wsw = ix.w0 * iy.w1, wse = ix.w1 * iy.w1;
#pragma unroll
        for (int c = 0; c < 3; ++c) s.rgb[c] = fmaf(se[c], wse, fmaf(sw[c], wsw, fmaf(ne[c], wne, nw[c] * wnw)));
#pragma unroll
        for (int c = 0; c < 16; ++c) f[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x2 w2 = {fwt[t], fwt[t]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 a = __builtin_elementwise_fma(f32x2{q[t][i][0], q[t][i][1]}, w2, f32x2{f[4 * i], f[4 * i + 1]});
                const f32x2 b = __builtin_elementwise_fma(f32x2{q[t][i][2], q[t][i][3]}, w2, f32x2{f[4 * i + 2], f[4 * i + 3]});
                f[4 * i] = a[0]; f[4 * i + 1] = a[1]; f[4 * i + 2] = b[0]; f[4 * i + 3] = b[1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    } else {
        {   // RGB from the full-resolution image
            const Axis ax = axis_taps(nx, iw), ay = axis_taps(ny, ih);
            const unsigned r0 = __umul24(ay.i0, (unsigned)iw * 16u), r1 = __umul24(ay.i1, (unsigned)iw * 16u);
            const unsigned x0 = ax.i0 * 16u, x1 = ax.i1 * 16u;
            const f32x4 nw = *reinterpret_cast<const f32x4*>(at_byte(img, r0 + x0));
            const f32x4 ne = *reinterpret_cast<const f32x4*>(at_byte(img, r0 + x1));
            const f32x4 sw = *reinterpret_cast<const f32x4*>(at_byte(img, r1 + x0));
            const f32x4 se = *reinterpret_cast<const f32x4*>(at_byte(img, r1 + x1));
            const float wnw = ax.w0 * ay.w0, wne = ax.w1 * ay.w0, wsw = ax.w0 * ay.w1, wse = ax.w1 * ay.w1;
#pragma unroll
            for (int c = 0; c < 3; ++c) s.rgb[c] = fmaf(se[c], wse, fmaf(sw[c], wsw, fmaf(ne[c], wne, nw[c] * wnw)));
        }
        {   // features from the quarter-resolution map, same normalised coordinates
            const Axis ax = axis_taps(nx, fw), ay = axis_taps(ny, fh);
#pragma unroll
            for (int c = 0; c < 16; ++c) f[c] = 0.f;
            const unsigned r0 = __umul24(ay.i0, (unsigned)fw * 128u), r1 = __umul24(ay.i1, (unsigned)fw * 128u);
            const unsigned x0 = ax.i0 * 128u + (unsigned)half * 64u, x1 = ax.i1 * 128u + (unsigned)half * 64u;
            fma16(at_byte(fm, r0 + x0), ax.w0 * ay.w0, f);
            fma16(at_byte(fm, r0 + x1), ax.w1 * ay.w0, f);
            fma16(at_byte(fm, r1 + x0), ax.w0 * ay.w1, f);
            fma16(at_byte(fm, r1 + x1), ax.w1 * ay.w1, f);
        }
    }
    return s;
}

// ---------------------------------------------------------------------------------------------
// the fused kernel
// ---------------------------------------------------------------------------------------------
struct FrameK {           // GpnerfFrame by value (kernel argument; lands in SGPRs / scalar loads)
    const float* vol[GPNERF_LEVELS];
    int vol_dhw[GPNERF_LEVELS][3];
    const float* featmaps;
    int feat_h, feat_w;
    const float* imgs;
    int img_h, img_w;
    float proj[NV][12];
    float Rh[9], Th[3], bounds_min[3], voxel[3];
    float out_sh[3];      // as float
    const float* head_blob;
    const float* head_blob_split;   // f16 hi/lo image (GPNERF_FLAG_SPLIT_F16) or nullptr
    const float* head_blob_ref;     // reference-order image (FORM_F32) or nullptr
    const float* occ;     // level-1-sized occupancy volume or nullptr
    const float* vol_fold[GPNERF_LEVELS];    // folded volumes (gpnerf_fold_volumes) or nullptr: FORM_F32_FOLD
};

struct OutK {
    float *rgb, *depth, *acc, *disp, *weights, *z_vals, *rgb_in, *raw;
    uint8_t* ray_mask;
    int32_t* samples_done;
    unsigned* step_stats;     // optional [8]: wave-steps walked and which of them took which bit-exact exit (include/gpnerf_hip.h)
    const int32_t* order;     // optional: slot i of the launch renders ray order[i] (locality-friendly tiling)
};

struct KArgs {            // the fused kernel's only argument (see render_fused_kernel on why it is one struct)
    FrameK fr;
    const float* rays;
    long n_rays;
    int S;
    unsigned flags;
    float term_eps;
    OutK out;
    int split;
    float* part;
    int dynamic;              // 1: persistent workgroups pulling tiles from `queue`
    unsigned* queue;          // 8 counters (one per XCD), zero at launch
    // chained sample segments (early termination, render_fused_kernel<., true>): this launch walks samples
    // [seg * chain, (seg + 1) * chain) of the rays listed in `list_in` (nullptr: every ray, segment 0), 32 list entries per
    // wavefront, and appends the rays that are neither finished nor opaque to `list_out` for the next launch
    int chain, seg, wave_cap;
    int k_begin, k_end;       // chained form: the launch's sample range [k_begin, k_end) (segments need not be equally long, chain_schedule())
    int stagger;              // experiment: wavefronts start up to this many x 1.7 us late, scattered over the chip
    int seg_major;            // split > 1 on the tile queue: units ordered segment-major (1) or tile-major (0)
    int skip;                 // reference-order form: bit 0 = empty-space exit of the sigma feature layer, bit 1 = zero-density exit of the colour branch
    int chunk;                // tiles per chunk of the XCD queues (queue_tile())
    int tail_p;               // chain_plan(): samples per step of the units the last whole round is cut into
    const int* list_in;
    const unsigned* count_in;
    int* list_out;            // SPARSE: entry i of this launch's input writes its launch slot (or -1: finished / opaque) at [i]
    unsigned* count_out;      // (written by compact_list_kernel, not by the render kernel)
    unsigned* chunk_cnt;      // survivors per LIST_CHUNK input entries (one atomic per wavefront), zero at launch
    long p_cap;               // chain_plan(): 32 x the wavefronts the launch's remainder units may spread over
    long first_slot, first_items;   // segment 0 (no list) renders launch slots [first_slot, first_slot + first_items)
    // occupancy culling: bit k of cull_mask[slot * 2 + (k >> 6)] = launch slot `slot` keeps its sample of composite step k
    // (occupancy_mask_kernel, one pass over every sample before the launch); nullptr: the sample loop tests as it goes
    const unsigned long long* cull_mask;
    const int* tile_order;    // with cull_mask: the launch's tiles, most steps first (order_tiles_by_steps); nullptr: as they come
    // range guard of the split form: guard[0] = number of flagged tiles, guard[64 + tile] = 1 when an MFMA operand of the tile
    // reached the f16 range; the fix-up launch (FORM_F32_FIXUP) renders exactly the flagged tiles again in the fp32 form
    unsigned* guard;
    // frame-level deferral of the colour branch (render_fused_kernel<., ., ., true, true> + colour_units_kernel +
    // colour_accumulate_kernel): gd_ctrl = a zeroed 256-byte block (GD_COUNT: entries appended so far; GD_QUEUE..+7: the unit queue's
    // counters), gd_ent[n] = (launch slot, sample | rank << 8, weight bits, 0), gd_rgbw[slot * S + rank] = (r, g, b, weight),
    // gd_cnt[slot] = entries of the ray
    unsigned* gd_ctrl;
    unsigned* gd_flag;        // unified form: gd_flag[u] = 1 once unit u's 32 entries are written (zero at launch)
    int gd_waves;             // unified form: wavefronts of the launch (every one reports to GD_DONE when it has listed its last entry)
    int uni_budget;           // unified form: units a wavefront may evaluate between two tiles
    uint4* gd_ent;
    f32x4* gd_rgbw;
    int* gd_cnt;
};
constexpr int GD_QUEUE = 0, GD_COUNT = 8, GD_TICKET = 9, GD_DONE = 10;      // words of gd_ctrl's block, zero at launch
// (GD_COUNT counts ENTRIES in the two-launch form and 32-entry UNITS in the unified one, where GD_TICKET hands the units out)

// the forms of the fused kernel
constexpr int FORM_F32 = 0, FORM_SPLIT = 1, FORM_SPLIT_GUARD = 2, FORM_F32_FIXUP = 3, FORM_F32_FOLD = 4;
constexpr int GUARD_HEADER_WORDS = 64;

// bijective XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), give each XCD a
// contiguous run of tiles so neighbouring ray tiles hit the same L2 (speed only, never correctness)
DEV int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// The tile queue of XCD x holds the tiles of chunks x, x + 8, x + 16 ... (`chunk` consecutive tiles each): neighbouring tiles
// share an L2, and every XCD gets a sample of the whole frame (with one contiguous run per XCD, an XCD whose eighth of the
// image terminates late or culls little finishes last: +-14 % of work per eighth on the early-termination bench frame).
DEV long queue_len(long n_tiles, int chunk, int x) {
    const long nc = (n_tiles + chunk - 1) / chunk;               // chunks in all; only the last one may be short
    const long mine = (nc - x + 7) >> 3;
    const bool has_last = nc > 0 && ((nc - 1) & 7) == x;
    return mine * chunk - (has_last ? nc * chunk - n_tiles : 0);
}
DEV long queue_tile(int chunk, int x, unsigned t) { return ((long)(t / (unsigned)chunk) * 8 + x) * chunk + t % (unsigned)chunk; }

// torch.linspace(0,1,S)[k] as the CPU kernel evaluates it (one rounding per element; see oracle)
DEV float linspace01(int k, int S, float step) {
    return (k < (S >> 1)) ? step * (float)k : fmaf(-step, (float)(S - 1 - k), 1.f);
}

// get_sampling_points (BaseRender.py:37-38,48), jitter off: sample ks of a ray.  One function for every place that needs the point
// of a sample (the sample loop, the deferred colour passes), so that they contract to the same instructions and agree to the bit.
DEV void sample_point(float ox, float oy, float oz, float dx, float dy, float dz, float near, float far, int ks, int S, float step,
                      float& z, float& px, float& py, float& pz) {
    const float t = (S > 1) ? linspace01(ks, S, step) : 0.f;
    z = near * (1.f - t) + far * t;
    px = ox + dx * z; py = oy + dy * z; pz = oz + dz * z;
}

// pts_to_can_pts (BaseRender.py:52-60): (p - Th) @ Rh, then get_grid_coords (:62-73): voxel-normalised
// coordinate in [-1,1], dhw arithmetic, returned in xyz order.
template <class FR>
DEV void grid_coords(const FR& fr, float px, float py, float pz, float& gx, float& gy, float& gz) {
    const float qx0 = px - fr.Th[0], qy0 = py - fr.Th[1], qz0 = pz - fr.Th[2];
    // torch.matmul([n,3], Rh) (:57): the same sgemm FMA chain over k = 0..2 -- the grid coordinates are then the reference's bits
    const float qx = fmaf(qz0, fr.Rh[6], fmaf(qy0, fr.Rh[3], qx0 * fr.Rh[0]));
    const float qy = fmaf(qz0, fr.Rh[7], fmaf(qy0, fr.Rh[4], qx0 * fr.Rh[1]));
    const float qz = fmaf(qz0, fr.Rh[8], fmaf(qy0, fr.Rh[5], qx0 * fr.Rh[2]));
    gx = ((qx - fr.bounds_min[0]) / fr.voxel[2]) / fr.out_sh[2] * 2.f - 1.f;
    gy = ((qy - fr.bounds_min[1]) / fr.voxel[1]) / fr.out_sh[1] * 2.f - 1.f;
    gz = ((qz - fr.bounds_min[2]) / fr.voxel[0]) / fr.out_sh[0] * 2.f - 1.f;
}

// A lane owns a ray, so an [N,S] output written sample by sample is one 4-byte store per lane at a stride of S floats, each
// its own 64-byte write request (measured: 2 GB of requests per 512x512x64 frame for the 134 MB of weights + z_vals).
// z_vals is a function of (near, far, k) alone, so it is written after the sample loop instead, the whole wave walking one
// ray's row: z_vals[ray][k] for k in [k_from, k_to), 64 consecutive samples per store instruction.
// lane_stride: ray rr of the wavefront sits in lane rr * lane_stride (render_tile's P)
DEV void write_z_vals(float* __restrict__ z_vals, const int lane, const int ray_lo, const float near, const float far,
                      const int n_active, const int S, const float step, const int k_from, const int k_to, const int lane_stride = 1) {
    for (int rr = 0; rr < n_active; ++rr) {
        const int rid = __builtin_amdgcn_readlane(ray_lo, rr * lane_stride);
        const float nr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, near), rr * lane_stride));
        const float fr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, far), rr * lane_stride));
        for (int kk = k_from + lane; kk < k_to; kk += 64) {
            const float tk = (S > 1) ? linspace01(kk, S, step) : 0.f;
            z_vals[(size_t)rid * S + kk] = nr * (1.f - tk) + fr * tk;
        }
    }
}

// Device-scope (sc1) accesses for the counters waves of different CUs share within a launch: they are performed at the L2 /
// the memory side rather than in the CU's vector L1, which is neither invalidated nor shared between CUs.
DEV unsigned agent_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane performs the access, the wave gets the value
DEV unsigned wave_load(const unsigned* p, int lane) {
    unsigned v = 0;
    if (lane == 0) v = agent_load(p);
    return __builtin_amdgcn_readfirstlane(v);
}
DEV unsigned wave_add(unsigned* p, unsigned d, int lane) {
    unsigned v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(p, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(v);
}

// Launched with 1..8 waves per workgroup (blockDim.x = 64 * waves): one workgroup per CU either way (LDS), so the
// host picks the width that balances the grid over the CUs (choose_geometry()).
#ifndef GPNERF_MAX_WAVES
#define GPNERF_MAX_WAVES 8
#endif
constexpr int DEFER_QUEUE = 64;                 // entries of a wavefront's queue of samples waiting for their colour pass (render_tile)
constexpr int TALLY_WORDS = 8;                  // per wavefront, behind the queues: the tile's step_stats (diagnostic launches)
// a tile's tally in LDS: steps, steps without the sigma feature layer, evaluations of the colour branch (in the step, or a deferred
// pass), steps settled behind the sample loop (exactly opaque rays), volume levels of the sigma feature layer left out
enum { T_STEPS = 0, T_EMPTY = 1, T_PASS = 2, T_OPAQUE = 3, T_LEVELS = 4 };
constexpr size_t DEFER_LDS_BYTES = (size_t)GPNERF_MAX_WAVES * (DEFER_QUEUE * 8 + TALLY_WORDS * 4);      // behind the head image
constexpr int LIST_CHUNK_SHIFT = 11, LIST_CHUNK = 1 << LIST_CHUNK_SHIFT;      // entries per survivor counter / per compaction workgroup
// number of launch slots a chained launch renders: the previous segment's survivors, or every ray (segment 0)
typedef const __attribute__((address_space(4))) KArgs* kargs_cptr;
DEV long chain_items(kargs_cptr k) { return k->list_in ? (long)*k->count_in : k->first_items; }

// Work units of a chained launch over n list entries on `slots` wavefronts: whole rounds of 32-entry tiles at one sample per step
// (throughput-bound), then the remainder -- the tiles of the last, partial round, or all of a level with few rays -- with as many
// samples of a ray side by side (render_tile's P) as still leave every wavefront a unit: what is left when the queue runs dry
// is bound by the latency of a tile's dependent steps, not by throughput.  Under occupancy culling: one sample per step.
struct ChainPlan { long bulk_tiles, rem_tiles; int rem_p; };
DEV ChainPlan chain_plan(kargs_cptr k) {
    const long n = chain_items(k), n32 = (n + RAYS_PER_WAVE - 1) / RAYS_PER_WAVE;
    ChainPlan c;
    c.bulk_tiles = n32; c.rem_tiles = 0; c.rem_p = 1;
    if ((k->flags & GPNERF_FLAG_OCC_CULL) || k->p_cap <= 0) return c;
    const long slots = k->p_cap / RAYS_PER_WAVE;
    long rounds = n32 / slots;
    // The last whole round of a launch of two rounds or more also runs as units of tail_p samples per step: a SIMD serves its older
    // wavefront first (it walks 5 units of 8, the younger 3), and with whole 16-step units the older ones walk their last alone
    // (tools/wave_times.py); shorter units let both finish together (512x512x128 bench frame, round 4: 9.02 -> 8.82 ms with 4 samples per
    // step; 2: 8.91, 8: 9.2.  Round 6, the colour work out of these launches: 2 samples per step 6.93 -> 6.87 ms, the default since)
    if (k->tail_p > 1 && rounds >= 2) --rounds;
    c.bulk_tiles = rounds * slots;
    const long rem = n - c.bulk_tiles * RAYS_PER_WAVE;
    if (rem <= 0) { c.bulk_tiles = n32; return c; }
    int p = 1;
    while (p < 8 && rem * (2 * p) <= k->p_cap) p *= 2;
    if (k->tail_p > p) p = k->tail_p;
    c.rem_p = p;
    c.rem_tiles = (rem * p + RAYS_PER_WAVE - 1) / RAYS_PER_WAVE;
    return c;
}

// Between two chained launches: the sparse list of the launch just finished (launch slot or -1 per input entry) becomes the dense
// input list of the next, survivors in input order -- which is the patch-major order the frame was launched in, so the 32 rays
// of a wavefront and the wavefronts of a CU stay neighbours in the image at every level.  One workgroup per LIST_CHUNK entries:
// its offset is the sum of the chunk counters before it (the render kernel's wavefronts filled them), its own entries are ranked
// with ballots.  The workgroup that holds the last entry writes the list's length.
__global__ void __launch_bounds__(256) compact_list_kernel(const int* __restrict__ sparse, const unsigned* __restrict__ chunk_cnt,
                                                           const unsigned* __restrict__ count_in, const long first_items,
                                                           int* __restrict__ dense, unsigned* __restrict__ count_out) {
    const long n_items = count_in ? (long)*count_in : first_items;
    const long start = (long)blockIdx.x * LIST_CHUNK;
    if (start >= n_items) return;
    __shared__ unsigned red[256 / 64];
    __shared__ unsigned wave_base[256 / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned part = 0;
    for (unsigned i = threadIdx.x; i < blockIdx.x; i += 256) part += chunk_cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) red[wave] = part;
    // this thread's 8 consecutive entries
    const long e0 = start + (long)threadIdx.x * 8;
    int v[8];
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = (e0 + j < n_items) ? sparse[e0 + j] : -1;
        mine += v[j] >= 0;
    }
    unsigned incl = mine;                      // inclusive scan over the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_base[wave] = incl;
    __syncthreads();
    unsigned offset = red[0] + red[1] + red[2] + red[3];
    unsigned before = 0;
    for (int w = 0; w < wave; ++w) before += wave_base[w];
    unsigned pos = offset + before + incl - mine;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (v[j] >= 0) dense[pos++] = v[j];
    if (start + LIST_CHUNK >= n_items && threadIdx.x == 255) *count_out = offset + before + incl;
}

// One work unit = (32-ray tile, sample segment) rendered by one wavefront: with split > 1 the samples of a tile are divided
// between `split` waves, whose partial composites are merged by combine_segments_kernel (finer load balance for small frames).
// Returns true when the tile goes on in a later work item (chained segments only).
// P (chained form only): samples of one ray a wavefront evaluates per step.  A level with few rays left is bound by the latency
// of its 16 dependent steps (~30 us each however few waves a CU runs), not by throughput: with P = 2, 4, 8 a wavefront renders
// 32 / P rays, P consecutive samples of each side by side in P neighbouring lanes, and walks a segment in 16 / P steps.  Every
// lane of a ray's group composites the group's P samples in order from values fetched across the lanes, so the ray's state is
// replicated in the group and the arithmetic per ray -- and with it every output bit -- is the same as with P = 1.
// Projector.compute (:326-363) for the NV views of one sample, in the slot order the form's first colour / density layers take
// (reference-order form: gpr::ref35, (r, g) (b, 0) (f0, f1) ... (f30, f31)); returns the number of views that see the sample
template <int FORM>
DEV float gather_views(const __attribute__((address_space(4))) FrameK& fr, float px, float py, float pz, bool neg, int half,
                       float (&x)[NV][18], float (&vrgb)[NV][3]) {
    float nvalid = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const ViewSample s = gather_view<FORM == FORM_F32_FOLD || FORM == FORM_F32>(fr.proj[v], fr.imgs + (size_t)v * fr.img_h * fr.img_w * 4, fr.img_h, fr.img_w,
                                         fr.featmaps + (size_t)v * fr.feat_h * fr.feat_w * 32, fr.feat_h, fr.feat_w,
                                         px, py, pz, neg, half, x[v]
                                         );
        if constexpr (FORM == FORM_F32) {
            float fk[16];
            interleave16(x[v], fk);
            x[v][0] = half ? s.rgb[1] : s.rgb[0];
            x[v][1] = half ? 0.f : s.rgb[2];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[v][2 + c] = fk[c];
        } else {
            x[v][16] = half ? s.rgb[1] : s.rgb[0];
            x[v][17] = half ? 0.f : s.rgb[2];
        }
        vrgb[v][0] = s.rgb[0]; vrgb[v][1] = s.rgb[1]; vrgb[v][2] = s.rgb[2];
        nvalid += s.valid;
    }
    return nvalid;
}

DEV void agent_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int FORM, bool CHAIN, int P = 1, bool CULL = false, bool DEFER = false, bool GDEF = false, bool UNI = false>
DEV bool render_tile(float* lds, const int lane, const long tile, const int seg, const long entry_base = 0) {
    static_assert(!UNI || GDEF, "unified form: a listing sample loop");
    static_assert(!CULL || (!CHAIN && P == 1), "occupancy culling: plain form only");
    static_assert(!GDEF || (DEFER && !CULL), "frame-level deferral: a deferred sample loop without culling");
    static_assert(P == 1 || CHAIN, "several samples per step: chained form only");
    constexpr int RAYS = RAYS_PER_WAVE / P;             // rays per wavefront
    constexpr bool SPLIT = FORM == FORM_SPLIT || FORM == FORM_SPLIT_GUARD;
    // tag: the split form's helpers check their operands' range (FORM_SPLIT_GUARD) or do not
    typename std::conditional<FORM == FORM_SPLIT_GUARD, Guard, NoGuard>::type gmax{};
    // Everything the sample loop reads from the arguments is re-read from the kernarg segment (scalar loads, scalar
    // cache) at the top of every iteration through `kp`, a pointer the optimiser cannot see through.  Held in SGPRs
    // across the loop instead, the ~130 argument dwords spill to VGPR lanes and come back as v_readlane_b32 -- VALU
    // instructions in the middle of the MFMA chains, which stall the matrix pipe (tools/micro/mfma_chains.hip).
    // The same goes for the per-tile values: a persistent wave renders many tiles, and arguments kept in SGPRs across the
    // tile loop spill just the same, so every tile starts from a fresh, laundered kernarg pointer.
    typedef const __attribute__((address_space(4))) KArgs* kargs_ptr;
    kargs_ptr k0 = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k0));
    const long n_rays = k0->n_rays;
    const int S = k0->S;
    const unsigned flags = k0->flags;
    const float term_eps = k0->term_eps;
    const int split = k0->split;
    const unsigned* const lw = reinterpret_cast<const unsigned*>(lds);

    Stamps st;
    st.start();
    const int n = lane & 31, half = lane >> 5;
    const int sub = n & (P - 1), rn = n / P;            // which of the step's P samples this lane evaluates, which ray of the wavefront
    const long ray0 = entry_base + tile * RAYS;      // (entry_base: where the launch's remainder units start in the ray list)
    // the chained form renders the launch slots listed in list_in (the rays the previous segment left alive, in no particular
    // order), 32 / P list entries per wavefront; everything else renders 32 consecutive slots
    const long n_items = CHAIN ? chain_items(k0) : n_rays;
    if (ray0 >= n_items) return false;
    const bool active = (ray0 + rn) < n_items;
    long slot = active ? ray0 + rn : n_items - 1;
    if constexpr (CHAIN) slot = k0->list_in ? (long)k0->list_in[slot] : k0->first_slot + slot;
    const int ray = k0->out.order ? k0->out.order[slot] : (int)slot;
    const bool neg = (flags & GPNERF_FLAG_NEG_RAY) != 0;             // Projector front test
    const bool flip = (flags & GPNERF_FLAG_FLIP_SAMPLES) != 0;       // raw2outputs(neg=True)
    const bool early = (flags & GPNERF_FLAG_EARLY_TERM) != 0;
    // occupancy culling: CULL = the form that walks keep bits computed before the launch (its own instantiation); without them
    // (outputs that need every step written, no workspace) the plain instantiation tests sample by sample
    const bool cull = !CULL && (flags & GPNERF_FLAG_OCC_CULL) != 0 && k0->fr.occ != nullptr;

    const f32x4 r0 = *reinterpret_cast<const f32x4*>(k0->rays + (size_t)ray * 8);
    const f32x4 r1 = *reinterpret_cast<const f32x4*>(k0->rays + (size_t)ray * 8 + 4);
    const float ox = r0[0], oy = r0[1], oz = r0[2], dx = r0[3], dy = r1[0], dz = r1[1], near = r1[2], far = r1[3];

    float T = 1.f, c_r = 0.f, c_g = 0.f, c_b = 0.f, depth = 0.f, acc = 0.f;
    int n_q = 0;                                // (frame-level deferral: entries this ray has listed = the rank of its next one)
    float rin[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) rin[i] = 0.f;
    int n_two = 0, n_done = 0;
    const float step = (S > 1) ? 1.f / (float)(S - 1) : 0.f;
    const bool writer = active && (half == 0) && (sub == 0);

    // the chained form is its own instantiation: the plain sample loop stays as it was
    const int k_end = CHAIN ? min(k0->k_end, S) : (int)(((long)S * (seg + 1)) / split);
    int k = CHAIN ? k0->k_begin : (int)(((long)S * seg) / split);
    const int k_begin = k;
    if (CHAIN && k > 0) {       // resume: what the previous segment of this ray left behind (the 16 floats of a split segment)
        const f32x4* p = reinterpret_cast<const f32x4*>(k0->part + (size_t)slot * 16);
        const f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
        c_r = a[0]; c_g = a[1]; c_b = a[2]; depth = a[3];
        if constexpr (GDEF) { n_q = __builtin_bit_cast(int, a[0]); c_r = 0.f; }      // (the colour map is not the sample loop's: its slot carries the rank)
        acc = b[0]; T = b[1]; rin[0] = b[3];
        const int packed = (int)b[2];
        n_two = packed & 4095; n_done = packed >> 12;
        rin[1] = c[0]; rin[2] = c[1]; rin[3] = c[2]; rin[4] = c[3];
        rin[5] = d[0]; rin[6] = d[1]; rin[7] = d[2]; rin[8] = d[3];
    }
    // occupancy culling with the keep bits computed beforehand (occupancy_mask_kernel): the wavefront walks only the steps at
    // which one of its rays keeps its sample; a skipped step costs nothing (tested as it goes, it cost ~12 % of a full step:
    // sample position, grid coordinates, eight occupancy taps, a ballot -- 3.4 ms for a frame that evaluates 14.5 % of its steps)
    unsigned long long my_keep[2] = {0ull, 0ull}, any_keep[2] = {0ull, 0ull};
    constexpr bool masked = CULL;
    if (masked) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            my_keep[w] = active ? k0->cull_mask[(size_t)slot * 2 + w] : 0ull;
            unsigned lo = (unsigned)my_keep[w], hi = (unsigned)(my_keep[w] >> 32);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { lo |= (unsigned)__shfl_xor((int)lo, o); hi |= (unsigned)__shfl_xor((int)hi, o); }
            any_keep[w] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
        }
    }
    STAMP(st, 7);
    // Deferred colour branch (reference-order form, plain sample loop; kp->skip bit 1, never with a `raw` output).  The colour branch
    // reads nothing the density branch computes -- only the three views' 35-vectors -- and a sample whose weight alpha * T is exactly
    // zero adds fma(0, rgb, c) = c to the colour map: its colour branch (434 of a step's 748 MFMAs) need not run.  Zero-density
    // samples are most of any frame (a trained model's empty space; 71 % of the synthetic bench frame, nn.ReLU on the density), but a
    // step's 32 rays are rarely ALL empty, so the exit is taken sample by sample: each step runs the density branch and composites
    // everything but the colour; rays with a non-zero weight append (ray lane, sample, weight) to a 64-entry queue of the wavefront in
    // LDS; when 32 are waiting, one colour pass evaluates them -- lane i regathers item i's views (the same loads, minutes-old in L2)
    // -- and every ray takes its own results back in sample order, so c accumulates in exactly the order of the plain loop.  Same bits.
    static_assert(!DEFER || P == 1 || GDEF, "colour passes of the wavefront: one sample per step");
    constexpr bool CAN_DEFER = DEFER, defer = DEFER;
    unsigned long long mine = 0ull;             // bit j: the queue entry j places behind the head is one of this ray's
    int q_head = 0, q_cnt = 0;                  // (uniform)
    uint2* dq = nullptr;
    // behind the head image (split form: and behind the range guard's slots): the wavefronts' queues, then their tallies
    constexpr int QUEUE_AT = SPLIT ? gph::BLOB_WORDS + (FORM == FORM_SPLIT_GUARD ? 2 * GUARD_LDS_SLOTS : 0) : gpl::BLOB_FLOATS;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (CAN_DEFER) dq = reinterpret_cast<uint2*>(lds + QUEUE_AT) + wave_in_wg * DEFER_QUEUE;
    // step_stats (the diagnostic launch only): the tile counts in LDS (T_*) and adds them to the launch's counters once, at its
    // end (one global atomic per step and counter made the counting launch 40 % slower than the launches it describes)
    unsigned* const tl = reinterpret_cast<unsigned*>(lds) + QUEUE_AT + GPNERF_MAX_WAVES * DEFER_QUEUE * 2 + wave_in_wg * TALLY_WORDS;
    const bool tally = k0->out.step_stats != nullptr;
    if (tally && lane < TALLY_WORDS) tl[lane] = 0u;
    int k_lim = k_end;                          // (early termination of the tile as a whole moves it to where the loop stopped)
    int opaque_from = -1;                       // (first step behind the one at which every ray's transmittance was exactly 0)
    for (;; k += P) {
        if constexpr (GDEF) {
            // Frame-level deferral: the samples waiting for their colour branch leave the wavefront altogether -- 32 at a time (what
            // is left at the tile's end) they are appended to the launch's entry list, which is evaluated 32 entries per wavefront
            // step, whichever tiles they came from (by this launch's own wavefronts once they have no tile left: UNI; by
            // colour_units_kernel otherwise); colour_accumulate_kernel then adds every ray's terms in sample order.  The same arithmetic
            // on the same operands in the same order as the passes below: the same bits.
            if (q_cnt >= 32 || (q_cnt > 0 && !(k < k_lim))) {
                const int nb = min(q_cnt, 32);
                const uint2 e = dq[(q_head + (n < nb ? n : 0)) & (DEFER_QUEUE - 1)];
                const int sl = __shfl((int)slot, (int)(e.x & 31u));
                kargs_ptr kb = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kb));
                if constexpr (UNI) {
                    // unified form: whole units (what is left at the tile's end padded with null entries), written through to the
                    // device's coherence point and acknowledged (vmcnt) before the unit's flag says so -- wavefronts of other CUs
                    // evaluate it while this launch is still running (see render_fused_kernel)
                    const unsigned u = wave_add(kb->gd_ctrl + GD_COUNT, 1u, lane);
                    if (lane < 32) {
                        unsigned* const pe = reinterpret_cast<unsigned*>(kb->gd_ent + (size_t)u * 32 + lane);
                        agent_store(pe, lane < nb ? (unsigned)sl : 0xffffffffu);
                        agent_store(pe + 1, e.x >> 5);
                        agent_store(pe + 2, e.y);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) agent_store(kb->gd_flag + u, 1u);
                } else {
                const unsigned base = wave_add(kb->gd_ctrl + GD_COUNT, (unsigned)nb, lane);
                if (lane < nb) kb->gd_ent[(size_t)base + lane] = uint4{(unsigned)sl, e.x >> 5, e.y, 0u};
                }
                q_head = (q_head + nb) & (DEFER_QUEUE - 1);
                q_cnt -= nb;
                k -= P;
                continue;
            }
        } else
        if constexpr (CAN_DEFER) {
            if (defer && (q_cnt >= 32 || (q_cnt > 0 && !(k < k_lim)))) {
                const int nb = min(q_cnt, 32);
                const uint2 e = dq[(q_head + (n < nb ? n : 0)) & (DEFER_QUEUE - 1)];        // lanes beyond the last entry redo entry 0
                const int r = (int)(e.x & 31u), kk = (int)(e.x >> 5);
                const float wq = __builtin_bit_cast(float, e.y);
                kargs_ptr kb = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kb));
                float zq, qx_, qy_, qz_;
                sample_point(__shfl(ox, r), __shfl(oy, r), __shfl(oz, r), __shfl(dx, r), __shfl(dy, r), __shfl(dz, r), __shfl(near, r), __shfl(far, r),
                             flip ? (S - 1 - kk) : kk, S, step, zq, qx_, qy_, qz_);
                float xq[NV][18], vq[NV][3], cq[3];
                gather_views<FORM>(kb->fr, qx_, qy_, qz_, neg, half, xq, vq);
                STAMP(st, 14);
                if constexpr (SPLIT) {
                    Frag mvf[6];
                    mean_var_s(gmax, xq, mvf);
                    mlp_colour_s(gmax, lw, lane, xq, mvf, cq, st);
                } else {
                    float mvq[36];
                    if constexpr (FORM == FORM_F32) { mean_var_ref(xq, mvq); mlp_colour_ref(lds, lane, xq, mvq, cq, st); }
                    else { mean_var(xq, mvq); mlp_colour(lds, lane, xq, mvq, cq, st); }
                }
                if (tally && lane == 0) tl[T_PASS] += 1u;
                unsigned todo = (unsigned)mine & (nb >= 32 ? ~0u : ((1u << nb) - 1u));
                while (__any(todo != 0u)) {             // a ray's entries of this pass, oldest first
                    const int j = todo ? __builtin_ctz(todo) : 0;
                    const float wj = __shfl(wq, j), r0 = __shfl(cq[0], j), r1 = __shfl(cq[1], j), r2 = __shfl(cq[2], j);
                    if (todo) { c_r = fmaf(wj, r0, c_r); c_g = fmaf(wj, r1, c_g); c_b = fmaf(wj, r2, c_b); }
                    todo &= todo - 1u;
                }
                STAMP(st, 15);
                mine >>= nb;
                q_head = (q_head + nb) & (DEFER_QUEUE - 1);
                q_cnt -= nb;
                k -= P;                                 // (no sample step this time round)
                continue;
            }
        }
        if (!(k < k_lim)) break;
        if (masked) {                           // the next step >= k at which the tile has anything to do
            const unsigned long long w0 = k < 64 ? (any_keep[0] >> k) << k : 0ull;
            const unsigned long long w1 = k < 64 ? any_keep[1] : (k < 128 ? (any_keep[1] >> (k - 64)) << (k - 64) : 0ull);
            k = w0 ? __builtin_ctzll(w0) : (w1 ? 64 + __builtin_ctzll(w1) : k_end);
            if (k >= k_end) {
                if (CAN_DEFER && q_cnt > 0) { k_lim = k; k -= P; continue; }       // (the colour passes still waiting, then out)
                break;
            }
        }
        kargs_ptr kp = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const __attribute__((address_space(4))) FrameK& fr = kp->fr;
        const __attribute__((address_space(4))) OutK& out = kp->out;
        // this lane's sample of the step (P > 1: the group's lanes take consecutive samples; one past the segment's end idles)
        const bool in_seg = P == 1 || k + sub < k_end;
        const int kl = P == 1 ? k : min(k + sub, k_end - 1);
        // raw2outputs(neg=True) flips rgb and sigma along the ray but not z (BaseRender.py:86-88,101):
        // composite step k consumes the network output of sample S-1-k.
        const int ks = flip ? (S - 1 - kl) : kl;
        // get_sampling_points (BaseRender.py:37-38,48), jitter off
        float z, px, py, pz;
        sample_point(ox, oy, oz, dx, dy, dz, near, far, ks, S, step, z, px, py, pz);

        float gx, gy, gz;
        grid_coords(fr, px, py, pz, gx, gy, gz);

        // progressive culling (demo_render.py:270-283): evaluate only samples whose occupancy interpolates to > 0;
        // a tile whose 32 samples are all culled skips its gathers and the MLP (alpha = 0 for all of them)
        bool keep = true;
        if constexpr (masked) {
            keep = ((k < 64 ? my_keep[0] >> k : my_keep[1] >> (k - 64)) & 1ull) != 0ull;
        } else if (cull) {
            keep = sample_occupancy(fr.occ, fr.vol_dhw[0][0], fr.vol_dhw[0][1], fr.vol_dhw[0][2], gx, gy, gz) > 0.f;
            if (!__any(keep)) {
                if (writer) {
                    if (out.weights) out.weights[(size_t)ray * S + kl] = 0.f;
                    if (out.raw) *reinterpret_cast<f32x4*>(out.raw + ((size_t)ray * S + ks) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                T = T * (1.f + 1e-10f);                 // cumprod(1 - alpha + 1e-10) with alpha = 0
                continue;
            }
        }

        // chained form: a ray that is opaque stops HERE, whatever the other rays of its wavefront do (its result is a function
        // of the ray alone, so it does not matter which rays are packed together); the plain form stops a tile as a whole
        const bool dead = CHAIN && P == 1 && T < term_eps;     // (P > 1: decided sample by sample in the composite below)
        if constexpr (P == 1) n_done += dead ? 0 : 1;
        float sf[32];
        Frag sff[4];
        float x[NV][18];                        // Projector.compute (:326-363)
        float vrgb[NV][3];
        float nvalid;
        if constexpr (FORM == FORM_F32_FOLD) {
            // the sigma feature layer's pre-activation: levels FOLD_FROM.. interpolated from the folded volumes (see gather_folded),
            // the finer levels' features through the layer's first 16 FOLD_FROM k-steps as before
            int hb = half;
            asm volatile("" : "+v"(hb));                  // the bias reads stay in the loop (hoisted, they hold 32 registers across it)
            f32x16 g0 = bias_tile<gpl::GEO>(lds, 0, hb), g1 = bias_tile<gpl::GEO>(lds, 1, hb);
            float fu[FOLD_FROM > 0 ? 16 * FOLD_FROM : 1];
#pragma unroll
            for (int l = 0; l < FOLD_FROM; ++l)
                gather_volume(fr.vol[l], fr.vol_dhw[l][0], fr.vol_dhw[l][1], fr.vol_dhw[l][2], gx, gy, gz, half, fu + 16 * l);
            if constexpr (FOLD_FROM > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = FOLD_FROM; l < GPNERF_LEVELS; ++l) {
                gather_folded(fr.vol_fold[l], fr.vol_dhw[l][0], fr.vol_dhw[l][1], fr.vol_dhw[l][2], gx, gy, gz, half, g0, g1);
            }
            STAMP(st, 0);
            if constexpr (FOLD_FROM > 0) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                mfma_tile<16 * FOLD_FROM>(wtile<gpl::GEO>(lds, 0), ln, fu, g0);
                mfma_tile<16 * FOLD_FROM>(wtile<gpl::GEO>(lds, 1), ln, fu, g1);
            }
            elus_n<16>(g0, sf);
            elus_n<16>(g1, sf + 16);
        } else if constexpr (FORM == FORM_F32 && DEFER) {
            // SparseConvNet.forward sampling (:113-122) and the sigma feature layer, level by level: level l's 16 k-steps (32 MFMAs,
            // ~2 000 cycles: about one L2 round trip) run while level l + 1's 32 loads are in flight, so three of the step's four
            // volume round trips hide behind the wavefront's OWN matrix work -- with the colour branch out of the step the other
            // wavefront of the SIMD no longer covers them.  The chain is the layer's: k ascending, level-major,
            // bias last.  A level whose 16 features are zero in all 32 samples adds fma(w, 0, s) = s: its MFMAs are left out (the
            // empty-space exit, level by level; step_stats counts the steps that skip all four).
            VolTaps taps;
            vol_issue(fr.vol[0], fr.vol_dhw[0][0], fr.vol_dhw[0][1], fr.vol_dhw[0][2], gx, gy, gz, half, taps);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 g0 = zero_tile(), g1 = zero_tile();
            int levels_skipped = 0;
#pragma unroll
            for (int l = 0; l < GPNERF_LEVELS; ++l) {
                float fl[16];
                vol_finish(taps, fl);
                __builtin_amdgcn_sched_barrier(0);
                if (l + 1 < GPNERF_LEVELS) {
                    vol_issue(fr.vol[l + 1], fr.vol_dhw[l + 1][0], fr.vol_dhw[l + 1][1], fr.vol_dhw[l + 1][2], gx, gy, gz, half, taps);
                    __builtin_amdgcn_sched_barrier(0);
                }
                unsigned bits = 0u;
#pragma unroll
                for (int i = 0; i < 16; i += 2) bits |= __builtin_bit_cast(unsigned, fl[i]) | __builtin_bit_cast(unsigned, fl[i + 1]);
                if ((kp->skip & 1) && __all((bits << 1) == 0u)) { ++levels_skipped; }
                else {
                    float fk[16];
                    interleave16(fl, fk);
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    g0 = mfma_tile_from<16>(wtile<gpl::GEO>(lds, 0) + l * 16 * 64, ln, fk, g0);
                    g1 = mfma_tile_from<16>(wtile<gpl::GEO>(lds, 1) + l * 16 * 64, ln, fk, g1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tally && lane == 0) { tl[T_LEVELS] += (unsigned)levels_skipped; if (levels_skipped == GPNERF_LEVELS) tl[T_EMPTY] += 1u; }
            STAMP(st, (k == k_begin ? 10 : (k == k_begin + P ? 11 : 0)));
            {
                int hb = half;
                asm volatile("" : "+v"(hb));
                f32x4 b0[4], b1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    b0[q] = *reinterpret_cast<const f32x4*>(bias_ptr<gpl::GEO>(lds, 0, hb) + 4 * q);
                    b1[q] = *reinterpret_cast<const f32x4*>(bias_ptr<gpl::GEO>(lds, 1, hb) + 4 * q);
                }
                elur_n<16>(g0, b0, sf);
                elur_n<16>(g1, b1, sf + 16);
            }
        } else {
            // SparseConvNet.forward sampling (:113-122): 4 levels, level-major concat
            float fv[64];
#pragma unroll
            for (int l = 0; l < GPNERF_LEVELS; ++l)
            {
                if constexpr (DEFER) {
                    gather_volume_batched<!SPLIT>(fr.vol[l], fr.vol_dhw[l][0], fr.vol_dhw[l][1], fr.vol_dhw[l][2], gx, gy, gz, half, fv + 16 * l);
                    continue;
                }
                gather_volume<!SPLIT>(fr.vol[l], fr.vol_dhw[l][0], fr.vol_dhw[l][1], fr.vol_dhw[l][2], gx, gy, gz, half, fv + 16 * l);
                if (l & 1) __builtin_amdgcn_sched_barrier(0);    // at most two levels' 64 loads in flight (register pressure)

            }
            STAMP(st, (k == k_begin ? 10 : (k == k_begin + P ? 11 : 0)));
            if constexpr (SPLIT) geo_eval_s(gmax, lw, lane, fv, sff);
            else {
                // reference order: k-step t of the layer = channels (2i, 2i + 1) of level t >> 4
                // Empty space: where none of the 32 samples of the step touches an active voxel of any level, all 128 volume features
                // are exactly zero, the layer's chain is fma(w, 0, s) from s = 0, i.e. 0, and its output is ELU(bias) -- the SAME
                // bits without the 128 MFMAs.  A real frame's pyramid is ReLU-sparse and mostly empty (body-like frame: 2 % of
                // level 0's voxels are sites), so most steps of most wavefronts take this exit; the dense synthetic bench frame
                // never does and pays the test (32 v_or3 + a ballot).  (kp->skip: off under GPNERF_FLAG_NO_EXITS.)
                unsigned bits = 0u;
#pragma unroll
                for (int i = 0; i < 64; i += 2) bits |= __builtin_bit_cast(unsigned, fv[i]) | __builtin_bit_cast(unsigned, fv[i + 1]);
                if ((kp->skip & 1) && __all((bits << 1) == 0u)) {          // (-0.0 counts as zero: fma(w, -0, +0) = +0 too)
                    if (tally && lane == 0) { tl[T_EMPTY] += 1u; tl[T_LEVELS] += (unsigned)GPNERF_LEVELS; }
                    geo_bias_ref(lds, lane, sf);
                } else {
                    float fk[64];
#pragma unroll
                    for (int l = 0; l < GPNERF_LEVELS; ++l) interleave16(fv + 16 * l, fk + 16 * l);
                    geo_eval_ref(lds, lane, fk, sf);
                }
            }
        }
        STAMP(st, 1);

        // Projector.compute (:326-363)
        // (deferred form: every gather of the step ahead of its matrix work -- one memory phase, one compute phase -- was tried:
        //  105 spilled registers, 10.35 -> 12.39 ms)
        nvalid = gather_views<FORM>(fr, px, py, pz, neg, half, x, vrgb);
        const bool two_views = nvalid > 1.f && keep;    // pixel_mask (:139); culled samples never count
        if constexpr (P == 1) { if (two_views && !dead) ++n_two; }

        STAMP(st, 2);
        float sigma, rgb[3];
        // (the zero-density exit in the reference-order form only: built into the folded and split forms too, the early return
        // cost their sample loops their register allocation -- 13.3 -> 15.1 ms and 7.5 -> 10.3 ms on the bench frame)
        if constexpr (DEFER) {
            if constexpr (SPLIT) {
                Frag mvf[6];
                mean_var_s(gmax, x, mvf);
                mlp_density_s(gmax, lw, lane, sff, mvf, nvalid, sigma);
            } else {
                float mv[36];
                if constexpr (FORM == FORM_F32) { mean_var_ref(x, mv); mlp_density_ref(lds, lane, sf, mv, nvalid, sigma); }
                else { mean_var(x, mv); mlp_density(lds, lane, sf, mv, nvalid, sigma); }
            }
            rgb[0] = 0.f; rgb[1] = 0.f; rgb[2] = 0.f;               // (the colour map's terms arrive with the colour passes)
        }
        else if constexpr (SPLIT) { mlp_eval_s(gmax, lw, lane, sff, x, nvalid, sigma, rgb, st); if (tally && lane == 0) tl[T_PASS] += 1u; }
        else if constexpr (FORM == FORM_F32) {
            {
                const bool may_skip = (kp->skip & 2) && !out.raw;
                mlp_eval_ref(lds, lane, sf, x, nvalid, sigma, rgb, st, may_skip);
                if (tally) {                                            // (the diagnostic launch only)
                    const bool all_zero = may_skip && __all(sigma == 0.f);
                    if (!all_zero && lane == 0) tl[T_PASS] += 1u;
                }
            }
        }
        else { mlp_eval(lds, lane, sf, x, nvalid, sigma, rgb, st); if (tally && lane == 0) tl[T_PASS] += 1u; }
        if (tally && lane == 0) tl[T_STEPS] += 1u;
        if (CULL || cull) {
            if (!keep) sigma = 0.f;                     // hold_alpha stays 0 for culled samples (demo_render.py:337-341)
            if (!(1.f - fast_exp(-sigma) > 1e-14f)) { rgb[0] = 0.f; rgb[1] = 0.f; rgb[2] = 0.f; }   // valid1 (:317)
        }

        if (out.raw && active && half == 0 && in_seg) {
            f32x4 rw; rw[0] = rgb[0]; rw[1] = rgb[1]; rw[2] = rgb[2]; rw[3] = sigma;
            *reinterpret_cast<f32x4*>(out.raw + ((size_t)ray * S + ks) * 4) = rw;
        }

        // rgb_in_map (:147) pairs weight k with the UN-flipped rgb_in of sample k
        float irgb[NV][3];
        float zk = z;
        if (flip) {
            const float tk = (S > 1) ? linspace01(kl, S, step) : 0.f;
            zk = near * (1.f - tk) + far * tk;
            const float ax_ = ox + dx * zk, ay_ = oy + dy * zk, az_ = oz + dz * zk;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float dump[16];
                const ViewSample s2 = gather_view(fr.proj[v], fr.imgs + (size_t)v * fr.img_h * fr.img_w * 4, fr.img_h, fr.img_w,
                                                  fr.featmaps + (size_t)v * fr.feat_h * fr.feat_w * 32, fr.feat_h, fr.feat_w,
                                                  ax_, ay_, az_, neg, half, dump);
                irgb[v][0] = s2.rgb[0]; irgb[v][1] = s2.rgb[1]; irgb[v][2] = s2.rgb[2];
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) { irgb[v][0] = vrgb[v][0]; irgb[v][1] = vrgb[v][1]; irgb[v][2] = vrgb[v][2]; }
        }

        // raw2outputs (:90-104): alpha = 1 - exp(-sigma); T = cumprod(1 - alpha + 1e-10) exclusive
        const float alpha = 1.f - fast_exp(-sigma);
        if constexpr (P == 1) {
            const float wgt = dead ? 0.f : alpha * T;
            T = dead ? T : T * ((1.f - alpha) + 1e-10f);
            c_r = fmaf(wgt, rgb[0], c_r); c_g = fmaf(wgt, rgb[1], c_g); c_b = fmaf(wgt, rgb[2], c_b);
            depth = fmaf(wgt, zk, depth);
            acc += wgt;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                rin[3 * v + 0] = fmaf(wgt, irgb[v][0], rin[3 * v + 0]);
                rin[3 * v + 1] = fmaf(wgt, irgb[v][1], rin[3 * v + 1]);
                rin[3 * v + 2] = fmaf(wgt, irgb[v][2], rin[3 * v + 2]);
            }
            if (writer) {
                if (out.weights) out.weights[(size_t)ray * S + k] = wgt;
            }
            if constexpr (CAN_DEFER) {
                if (defer) {
                    // (culled samples and, under culling, samples with alpha <= 1e-14 carry rgb = 0: nothing to add either)
                    const bool need = active && wgt != 0.f && !((CULL || cull) && !(alpha > 1e-14f));
                    const unsigned m = (unsigned)__ballot(need);            // (both lane halves hold the ray: the low word has it all)
                    const int pos = q_cnt + __popc(m & ((1u << n) - 1u));
                    if (need) {
                        if constexpr (!GDEF) mine |= 1ull << pos;
                        if (half == 0) dq[(q_head + pos) & (DEFER_QUEUE - 1)] = uint2{(unsigned)n | ((unsigned)k << 5) | (GDEF ? (unsigned)n_q << 13 : 0u), __builtin_bit_cast(unsigned, wgt)};
                        ++n_q;
                    }
                    q_cnt += __popc(m);
                }
            }
        } else {
            // the group's P samples in order, in every lane of the group alike: sample j's values come from lane (group base + j)
            float my_wgt = 0.f;
            int my_rank = 0;
            const int base_lane = lane & ~(P - 1);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int src = base_lane + j;
                const float a_j = __shfl(alpha, src);
                const bool live = __shfl((int)(in_seg ? 1 : 0), src) != 0 && !(T < term_eps);     // this ray stops when ITS T is below the threshold
                const float wgt = live ? a_j * T : 0.f;
                T = live ? T * ((1.f - a_j) + 1e-10f) : T;
                c_r = fmaf(wgt, __shfl(rgb[0], src), c_r); c_g = fmaf(wgt, __shfl(rgb[1], src), c_g); c_b = fmaf(wgt, __shfl(rgb[2], src), c_b);
                depth = fmaf(wgt, __shfl(zk, src), depth);
                acc += wgt;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    rin[3 * v + 0] = fmaf(wgt, __shfl(irgb[v][0], src), rin[3 * v + 0]);
                    rin[3 * v + 1] = fmaf(wgt, __shfl(irgb[v][1], src), rin[3 * v + 1]);
                    rin[3 * v + 2] = fmaf(wgt, __shfl(irgb[v][2], src), rin[3 * v + 2]);
                }
                n_done += live ? 1 : 0;
                n_two += (live && __shfl((int)(two_views ? 1 : 0), src) != 0) ? 1 : 0;
                if (j == sub) { my_wgt = wgt; my_rank = n_q; }
                if constexpr (GDEF) n_q += wgt != 0.f ? 1 : 0;
            }
            if (active && half == 0 && in_seg && out.weights) out.weights[(size_t)ray * S + kl] = my_wgt;
            if constexpr (GDEF) {               // this lane's sample joins the launch's entry list (see the top of the loop)
                const bool need = active && in_seg && my_wgt != 0.f;
                const unsigned m = (unsigned)__ballot(need);
                const int pos = q_cnt + __popc(m & ((1u << n) - 1u));
                if (need && half == 0)
                    dq[(q_head + pos) & (DEFER_QUEUE - 1)] = uint2{(unsigned)n | ((unsigned)kl << 5) | ((unsigned)my_rank << 13), __builtin_bit_cast(unsigned, my_wgt)};
                q_cnt += __popc(m);
            }
        }
        STAMP(st, 6);
        // wavefront-level early termination (not in the reference): every ray of the tile is opaque
        if (early && __all(T < term_eps)) {
            if constexpr (DEFER) k_lim = k + P;         // (the colour passes still waiting, then out)
            else { k += P; break; }
        }
        // Exactly opaque (plain deferred loop): once the transmittance of all 32 rays has underflowed to 0 -- a few samples behind
        // a trained model's surface, where 1 - alpha + 1e-10 = 1e-10 -- every later weight is alpha * 0 = 0 and T stays 0: no map
        // can change any more (fma(0, x, m) = m for finite x).  The sample loop ends here; what the rest of the segment still owes
        // the outputs -- zero weights, and ray_mask's count of the samples two views see -- is settled behind it without a gather
        // or an MFMA.  Bit-exact like the other exits (kp->skip; never while culling, whose count depends on the occupancy).
        if constexpr (DEFER && !CULL && P == 1) {      // (chained form too: its launches without early termination are whole rays)
            if ((kp->skip & 2) && !cull && !early && __all(T == 0.f)) { k_lim = k + P; opaque_from = k + P; }
        }
    }
    if constexpr (DEFER && !CULL && P == 1) {      // (chained form too: its launches without early termination are whole rays)
        if (opaque_from >= 0) {
            kargs_ptr ko = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ko));
            for (int kk = opaque_from; kk < k_end; ++kk) {
                float z, px, py, pz, nv = 0.f;
                sample_point(ox, oy, oz, dx, dy, dz, near, far, flip ? (S - 1 - kk) : kk, S, step, z, px, py, pz);
#pragma unroll
                for (int v = 0; v < NV; ++v) nv += view_valid(ko->fr.proj[v], ko->fr.img_h, ko->fr.img_w, px, py, pz, neg);
                if (nv > 1.f) ++n_two;
                ++n_done;
                if (writer && ko->out.weights) ko->out.weights[(size_t)ray * S + kk] = 0.f;
                if (tally && lane == 0) { tl[T_STEPS] += 1u; tl[T_EMPTY] += 1u; tl[T_OPAQUE] += 1u; }
            }
            k = k_end;
        }
    }
    st.flush(lane);
    kargs_ptr kp = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    if (tally && lane < 6) {        // step_stats[0..5] = steps, empty-space steps, steps MINUS colour passes, opaque-tail steps, levels left out, colour passes
        const unsigned v = lane == 2 ? tl[T_STEPS] - tl[T_PASS] : (lane == 3 ? tl[T_OPAQUE] : (lane == 4 ? tl[T_LEVELS] : (lane == 5 ? tl[T_PASS] : tl[lane])));
        if (v) atomicAdd(kp->out.step_stats + lane, v);
    }

    if constexpr (FORM == FORM_SPLIT_GUARD) {
        // an operand at or beyond the f16 range (or a NaN): the hi/lo pair no longer carries the value, flag the tile
        unsigned* const gs = guard_slot();
        if (*gs) {                              // uniform: every lane reads the wave's slot
            unsigned* const gd = kp->guard;
            // the fix-up launch re-renders 32 consecutive launch slots at a time: flag the tile of every ray of this wavefront
            // (the chained form packs rays of many tiles together)
            const bool mine = CHAIN ? writer : lane == 0;
            if (mine && atomicExch(gd + GUARD_HEADER_WORDS + (CHAIN ? slot / RAYS_PER_WAVE : tile), 1u) == 0u) atomicAdd(gd, 1u);
            *gs = 0u;
        }
    }
    const __attribute__((address_space(4))) OutK& out = kp->out;
    float* const part = kp->part;
    if constexpr (CHAIN) {
        // z_vals is a function of (near, far, k): segment 0 writes every row completely
        if (out.z_vals && seg == 0)
            write_z_vals(out.z_vals, lane, (int)ray, near, far, (int)min((long)RAYS, n_items - ray0), S, step, 0, S, P);
        // a ray goes on in the next launch unless it has walked all S samples or is opaque.  Every ray of the launch writes its
        // launch slot (or -1) at ITS OWN position of the sparse list, and the wavefront adds its survivors to the counter of the
        // LIST_CHUNK entries it belongs to; compact_list_kernel then closes the gaps IN ORDER.  (Round 2 appended the survivors
        // to a dense list with one atomic per wavefront: the next launch then walked the rays in the order their wavefronts had
        // happened to finish, neighbouring wavefronts rendered unrelated patches of the image, and the re-packed levels read
        // 2-4x the bytes of level 0 with a quarter of its rays -- L2 hit rate 75-85 % against 96 %.)
        const bool goes_on = writer && k_end < S && !(T < term_eps);
        if (kp->list_out) {
            if (writer) kp->list_out[ray0 + rn] = goes_on ? (int)slot : -1;
            const unsigned long long alive = __ballot(goes_on);
            if (alive && lane == 0) atomicAdd(kp->chunk_cnt + (ray0 >> LIST_CHUNK_SHIFT), (unsigned)__popcll(alive));
        }
        if (goes_on) {
            f32x4* p = reinterpret_cast<f32x4*>(part + (size_t)slot * 16);
            f32x4 a, b, c, d;
            a[0] = GDEF ? __builtin_bit_cast(float, n_q) : c_r; a[1] = c_g; a[2] = c_b; a[3] = depth;
            b[0] = acc; b[1] = T; b[2] = (float)(n_two + 4096 * n_done); b[3] = rin[0];
            c[0] = rin[1]; c[1] = rin[2]; c[2] = rin[3]; c[3] = rin[4];
            d[0] = rin[5]; d[1] = rin[6]; d[2] = rin[7]; d[3] = rin[8];
            p[0] = a; p[1] = b; p[2] = c; p[3] = d;
        }
        if (goes_on || !writer) return false;
    } else {
        if (out.z_vals)
            write_z_vals(out.z_vals, lane, (int)ray, near, far, (int)min((long)RAYS_PER_WAVE, n_rays - ray0), S, step, k_begin, k_end);
    }
    if (writer && split > 1) {
        // partial composite of this segment: rgb, depth, acc, segment transmittance, rgb_in, #samples with >1 valid view
        float* p = part + ((size_t)slot * split + seg) * 16;      // by launch slot: `ray` may be a row of a larger array
        f32x4 a, b, c, d;
        a[0] = c_r; a[1] = c_g; a[2] = c_b; a[3] = depth;
        b[0] = acc; b[1] = T; b[2] = (float)n_two; b[3] = rin[0];
        c[0] = rin[1]; c[1] = rin[2]; c[2] = rin[3]; c[3] = rin[4];
        d[0] = rin[5]; d[1] = rin[6]; d[2] = rin[7]; d[3] = rin[8];
        reinterpret_cast<f32x4*>(p)[0] = a; reinterpret_cast<f32x4*>(p)[1] = b;
        reinterpret_cast<f32x4*>(p)[2] = c; reinterpret_cast<f32x4*>(p)[3] = d;
        return false;
    }
    if (writer) {
        // samples skipped by early termination carry weight 0 (the loop has written this segment's up to where it stopped)
        for (k = min(k, k_end); k < S; ++k) {
            if (out.weights) out.weights[(size_t)ray * S + k] = 0.f;
        }
        if constexpr (GDEF) kp->gd_cnt[slot] = n_q;          // (the colour map is colour_accumulate_kernel's)
        else { out.rgb[(size_t)ray * 3 + 0] = c_r; out.rgb[(size_t)ray * 3 + 1] = c_g; out.rgb[(size_t)ray * 3 + 2] = c_b; }
        out.depth[ray] = depth;
        out.acc[ray] = acc;
        const float q = depth / acc;                    // 1 / max(1e-10, depth / acc); torch.max keeps NaN
        out.disp[ray] = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
        if (out.rgb_in) {
#pragma unroll
            for (int i = 0; i < 9; ++i) out.rgb_in[ray * 9 + i] = rin[i];
        }
        if (out.ray_mask) out.ray_mask[ray] = (uint8_t)(n_two > 8);
        if (out.samples_done) out.samples_done[ray] = n_done;
    }
    return false;
}

// Workgroups are persistent when the launch is `dynamic`: the head image is staged into LDS once, then every wavefront
// pulls tiles from eight per-XCD queues (ka.queue: one counter per XCD, zeroed by the host before the launch) until all are
// empty.  Each XCD's queue holds a contiguous run of tiles, so neighbouring tiles still share an L2; a wave whose own
// queue is dry steals from the next XCD's.  A tile's cost varies (early termination, sample culling, padding), and with one
// 8-wave workgroup resident per CU a static grid holds the CU until its slowest tile is done -- the queue hands the next
// tile to whichever wave is free.  Static launches (one unit per wave, XCD-aware remap) remain for frames smaller than
// one round and for the sample-split geometry.
// Unified form (render_fused_kernel<., false, false, true, true, true>): the launch's wavefronts evaluate the list themselves,
// between tiles and when the tile queue has nothing left for them.  `pending` = the unit this wavefront holds a ticket for (-1:
// none).  Takes tickets UNI_BATCH at a time, evaluates the next unit if its flag is up, returns whether it did.
// One 32-entry unit of the colour list: lane i (both halves) evaluates entry i exactly as render_tile's colour pass does -- sample_point
// from the ray's row, gather_views, mean / variance, mlp_colour: same operands, same order, same bits -- and writes (r, g, b, w) where
// colour_accumulate_kernel finds it.  valid: this lane's entry is one (null entries pad a unit; an invalid lane computes on ray 0).
template <int FORM>
DEV void colour_entries(float* lds, const int lane, const bool valid, const uint4 e) {
    typedef const __attribute__((address_space(4))) KArgs* kargs_ptr;
    kargs_ptr kb = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kb));
    const int half = lane >> 5;
    const int slot = valid ? (int)e.x : 0, kk = (int)(e.y & 255u), rank = (int)(e.y >> 8);
    const int ray = kb->out.order ? kb->out.order[slot] : slot;
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(kb->rays + (size_t)ray * 8);
    const f32x4 r1 = *reinterpret_cast<const f32x4*>(kb->rays + (size_t)ray * 8 + 4);
    const int S = kb->S;
    const unsigned flags = kb->flags;
    const bool neg = (flags & GPNERF_FLAG_NEG_RAY) != 0, flip = (flags & GPNERF_FLAG_FLIP_SAMPLES) != 0;
    const float step = (S > 1) ? 1.f / (float)(S - 1) : 0.f;
    float zq, qx_, qy_, qz_;
    sample_point(r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3], flip ? (S - 1 - kk) : kk, S, step, zq, qx_, qy_, qz_);
    float xq[NV][18], vq[NV][3], cq[3], mvq[36];
    gather_views<FORM>(kb->fr, qx_, qy_, qz_, neg, half, xq, vq);
    Stamps st;
    if constexpr (FORM == FORM_F32) { mean_var_ref(xq, mvq); mlp_colour_ref(lds, lane, xq, mvq, cq, st); }
    else { mean_var(xq, mvq); mlp_colour(lds, lane, xq, mvq, cq, st); }
    kargs_ptr ko = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ko));
    if (valid && half == 0) {
        f32x4 v; v[0] = cq[0]; v[1] = cq[1]; v[2] = cq[2]; v[3] = __builtin_bit_cast(float, e.z);
        ko->gd_rgbw[(size_t)slot * S + rank] = v;
    }
    if (ko->out.step_stats && lane == 0) {     // (the diagnostic launch) one colour evaluation; [2] = steps minus colour evaluations
        atomicAdd(ko->out.step_stats + 5, 1u);
        atomicAdd(ko->out.step_stats + 2, 0xffffffffu);
    }
}

constexpr unsigned UNI_BATCH = 1;       // tickets a wavefront takes at a time: one atomic on one address per unit is no burden (what was, in a first
                                        // version: two extra device-scope loads of the counters per attempt, 9.8 -> 14.1 ms); 1 / 2 / 4 / 8 at a
                                        // time: bench frame 9.84 / 9.86 / 9.84 / 9.90 ms, 320 x 320 4.50 / 4.53 / 4.55 / 4.65 -- a wavefront that
                                        // holds tickets it is not working on yet only delays the launch's end
template <int FORM>
DEV bool consume_unit(float* lds, const int lane, long& pending, long& pending_end) {
    typedef const __attribute__((address_space(4))) KArgs* kargs_ptr;
    kargs_ptr kb = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kb));
    if (pending >= pending_end) {
        pending = (long)wave_add(kb->gd_ctrl + GD_TICKET, UNI_BATCH, lane);
        pending_end = pending + UNI_BATCH;
    }
    if (wave_load(kb->gd_flag + pending, lane) == 0u) return false;         // (not written yet, or a ticket beyond the list so far)
    const unsigned* const pe = reinterpret_cast<const unsigned*>(kb->gd_ent + (size_t)pending * 32 + (lane & 31));
    uint4 e;
    e.x = agent_load(pe); e.y = agent_load(pe + 1); e.z = agent_load(pe + 2); e.w = 0u;      // (written through by another CU: read past this one's caches)
    colour_entries<FORM>(lds, lane, e.x != 0xffffffffu, e);
    ++pending;
    return true;
}

template <int FORM, bool CHAIN, bool CULL = false, bool DEFER = false, bool GDEF = false, bool UNI = false>     // DEFER: the colour branch sample by sample; GDEF: for the launch as a whole (render_tile); UNI: ... and evaluated by this launch's own wavefronts
__global__ void __launch_bounds__(64 * GPNERF_MAX_WAVES, GPNERF_MAX_WAVES / 4)
render_fused_kernel(const KArgs ka) {
    static_assert(!DEFER || FORM != FORM_F32_FIXUP, "the fix-up launch evaluates everything");
    static_assert(!UNI || (GDEF && !CULL && (FORM == FORM_F32 || FORM == FORM_F32_FOLD)), "unified form: a listing launch of the fp32 forms");
    constexpr bool SPLIT = FORM == FORM_SPLIT || FORM == FORM_SPLIT_GUARD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    WT(0);
    if constexpr (FORM == FORM_F32_FIXUP) {
        if (__hip_atomic_load(ka.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;     // nothing was flagged
    }
    if constexpr (CHAIN) {
        if (ka.list_in && *ka.count_in == 0u) return;               // no ray is left for this segment
    }
    {
        constexpr bool REF = FORM == FORM_F32 || FORM == FORM_F32_FIXUP;
        const f32x4* src = reinterpret_cast<const f32x4*>(SPLIT ? ka.fr.head_blob_split : (REF ? ka.fr.head_blob_ref : ka.fr.head_blob));
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < (SPLIT ? gph::BLOB_WORDS : gpl::BLOB_FLOATS) / 4; i += blockDim.x) dst[i] = src[i];
        if constexpr (FORM == FORM_SPLIT_GUARD) {
            if (threadIdx.x < GUARD_LDS_SLOTS) reinterpret_cast<unsigned*>(lds)[gph::BLOB_WORDS + threadIdx.x] = 0u;
        }
    }
    __syncthreads();
    WT(1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int home = blockIdx.x & 7;            // workgroups are dealt to the XCDs round-robin
    int qx = home, dry = 0;
    int samples_per_step = 1;                   // chained form: render_tile's P and list offset of the current unit
    long entry_base = 0;
    long uni_pending = 0, uni_pending_end = 0;  // unified form: the units this wavefront holds tickets for
    int uni_left = 0;
    bool uni_drain = false;
    if (ka.stagger) {
        const int u = (int)((blockIdx.x * 8u + (unsigned)wave) * 2654435761u >> 27);        // 0..31, scattered over the chip
        for (int i = 0; i < ((u * ka.stagger) >> 5); ++i) __builtin_amdgcn_s_sleep(64);
    }
    for (;;) {
        typedef const __attribute__((address_space(4))) KArgs* kargs_ptr;
        kargs_ptr kq = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kq));            // re-read per tile rather than held across render_tile (see there)
        long tile = 0;
        int seg = 0;
        bool have_tile = !(UNI && dry >= 8);
        if (!kq->dynamic) {                     // static launch: exactly one unit per wave
            if (dry) return;
            const long unit = (long)xcd_remap(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wave;
            tile = unit / kq->split;
            seg = (int)(unit % kq->split);
            dry = 8;
        } else if (dry < 8) {
            // every tile of the launch (32 consecutive slots; chained form: the units of chain_plan())
            long n_tiles = (kq->n_rays + RAYS_PER_WAVE - 1) / RAYS_PER_WAVE;
            ChainPlan plan;
            if constexpr (CHAIN) { plan = chain_plan(kq); n_tiles = plan.bulk_tiles + plan.rem_tiles; }
            // plain form with kq->split > 1: the queue's units are (tile, sample segment) pairs, tile-major -- a frame of one to two
            // rounds of wavefronts then ends in short units instead of whole 64-step tiles (see gpnerf_render_fused)
            const int usplit = CHAIN ? 1 : kq->split;
            const long tiles_per_seg = n_tiles;
            n_tiles *= usplit;
            // fewer tiles than waves: deal them evenly, so that every CU runs the same few waves (each then steps faster) rather
            // than the first workgroups to arrive running eight and the rest none
            const long share = kq->wave_cap ? (long)kq->wave_cap : (n_tiles + gridDim.x - 1) / gridDim.x;
            if (wave >= share) { if constexpr (UNI) { have_tile = false; dry = 8; } else return; }
            STAMP_T0();
            if (have_tile) {
            const unsigned t = wave_add(kq->queue + qx, 1u, lane);
            STAMP_ADD(9, lane);
            if ((long)t >= queue_len(n_tiles, kq->chunk, qx)) {     // this XCD's queue is dry: move on to the next one
                qx = (qx + 1) & 7;
                if (++dry == 8) { if constexpr (!UNI) return; }
                if (!UNI || dry < 8) continue;
                have_tile = false;
            } else
            tile = queue_tile(kq->chunk, qx, t);
            }
            if (usplit > 1) {
                if (kq->seg_major) { seg = (int)(tile / tiles_per_seg); tile -= (long)seg * tiles_per_seg; }      // all tiles' first segment, then the second ...
                else { seg = (int)(tile % usplit); tile /= usplit; }
            }
            if constexpr (CULL) { if (kq->tile_order) tile = kq->tile_order[tile]; }
            if constexpr (CHAIN) {
                seg = kq->seg;
                samples_per_step = 1; entry_base = 0;
                if (tile >= plan.bulk_tiles) { samples_per_step = plan.rem_p; entry_base = plan.bulk_tiles * RAYS_PER_WAVE; tile -= plan.bulk_tiles; }
            }
            if constexpr (FORM == FORM_F32_FIXUP) {                 // only the tiles the split form flagged
                if (wave_load(kq->guard + GUARD_HEADER_WORDS + tile, lane) == 0u) continue;
            }
        }
        constexpr int F = FORM == FORM_F32_FIXUP ? FORM_F32 : FORM;
        STAMP_T0();
        if (have_tile) {
        if constexpr (CHAIN) {
            if (samples_per_step == 8) render_tile<F, true, 8, false, GDEF, GDEF, UNI>(lds, lane, tile, seg, entry_base);
            else if (samples_per_step == 4) render_tile<F, true, 4, false, GDEF, GDEF, UNI>(lds, lane, tile, seg, entry_base);
            else if (samples_per_step == 2) render_tile<F, true, 2, false, GDEF, GDEF, UNI>(lds, lane, tile, seg, entry_base);
            else render_tile<F, true, 1, false, DEFER, GDEF, UNI>(lds, lane, tile, seg, entry_base);
        } else {
            render_tile<F, false, 1, CULL, DEFER, GDEF, UNI>(lds, lane, tile, seg);
        }
        STAMP_ADD(8, lane);
        WT(3);
        WT_COUNT();
        }
        if constexpr (UNI) {
            // between tiles: what the list holds beyond the tickets, at most uni_budget units; with no tile left (drain): units until
            // every wavefront has listed its last entry -- render_tile flushes at every tile's end -- and this one's ticket lies
            // beyond the list.  ONE site for both (the colour branch's code is 28 KB of the instruction cache).
            uni_left = kq->uni_budget;
            if (!have_tile && !uni_drain) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_add(kq->gd_ctrl + GD_DONE, 1u, lane);
                uni_drain = true;
            }
            for (;;) {
                if (!uni_drain && uni_left <= 0) break;
                if (consume_unit<F>(lds, lane, uni_pending, uni_pending_end)) { --uni_left; continue; }
                if (!uni_drain) break;
                typedef const __attribute__((address_space(4))) KArgs* kargs_ptr;
                kargs_ptr kd = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kd));
                if (wave_load(kd->gd_ctrl + GD_DONE, lane) >= (unsigned)kd->gd_waves) {
                    if (uni_pending >= (long)wave_load(kd->gd_ctrl + GD_COUNT, lane)) return;     // (every ticket it holds lies beyond the list)
                }
                __builtin_amdgcn_s_sleep(32);
            }
        }
    }
}


// Frame-level deferral, second launch: the colour branch of the entries render_fused_kernel<., ., ., true, true> appended, 32 per
// wavefront step, by persistent workgroups on a unit queue (unit = 32 consecutive entries: neighbouring tiles' samples, dealt to the
// XCDs in chunks like the tiles).  A unit costs the same whichever rays its entries belong to, so the launch ends with every
// wavefront within one ~15 us unit of the others -- where the tile-level passes left a frame's end to whichever wavefronts had
// drawn the tiles with the most non-zero weights (tools/wave_times.py: last exit 0.9 ms behind the median on the bench frame).
// Lane i evaluates entry i exactly as render_tile's colour pass does (sample_point from the ray's row, gather_views, mean / variance,
// mlp_colour): same operands, same order, same bits.
template <int FORM>
DEV void colour_unit(float* lds, const int lane, const long unit, const unsigned total) {
    typedef const __attribute__((address_space(4))) KArgs* kargs_ptr;
    kargs_ptr kb = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kb));
    const size_t at = (size_t)unit * 32 + (lane & 31);
    const bool valid = at < total;
    colour_entries<FORM>(lds, lane, valid, kb->gd_ent[valid ? at : (size_t)total - 1]);
}

constexpr int UNIT_WAVES = 8;            // wavefronts of a colour_units_kernel workgroup (12, three per SIMD at 168 registers: the same time, A/B on one box)
template <int FORM>
__global__ void __launch_bounds__(64 * UNIT_WAVES)
colour_units_kernel(const KArgs ka) {
    static_assert(FORM == FORM_F32 || FORM == FORM_F32_FOLD, "frame-level deferral: the fp32 forms");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned total = ka.gd_ctrl[GD_COUNT];            // (written by the launch before this one)
    if (total == 0u) return;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(FORM == FORM_F32 ? ka.fr.head_blob_ref : ka.fr.head_blob);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < gpl::BLOB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long n_units = ((long)total + 31) / 32;
    int qx = blockIdx.x & 7, dry = 0;
    for (;;) {
        typedef const __attribute__((address_space(4))) KArgs* kargs_ptr;
        kargs_ptr kq = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kq));
        const unsigned t = wave_add(kq->gd_ctrl + GD_QUEUE + qx, 1u, lane);
        if ((long)t >= queue_len(n_units, kq->chunk, qx)) {
            qx = (qx + 1) & 7;
            if (++dry == 8) return;
            continue;
        }
        colour_unit<FORM>(lds, lane, queue_tile(kq->chunk, qx, t), total);
    }
}

// Frame-level deferral, third launch: c = fma(w, rgb, c) over a ray's entries in sample order (rank order) -- the colour map exactly
// as the sample loop accumulates it.  One thread per launch slot; a ray's entries are consecutive 16-byte rows.
__global__ void __launch_bounds__(256) colour_accumulate_kernel(const int* __restrict__ gd_cnt, const f32x4* __restrict__ gd_rgbw, const long n_slots,
                                                                const int S, const int32_t* __restrict__ order, float* __restrict__ rgb) {
    const long slot = (long)blockIdx.x * 256 + threadIdx.x;
    if (slot >= n_slots) return;
    const int cnt = gd_cnt[slot];
    const f32x4* __restrict__ p = gd_rgbw + (size_t)slot * S;
    float c_r = 0.f, c_g = 0.f, c_b = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const f32x4 v = p[i];
        c_r = fmaf(v[3], v[0], c_r); c_g = fmaf(v[3], v[1], c_g); c_b = fmaf(v[3], v[2], c_b);
    }
    const long ray = order ? (long)order[slot] : slot;
    rgb[ray * 3 + 0] = c_r; rgb[ray * 3 + 1] = c_g; rgb[ray * 3 + 2] = c_b;
}

// The keep bits of occupancy culling for every sample of a launch, before it.  One wavefront per tile: lane = (ray of the tile,
// parity of the sample index), so one load instruction reads the occupancy around 32 neighbouring rays at (nearly) the same depth
// -- a few cache lines -- where lane = sample along ONE ray touched 64 (0.22 -> 0.09 ms per 512x512x64 frame).  The position,
// grid coordinate and occupancy arithmetic is the sample loop's own (same functions), so the bits are what the loop would decide.
// mask[slot * 2 + w] bit j <-> composite step 64 w + j (which evaluates sample S-1-k under flip).
__global__ void __launch_bounds__(256) occupancy_mask_kernel(const FrameK fr, const float* __restrict__ rays, const int32_t* __restrict__ order,
                                                             const long n_rays, const int S, const int flip,
                                                             unsigned long long* __restrict__ mask) {
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, parity = lane >> 5;
    const long slot = tile * RAYS_PER_WAVE + (lane & 31);
    if (tile * RAYS_PER_WAVE >= n_rays) return;
    const bool active = slot < n_rays;
    const long sl = active ? slot : n_rays - 1;
    const int ray = order ? order[sl] : (int)sl;
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rays + (size_t)ray * 8);
    const f32x4 r1 = *reinterpret_cast<const f32x4*>(rays + (size_t)ray * 8 + 4);
    const float step = (S > 1) ? 1.f / (float)(S - 1) : 0.f;
    unsigned long long m0 = 0ull, m1 = 0ull;
#pragma unroll 4
    for (int k = parity; k < S; k += 2) {
        const int ks = flip ? (S - 1 - k) : k;
        const float t = (S > 1) ? linspace01(ks, S, step) : 0.f;
        const float z = r1[2] * (1.f - t) + r1[3] * t;
        const float px = r0[0] + r0[3] * z, py = r0[1] + r1[0] * z, pz = r0[2] + r1[1] * z;
        float gx, gy, gz;
        grid_coords(fr, px, py, pz, gx, gy, gz);
        const bool keep = sample_occupancy(fr.occ, fr.vol_dhw[0][0], fr.vol_dhw[0][1], fr.vol_dhw[0][2], gx, gy, gz) > 0.f;
        const unsigned long long bit = keep ? 1ull << (k & 63) : 0ull;
        if (k < 64) m0 |= bit; else m1 |= bit;
    }
    m0 |= ((unsigned long long)(unsigned)__shfl_xor((int)(m0 >> 32), 32) << 32) | (unsigned)__shfl_xor((int)m0, 32);
    m1 |= ((unsigned long long)(unsigned)__shfl_xor((int)(m1 >> 32), 32) << 32) | (unsigned)__shfl_xor((int)m1, 32);
    if (active && parity == 0) { mask[slot * 2] = m0; mask[slot * 2 + 1] = m1; }
}

// Under culling a tile costs as many steps as its 32 rays keep samples at (the union of their masks): 0 ... S, known before the
// launch.  Handed out in launch order, the long tiles that come late set the frame's time (10 % occupied blocks: 56 steps on the
// busiest wavefront slot against 37 on average); longest first, the queue packs them (38).  tile_steps_kernel counts, and
// tile_sort_kernel -- one workgroup, a STABLE counting sort over 16 classes of step counts -- orders: tiles of one class keep
// their launch order, so neighbouring tiles still follow each other (an order that scatters them costs the gathers their L2
// locality: +1.2 ms on a frame that culls nothing).
constexpr int CULL_CLASSES = 16, SORT_THREADS = 1024;
__global__ void __launch_bounds__(256) tile_steps_kernel(const unsigned long long* __restrict__ mask, const long n_rays, int* __restrict__ steps) {
    const long slot = (long)blockIdx.x * 256 + threadIdx.x;
    unsigned long long m0 = 0ull, m1 = 0ull;
    if (slot < n_rays) { m0 = mask[slot * 2]; m1 = mask[slot * 2 + 1]; }
    unsigned a = (unsigned)m0, b = (unsigned)(m0 >> 32), c = (unsigned)m1, d = (unsigned)(m1 >> 32);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        a |= (unsigned)__shfl_xor((int)a, o); b |= (unsigned)__shfl_xor((int)b, o);
        c |= (unsigned)__shfl_xor((int)c, o); d |= (unsigned)__shfl_xor((int)d, o);
    }
    if ((threadIdx.x & 31) == 0 && slot < n_rays) steps[slot / RAYS_PER_WAVE] = __popc(a) + __popc(b) + __popc(c) + __popc(d);
}
__global__ void __launch_bounds__(SORT_THREADS) tile_sort_kernel(const int* __restrict__ steps, const int n_tiles, const int S, int* __restrict__ order) {
    __shared__ int cnt[CULL_CLASSES * SORT_THREADS];           // [class][thread], class 0 = most steps; 64 KB
    __shared__ int part[SORT_THREADS];
    const int i = threadIdx.x, per = (n_tiles + SORT_THREADS - 1) / SORT_THREADS, width = (S + CULL_CLASSES - 1) / CULL_CLASSES;
    const int lo = min(i * per, n_tiles), hi = min(lo + per, n_tiles);
    auto cls = [&](int t) { return CULL_CLASSES - 1 - min(CULL_CLASSES - 1, steps[t] / width); };
    for (int c = 0; c < CULL_CLASSES; ++c) cnt[c * SORT_THREADS + i] = 0;
    for (int t = lo; t < hi; ++t) ++cnt[cls(t) * SORT_THREADS + i];
    __syncthreads();
    // exclusive prefix over the flattened [class][thread] table: thread i owns entries 16 i ... 16 i + 15 (all of one class)
    int sum = 0;
    for (int e = 0; e < CULL_CLASSES; ++e) sum += cnt[i * CULL_CLASSES + e];
    part[i] = sum;
    __syncthreads();
    for (int o = 1; o < SORT_THREADS; o <<= 1) {
        const int v = i >= o ? part[i - o] : 0;
        __syncthreads();
        part[i] += v;
        __syncthreads();
    }
    int run = part[i] - sum;
    for (int e = 0; e < CULL_CLASSES; ++e) { const int v = cnt[i * CULL_CLASSES + e]; cnt[i * CULL_CLASSES + e] = run; run += v; }
    __syncthreads();
    for (int t = lo; t < hi; ++t) order[cnt[cls(t) * SORT_THREADS + i]++] = t;
}

// gpnerf_fold_volumes: out_geometry_fc's 32 columns of level l applied to every voxel of level l (see gather_folded), both
// folded levels in one launch.  One wavefront per 32 voxels: B operand = the voxel's 32 channels in the sample loop's own k-step order, A = the layer's two weight
// tiles from the head image (k-steps 16 l ... 16 l + 15), fp32 MFMA; the lane of (voxel, half) stores its 2 x 16 accumulator
// registers as 128 contiguous bytes, which is what the lane of (ray, half) reads back per tap.  No bias: it is added once, after
// the interpolation (out-of-volume taps contribute nothing, exactly as zero padding does before the layer).
struct FoldArgs {                  // the levels GPNERF_FOLD_FIRST_LEVEL .. of one frame, one launch
    const float* vol[GPNERF_LEVELS - GPNERF_FOLD_FIRST_LEVEL];
    float* out[GPNERF_LEVELS - GPNERF_FOLD_FIRST_LEVEL];
    long tiles_before[GPNERF_LEVELS - GPNERF_FOLD_FIRST_LEVEL + 1];      // 32-voxel tiles of the levels before this one
    long n_vox[GPNERF_LEVELS - GPNERF_FOLD_FIRST_LEVEL];
};
__global__ void __launch_bounds__(256) fold_volume_kernel(const float* __restrict__ head_blob, const FoldArgs a) {
    constexpr int NL = GPNERF_LEVELS - GPNERF_FOLD_FIRST_LEVEL;
    __shared__ __attribute__((aligned(16))) float w[NL * 2 * 16 * 64];     // [level][tile][4 groups][64 lanes][4 k-steps]
    for (int i = threadIdx.x; i < NL * 2 * 16 * 64 / 4; i += blockDim.x) {
        const int l = i / 512, m = (i % 512) / 256, r = i % 256;
        reinterpret_cast<f32x4*>(w)[i] = reinterpret_cast<const f32x4*>(head_blob + gpl::w_off(gpl::GEO) + m * gpl::NT[gpl::GEO] * 64 +
                                                                         (GPNERF_FOLD_FIRST_LEVEL + l) * 16 * 64)[r];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    for (long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6); unit < a.tiles_before[NL]; unit += (long)gridDim.x * 4) {
        int l = 0;
#pragma unroll
        for (int i = 1; i < NL; ++i) l = unit >= a.tiles_before[i] ? i : l;
        const long tile = unit - a.tiles_before[l], n_vox = a.n_vox[l];
        const long vox = tile * 32 + n;
        const long v = vox < n_vox ? vox : n_vox - 1;
        const f32x4* q = reinterpret_cast<const f32x4*>(a.vol[l] + v * GPNERF_CH + half * 16);
        float b[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const f32x4 t = q[i]; b[4 * i] = t[0]; b[4 * i + 1] = t[1]; b[4 * i + 2] = t[2]; b[4 * i + 3] = t[3]; }
        f32x16 g0, g1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
        mfma_tile<16>(w + l * 2048, lane, b, g0);
        mfma_tile<16>(w + l * 2048 + 16 * 64, lane, b, g1);
        if (vox < n_vox) {
            f32x4* o = reinterpret_cast<f32x4*>(a.out[l] + vox * (2 * GPNERF_CH) + half * GPNERF_CH);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = f32x4{g0[4 * i], g0[4 * i + 1], g0[4 * i + 2], g0[4 * i + 3]};
                o[4 + i] = f32x4{g1[4 * i], g1[4 * i + 1], g1[4 * i + 2], g1[4 * i + 3]};
            }
        }
    }
}

// merge the per-segment partial composites of a ray front to back: out = sum_s (prod_{j<s} T_j) * partial_s
__global__ void combine_segments_kernel(const float* __restrict__ part, const long n_rays, const int S, const int split,
                                        const OutK out) {
    const long slot = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_rays) return;
    const long ray = out.order ? (long)out.order[slot] : slot;     // the row the slot's ray lives in
    float Tp = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, depth = 0.f, acc = 0.f, n_two = 0.f;
    float rin[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) rin[i] = 0.f;
    for (int s = 0; s < split; ++s) {
        const f32x4* p = reinterpret_cast<const f32x4*>(part + ((size_t)slot * split + s) * 16);
        const f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
        cr = fmaf(Tp, a[0], cr); cg = fmaf(Tp, a[1], cg); cb = fmaf(Tp, a[2], cb);
        depth = fmaf(Tp, a[3], depth);
        acc = fmaf(Tp, b[0], acc);
        n_two += b[2];
        rin[0] = fmaf(Tp, b[3], rin[0]);
        rin[1] = fmaf(Tp, c[0], rin[1]); rin[2] = fmaf(Tp, c[1], rin[2]); rin[3] = fmaf(Tp, c[2], rin[3]); rin[4] = fmaf(Tp, c[3], rin[4]);
        rin[5] = fmaf(Tp, d[0], rin[5]); rin[6] = fmaf(Tp, d[1], rin[6]); rin[7] = fmaf(Tp, d[2], rin[7]); rin[8] = fmaf(Tp, d[3], rin[8]);
        if (out.weights && s > 0) {
            const int k0 = (int)(((long)S * s) / split), k1 = (int)(((long)S * (s + 1)) / split);
            for (int k = k0; k < k1; ++k) out.weights[(size_t)ray * S + k] *= Tp;
        }
        Tp *= b[1];
    }
    out.rgb[ray * 3 + 0] = cr; out.rgb[ray * 3 + 1] = cg; out.rgb[ray * 3 + 2] = cb;
    out.depth[ray] = depth;
    out.acc[ray] = acc;
    const float q = depth / acc;
    out.disp[ray] = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
    if (out.rgb_in) {
#pragma unroll
        for (int i = 0; i < 9; ++i) out.rgb_in[ray * 9 + i] = rin[i];
    }
    if (out.ray_mask) out.ray_mask[ray] = (uint8_t)(n_two > 8.f);
}

// ---------------------------------------------------------------------------------------------
// NeRFHead.forward on pre-gathered features, and its two halves as the reference's renderers call them separately
// ---------------------------------------------------------------------------------------------
// MODE 0: NeRFHead.forward (trainhead.py:159-163): vol_feat, rgb_feat, mask -> raw
// MODE 1: NeRFSigmaHead.test_forward (:61-76): vol_feat, rgb_feat -> sigma_feat [P][64], globalfeat [P][134] = [sigma_feat, mean, var]
// MODE 2: NeRFRGBHead.forward (:118-145): sigma_feat [P][64] (in place of vol_feat), rgb_feat, mask -> raw
template <int NWAVES, int MODE>
__global__ void __launch_bounds__(NWAVES * 64, NWAVES / 4)
head_forward_kernel(const float* __restrict__ blob, const float* __restrict__ vol_feat, const float* __restrict__ rgb_feat,
                    const float* __restrict__ mask, const long P, float* __restrict__ raw, float* __restrict__ globalfeat) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(blob);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < gpl::BLOB_FLOATS / 4; i += NWAVES * 64) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
    const long ntile = (P + 31) / 32;
    for (long tile = (long)blockIdx.x * NWAVES + wave; tile < ntile; tile += (long)gridDim.x * NWAVES) {
        const bool active = tile * 32 + n < P;
        const long p = active ? tile * 32 + n : P - 1;
        // reference order (head_layout.h gpr): slot t of half h = element ref35(t, h) of [r, g, b, f0 .. f31]; read straight in k order
        float x[NV][18];
        float nvalid = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float* xv = rgb_feat + (p * NV + v) * 35;
            x[v][0] = half ? xv[1] : xv[0];
            x[v][1] = half ? 0.f : xv[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) x[v][2 + i] = xv[3 + 2 * i + half];
            if (MODE != 1) nvalid += mask[p * NV + v];
        }
        float sf[32];
        if constexpr (MODE == 2) {
            // the accumulator layout of geo_eval_ref's output: feature 32 m + 2 r + half in sf[16 m + r]
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) sf[16 * m + r] = vol_feat[p * 64 + 32 * m + 2 * r + half];
        } else {
            float fv[64];
#pragma unroll
            for (int t = 0; t < 64; ++t) fv[t] = vol_feat[p * 128 + 2 * t + half];
            geo_eval_ref(lds, lane, fv, sf);
        }
        if constexpr (MODE == 1) {
            if (active) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = sf[16 * m + r];
                        raw[p * 64 + 32 * m + 2 * r + half] = v;
                        globalfeat[p * 134 + 32 * m + 2 * r + half] = v;
                    }
#pragma unroll
                for (int t = 0; t < 18; t += 2) {       // fused_mean_variance (trainhead.py:20-24), as mlp_eval_ref computes it
                    const f32x2 x0 = {x[0][t], x[0][t + 1]}, x1 = {x[1][t], x[1][t + 1]}, x2 = {x[2][t], x[2][t + 1]};
                    const f32x2 m_ = div3((x0 + x1) + x2);
                    const f32x2 a = x0 - m_, b = x1 - m_, cc = x2 - m_;
                    const f32x2 vr = div3((a * a + b * b) + cc * cc);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int c = half ? gpr::ref35(t + j, 1) : gpr::ref35(t + j, 0);      // this lane's slot of the 35-vector, -1 = pad
                        if (c >= 0) {
                            globalfeat[p * 134 + 64 + c] = m_[j];
                            globalfeat[p * 134 + 99 + c] = vr[j];
                        }
                    }
                }
            }
        } else {
            float sigma, rgb[3];
            Stamps st;
            mlp_eval_ref(lds, lane, sf, x, nvalid, sigma, rgb, st);
            if (active && half == 0) {
                f32x4 rw; rw[0] = rgb[0]; rw[1] = rgb[1]; rw[2] = rgb[2]; rw[3] = sigma;
                *reinterpret_cast<f32x4*>(raw + p * 4) = rw;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stage entry points: the same device functions as the fused kernel, one stage per launch
// ---------------------------------------------------------------------------------------------
// get_sampling_points + pts_to_can_pts + get_grid_coords (BaseRender.py:35-73): one lane per sample
__global__ void stage_points_kernel(const FrameK fr, const float* __restrict__ rays, const long N, const int S,
                                    float* __restrict__ pts, float* __restrict__ zv, float* __restrict__ grid) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * S) return;
    const long r = i / S;
    const int k = (int)(i % S);
    const float* ry = rays + r * 8;
    const float step = (S > 1) ? 1.f / (float)(S - 1) : 0.f;
    const float t = (S > 1) ? linspace01(k, S, step) : 0.f;
    const float z = ry[6] * (1.f - t) + ry[7] * t;
    const float px = ry[0] + ry[3] * z, py = ry[1] + ry[4] * z, pz = ry[2] + ry[5] * z;
    float gx, gy, gz;
    grid_coords(fr, px, py, pz, gx, gy, gz);
    if (pts) { pts[i * 3 + 0] = px; pts[i * 3 + 1] = py; pts[i * 3 + 2] = pz; }
    if (zv) zv[i] = z;
    if (grid) { grid[i * 3 + 0] = gx; grid[i * 3 + 1] = gy; grid[i * 3 + 2] = gz; }
}

// SparseConvNet.forward's sampling (SparseConvNet.py:113-122): two lanes per point (16 channels each)
__global__ void stage_volume_kernel(const FrameK fr, const float* __restrict__ grid, const long P, float* __restrict__ vol_feat) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long p = i >> 1;
    const int half = (int)(i & 1);
    if (p >= P) return;
    const float gx = grid[p * 3 + 0], gy = grid[p * 3 + 1], gz = grid[p * 3 + 2];
#pragma unroll
    for (int l = 0; l < GPNERF_LEVELS; ++l) {
        float f[16];
        gather_volume(fr.vol[l], fr.vol_dhw[l][0], fr.vol_dhw[l][1], fr.vol_dhw[l][2], gx, gy, gz, half, f);
        f32x4* o = reinterpret_cast<f32x4*>(vol_feat + p * 128 + 32 * l + 16 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) { f32x4 v; v[0] = f[4 * q]; v[1] = f[4 * q + 1]; v[2] = f[4 * q + 2]; v[3] = f[4 * q + 3]; o[q] = v; }
    }
}

// Projector.compute (BaseRender.py:326-363) for arbitrary points: two lanes per point
__global__ void stage_project_kernel(const FrameK fr, const float* __restrict__ pts, const long P, const int neg,
                                     float* __restrict__ rgb_feat, float* __restrict__ mask) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long p = i >> 1;
    const int half = (int)(i & 1);
    if (p >= P) return;
    const float px = pts[p * 3 + 0], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float f[16];
        const ViewSample s = gather_view(fr.proj[v], fr.imgs + (size_t)v * fr.img_h * fr.img_w * 4, fr.img_h, fr.img_w,
                                         fr.featmaps + (size_t)v * fr.feat_h * fr.feat_w * 32, fr.feat_h, fr.feat_w, px, py, pz,
                                         neg != 0, half, f);
        float* o = rgb_feat + (p * NV + v) * 35;
#pragma unroll
        for (int c = 0; c < 16; ++c) o[3 + 16 * half + c] = f[c];
        if (half == 0) {
            o[0] = s.rgb[0]; o[1] = s.rgb[1]; o[2] = s.rgb[2];
            mask[p * NV + v] = s.valid;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// raw2outputs alone (BaseRender.py:75-107): one lane per ray, sequential product over samples
// ---------------------------------------------------------------------------------------------
__global__ void composite_kernel(const float* __restrict__ raw, const float* __restrict__ zv, const float* __restrict__ nvalid,
                                 const long N, const int S, const int neg, const OutK out) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, d = 0.f, a = 0.f;
    int n_two = 0;
    for (int k = 0; k < S; ++k) {
        const int ks = neg ? (S - 1 - k) : k;
        const f32x4 rw = *reinterpret_cast<const f32x4*>(raw + ((size_t)r * S + ks) * 4);
        const float alpha = 1.f - fast_exp(-rw[3]);
        const float w = alpha * T;
        T = T * ((1.f - alpha) + 1e-10f);
        cr = fmaf(w, rw[0], cr); cg = fmaf(w, rw[1], cg); cb = fmaf(w, rw[2], cb);
        d = fmaf(w, zv[(size_t)r * S + k], d);
        a += w;
        if (out.weights) out.weights[(size_t)r * S + k] = w;
        if (nvalid && nvalid[(size_t)r * S + k] > 1.f) ++n_two;
    }
    out.rgb[r * 3 + 0] = cr; out.rgb[r * 3 + 1] = cg; out.rgb[r * 3 + 2] = cb;
    out.depth[r] = d; out.acc[r] = a;
    const float q = d / a;
    out.disp[r] = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
    if (out.ray_mask) out.ray_mask[r] = (uint8_t)(n_two > 8);
}

// ---------------------------------------------------------------------------------------------
// get_rays + get_near_far (libs/datasets/data_utils.py:47-63,96-130), one lane per pixel
// ---------------------------------------------------------------------------------------------
struct RayCam {
    float Kinv[9], Rinv[9], o[3], bmin[3], bmax[3], T[3];      // the inference renderer's float32 pipeline (demo = 1)
    double Kinv_d[9], Rinv_d[9], o_d[3], bmin_d[3], bmax_d[3];  // the dataset's float64 pipeline (demo = 0)
};

// one length-3 row of torch's CPU `@` (sgemm): k = 0, 1, 2 accumulated with fused multiply-adds (oracle: MM3)
DEV float mm3(float a0, float b0, float a1, float b1, float a2, float b2) { return fmaf(a2, b2, fmaf(a1, b1, a0 * b0)); }

// demo = 0: the dataset's get_rays + get_near_far in the precision numpy runs them in (data_utils.py:47-63,96-130 via sample_ray
// :294-300): float64 camera products (dgemm order: fused multiply-adds over k = 0,1,2) rounded once to float32 rays; float64
// plane hits and on-box tests on those float32 values against float64 bounds (+-0.01 added in float64, :98); norm_ray in
// float32; the two float64 distances rounded to float32.  Bit-exact against tests/golden/rays_*.npz.
DEV void dataset_ray(const RayCam& cam, const float i, const float j, float (&o)[3], float (&d)[3], bool& keep, float& near, float& far) {
    double pc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) pc[a] = fma(1.0, cam.Kinv_d[a * 3 + 2], fma((double)j, cam.Kinv_d[a * 3 + 1], (double)i * cam.Kinv_d[a * 3 + 0]));
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double pw = fma(pc[2], cam.Rinv_d[a * 3 + 2], fma(pc[1], cam.Rinv_d[a * 3 + 1], pc[0] * cam.Rinv_d[a * 3 + 0])) + cam.o_d[a];
        o[a] = (float)cam.o_d[a];
        float da = (float)(pw - cam.o_d[a]);
        if (fabsf(da) < 1e-5f) da = 1e-5f;             // ray_d[np.abs(ray_d) < 1e-5] = 1e-5 (:101), on the float32 array
        d[a] = da;
    }
    const double eps = 1e-6;
    double p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0};
    int cnt = 0;
#pragma unroll
    for (int m = 0; m < 6; ++m) {                      // order: min_x,min_y,min_z,max_x,max_y,max_z (:99-103)
        const int a = m % 3;
        const double bd = (m < 3) ? cam.bmin_d[a] : cam.bmax_d[a];
        const double tt = (bd - (double)o[a]) / (double)d[a];
        const double hx = tt * (double)d[0] + (double)o[0], hy = tt * (double)d[1] + (double)o[1], hz = tt * (double)d[2] + (double)o[2];
        const bool ok = hx >= cam.bmin_d[0] - eps && hx <= cam.bmax_d[0] + eps && hy >= cam.bmin_d[1] - eps &&
                        hy <= cam.bmax_d[1] + eps && hz >= cam.bmin_d[2] - eps && hz <= cam.bmax_d[2] + eps;
        if (ok) {
            if (cnt == 0) { p0[0] = hx; p0[1] = hy; p0[2] = hz; }
            else if (cnt == 1) { p1[0] = hx; p1[1] = hy; p1[2] = hz; }
            ++cnt;
        }
    }
    keep = (cnt == 2);
    near = 0.f; far = 0.f;
    if (keep) {
        const float nd = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);          // np.linalg.norm on float32 (:120)
        const double v0x = p0[0] - o[0], v0y = p0[1] - o[1], v0z = p0[2] - o[2];
        const double v1x = p1[0] - o[0], v1y = p1[1] - o[1], v1z = p1[2] - o[2];
        const double sg = ((v0x * d[0] + v0y * d[1]) + v0z * d[2]) < 0.0 ? -1.0 : 1.0;   // both from p0 (:123,126)
        const double d0 = sqrt((v0x * v0x + v0y * v0y) + v0z * v0z) / (double)nd * sg;
        const double d1 = sqrt((v1x * v1x + v1y * v1y) + v1z * v1z) / (double)nd * sg;
        near = (float)fmin(d0, d1); far = (float)fmax(d0, d1);
    }
}

// demo = 1: the inference renderer's variant (libs/renders/demo_render.py:201-239), float32 throughout: box used as given,
// no small-|d| clamp, d1 negated under neg_ray instead of the sign test; sel (optional) restricts the pixels (:179-200)
__global__ void make_rays_kernel(const int H, const int W, RayCam cam, const int demo, const int neg,
                                 const uint8_t* __restrict__ sel, const int* __restrict__ box_bits, float* __restrict__ rays,
                                 uint8_t* __restrict__ hit) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    if (sel && !sel[idx]) { hit[idx] = 0; return; }
    if (box_bits) {
        // the world box of the occupied voxels as gpnerf_select_pixels left it on the device (order-preserving integer images of
        // min xyz / max xyz), z padded by 5 cm (demo_render.py:168-175): no host round trip between the two launches
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int lo = box_bits[a], hi = box_bits[3 + a];
            cam.bmin[a] = __int_as_float(lo >= 0 ? lo : lo ^ 0x7FFFFFFF);
            cam.bmax[a] = __int_as_float(hi >= 0 ? hi : hi ^ 0x7FFFFFFF);
        }
        cam.bmin[2] -= 0.05f;
        cam.bmax[2] += 0.05f;
    }
    const float i = (float)(idx % W), j = (float)(idx / W);
    float o[3], d[3], near = 0.f, far = 0.f;
    bool keep;
    if (!demo) {
        dataset_ray(cam, i, j, o, d, keep, near, far);
    } else {
        // pixel_camera = xy1 @ K_inv^T; pixel_world = (pixel_camera - T) @ R; rays_d = pixel_world - rays_o (demo_render.py:204-210)
        float pc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) pc[a] = mm3(i, cam.Kinv[a * 3 + 0], j, cam.Kinv[a * 3 + 1], 1.f, cam.Kinv[a * 3 + 2]);
        const float t0 = pc[0] - cam.T[0], t1 = pc[1] - cam.T[1], t2 = pc[2] - cam.T[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[a] = cam.o[a];
            d[a] = mm3(t0, cam.Rinv[a * 3 + 0], t1, cam.Rinv[a * 3 + 1], t2, cam.Rinv[a * 3 + 2]) - cam.o[a];
        }
        const float eps = 1e-6f;
        float p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0};
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < 6; ++m) {                     // order: min_x,min_y,min_z,max_x,max_y,max_z (:211-214)
            const int a = m % 3;
            const float bd = (m < 3) ? cam.bmin[a] : cam.bmax[a];
            const float tt = (bd - o[a]) / d[a];
            const float hx = tt * d[0] + o[0], hy = tt * d[1] + o[1], hz = tt * d[2] + o[2];
            const bool ok = hx >= cam.bmin[0] - eps && hx <= cam.bmax[0] + eps && hy >= cam.bmin[1] - eps &&
                            hy <= cam.bmax[1] + eps && hz >= cam.bmin[2] - eps && hz <= cam.bmax[2] + eps;
            if (ok) {
                if (cnt == 0) { p0[0] = hx; p0[1] = hy; p0[2] = hz; }
                else if (cnt == 1) { p1[0] = hx; p1[1] = hy; p1[2] = hz; }
                ++cnt;
            }
        }
        keep = (cnt == 2);
        if (keep) {
            // torch.norm(dim=1) is sqrt(fma(z, z, fma(y, y, x*x))) on the reference's CPU path (demo_render.py:232-234)
            const auto norm3 = [](float x, float y, float z) { return sqrtf(fmaf(z, z, fmaf(y, y, x * x))); };
            const float nd = norm3(d[0], d[1], d[2]);
            const float d0 = norm3(p0[0] - o[0], p0[1] - o[1], p0[2] - o[2]) / nd;
            const float d1 = norm3(p1[0] - o[0], p1[1] - o[1], p1[2] - o[2]) / nd * (neg ? -1.f : 1.f);
            near = fminf(d0, d1); far = fmaxf(d0, d1);
        }
    }
    hit[idx] = (uint8_t)keep;
    f32x4 a, b;
    a[0] = o[0]; a[1] = o[1]; a[2] = o[2]; a[3] = d[0];
    b[0] = d[1]; b[1] = d[2]; b[2] = near; b[3] = far;
    *reinterpret_cast<f32x4*>(rays + (size_t)idx * 8) = a;
    *reinterpret_cast<f32x4*>(rays + (size_t)idx * 8 + 4) = b;
}

// ---------------------------------------------------------------------------------------------
// progressive ray selection (libs/renders/demo_render.py:166-200): every occupied level-1 voxel (masks3d > threshold)
// becomes a world point; its 4 neighbouring pixels in the target view are marked, and the points' world AABB is reduced
// ---------------------------------------------------------------------------------------------
struct SelGeom { float voxel[3], bmin[3], Rh[9], Th[3], pose[12], K[9]; };

DEV int ordered_int(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }

__global__ void select_pixels_kernel(const float* __restrict__ occ, const int D, const int H, const int W, const float thr,
                                     const SelGeom g, const int ih, const int iw, uint8_t* __restrict__ sel, int* __restrict__ mm) {
    // grid-stride over the voxels: the world box of the occupied ones is kept per lane, reduced over the wavefront at the end and
    // merged with ONE atomic per wave and bound (same-address atomics run at ~10 ns each: one per occupied voxel was a millisecond)
    const long total = (long)D * H * W;
    int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (!(occ[i] > thr)) continue;
        const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)(i / ((long)W * H));
        // mask_xyz = (w,h,d) * 2 (SparseConvNet.py:140-141); pts = mask_xyz * voxel_size + bounds_min; world = pts @ R^T + Th
        const float sx = (float)w * 2.f * g.voxel[0] + g.bmin[0], sy = (float)h * 2.f * g.voxel[1] + g.bmin[1],
                    sz = (float)d * 2.f * g.voxel[2] + g.bmin[2];
        float p[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = mm3(sx, g.Rh[a * 3 + 0], sy, g.Rh[a * 3 + 1], sz, g.Rh[a * 3 + 2]) + g.Th[a];
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], ordered_int(p[a])); hi[a] = max(hi[a], ordered_int(p[a])); }
        float c[3], q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a] = mm3(p[0], g.pose[a * 4 + 0], p[1], g.pose[a * 4 + 1], p[2], g.pose[a * 4 + 2]) + g.pose[a * 4 + 3];
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = mm3(c[0], g.K[a * 3 + 0], c[1], g.K[a * 3 + 1], c[2], g.K[a * 3 + 2]);
        const float fx = q[0] / q[2], fy = q[1] / q[2];
        if (!(fabsf(fx) < 1e9f) || !(fabsf(fy) < 1e9f)) continue;          // .long() of inf/nan is undefined in the reference
        int x0 = (int)fx, y0 = (int)fy;                                     // .long(): truncation toward zero
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = min(max(x0, 0), iw - 1); x1 = min(max(x1, 0), iw - 1);        // the reference clamps to its literal W = 512
        y0 = min(max(y0, 0), ih - 1); y1 = min(max(y1, 0), ih - 1);
        // several voxels project onto every marked pixel: look before storing
        uint8_t* const t[4] = {sel + y0 * iw + x0, sel + y1 * iw + x0, sel + y0 * iw + x1, sel + y1 * iw + x1};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (!*reinterpret_cast<volatile uint8_t*>(t[k])) *t[k] = 1;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], __shfl_xor(lo[a], o)); hi[a] = max(hi[a], __shfl_xor(hi[a], o)); }
        if ((threadIdx.x & 63) == 0 && lo[a] <= hi[a]) { atomicMin(mm + a, lo[a]); atomicMax(mm + 3 + a, hi[a]); }
    }
}

// ---------------------------------------------------------------------------------------------
// channels-last re-layouts: [C=32][P] -> [P][32] through a padded LDS tile (coalesced both ways)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cfirst_to_clast32_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                const long P, const long src_img_stride,
                                                                const long dst_img_stride) {
    __shared__ float tile[32][65];
    const float* s = src + (size_t)blockIdx.y * src_img_stride;
    float* d = dst + (size_t)blockIdx.y * dst_img_stride;
    const long p0 = (long)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
#pragma unroll
    for (int c = ty; c < 32; c += 4) {
        const long p = p0 + tx;
        tile[c][tx] = (p < P) ? s[(size_t)c * P + p] : 0.f;
    }
    __syncthreads();
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
    for (int pp = q; pp < 64; pp += 8) {
        const long p = p0 + pp;
        if (p < P) d[(size_t)p * 32 + c] = tile[c][pp];
    }
}

__global__ void init_minmax_kernel(int* __restrict__ mm) {
    if (threadIdx.x < 6) mm[threadIdx.x] = threadIdx.x < 3 ? 0x7FFFFFFF : (int)0x80000000;
}

// SparseConvNet.encode's masks3d (SparseConvNet.py:135-139): one lane per level-1 voxel
__global__ void occupancy_kernel(const FrameK fr, float* __restrict__ occ) {
    const int D = fr.vol_dhw[0][0], H = fr.vol_dhw[0][1], W = fr.vol_dhw[0][2];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)D * H * W) return;
    const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)(i / ((long)W * H));
    float total = 0.f;
#pragma unroll
    for (int l = 0; l < GPNERF_LEVELS; ++l) {
        const int Dl = fr.vol_dhw[l][0], Hl = fr.vol_dhw[l][1], Wl = fr.vol_dhw[l][2];
        // F.interpolate(mode='nearest'): src = floor(dst * in / out)
        const int dl = min((int)floorf((float)d * ((float)Dl / (float)D)), Dl - 1);
        const int hl = min((int)floorf((float)h * ((float)Hl / (float)H)), Hl - 1);
        const int wl = min((int)floorf((float)w * ((float)Wl / (float)W)), Wl - 1);
        const f32x4* p = reinterpret_cast<const f32x4*>(fr.vol[l] + (((size_t)dl * Hl + hl) * Wl + wl) * 32);
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const f32x4 v = p[q]; sum += (v[0] + v[1]) + (v[2] + v[3]); }
        total += sum;
    }
    occ[i] = total;
}

__global__ void images_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, const long HW, const int V) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW * V) return;
    const long v = i / HW, p = i % HW;
    const float* s = src + (size_t)v * 3 * HW + p;
    f32x4 o;
    o[0] = s[0] * 0.5f + 0.5f; o[1] = s[HW] * 0.5f + 0.5f; o[2] = s[2 * HW] * 0.5f + 0.5f; o[3] = 0.f;   // BaseRender.py:231
    *reinterpret_cast<f32x4*>(dst + (size_t)i * 4) = o;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
namespace {
constexpr int FUSED_WAVES = 8;      // waves per workgroup of head_forward_kernel

int col_ok(int c, int n_in) { return c >= 0 && c < n_in; }

// f16(x) rounded toward zero (what v_cvt_pkrtz_f16_f32 does on the device), on the host
_Float16 f16_rtz(float v) {
    _Float16 h = (_Float16)v;                                  // round to nearest
    if (v == v && (float)h != v && ((float)h > v) == (v > 0.f)) {   // rounded away from zero: step one ulp back
        uint16_t b;
        memcpy(&b, &h, 2);
        b -= 1;                                                // magnitude bits decrease toward zero for either sign
        memcpy(&h, &b, 2);
    }
    return h;
}

// scaled-domain packing (see elus()): factor on weight column c of MFMA layer L -- log2(e) where the column multiplies a
// raw input (volume features, cross-view mean / variance, per-view features), 1 where it multiplies a scaled activation
constexpr float PACK_LOG2E = 1.4426950408889634f, PACK_LN2 = 0.6931471805599453f;
float pack_scale(int L, int c) {
    switch (L) {
        case gpl::GEO: case gpl::BS: case gpl::BV: return PACK_LOG2E;
        case gpl::D1: return c >= 64 ? PACK_LOG2E : 1.f;
        case gpl::V1: return 1.f / 3.f;       // vis_fc(x * 1.0 / num_views) (trainhead.py:140): the 1/V rides on the weights
        default: return 1.f;
    }
}

void pack_layer(int L, const float* W, const float* b, int n_out, int n_in, float* blob) {
    for (int m = 0; m < gpl::MT[L]; ++m) {
        float* wt = blob + gpl::w_off(L) + m * gpl::NT[L] * 64;
        const int NT = gpl::NT[L], NG = NT / 4;
        for (int t = 0; t < NT; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = 32 * m + (lane & 31), h = lane >> 5;
                const int c = gpl::col_of(L, t, h);
                const float v = (row < n_out && col_ok(c, n_in)) ? W[(size_t)row * n_in + c] * pack_scale(L, c) : 0.f;
                const int g = t / 4;
                if (g < NG) wt[(g * 64 + lane) * 4 + (t & 3)] = v;
                else wt[NG * 256 + lane * 2 + (t - 4 * NG)] = v;
            }
        float* bt = blob + gpl::b_off(L) + m * 32;
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + gpl::ft(r, h);
                bt[h * 16 + r] = (b && row < n_out) ? b[row] * PACK_LOG2E : 0.f;
            }
    }
}

// reference-order image (head_layout.h gpr): tile row i carries output feature gpr::feat_of_row(i), k-step t the columns
// (2t, 2t+1) (gpr::col_of), nothing is scaled, the bias tile holds feature 2r + h at [h][r]
void pack_layer_ref(int L, const float* W, const float* b, int n_out, int n_in, float* blob) {
    for (int m = 0; m < gpl::MT[L]; ++m) {
        float* wt = blob + gpl::w_off(L) + m * gpl::NT[L] * 64;
        const int NT = gpl::NT[L], NG = NT / 4;
        for (int t = 0; t < NT; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = 32 * m + gpr::feat_of_row(lane & 31), h = lane >> 5;
                const int c = gpr::col_of(L, t, h);
                const float v = (row < n_out && col_ok(c, n_in)) ? W[(size_t)row * n_in + c] : 0.f;
                const int g = t / 4;
                if (g < NG) wt[(g * 64 + lane) * 4 + (t & 3)] = v;
                else wt[NG * 256 + lane * 2 + (t - 4 * NG)] = v;
            }
        float* bt = blob + gpl::b_off(L) + m * 32;
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + 2 * r + h;
                bt[h * 16 + r] = (b && row < n_out) ? b[row] : 0.f;
            }
    }
}

// Launch geometry of the fused kernel.  One workgroup is resident per CU (LDS), a wave's step time depends on how many
// waves share its SIMD (measured: ~86k cycles per 32-sample step with one wave per SIMD, ~134k with two), and a work unit
// (32 rays x S samples) is long, so a grid that is not many times 256 workgroups quantises badly.  Choose the waves per
// workgroup and, when the caller lends a workspace, how many waves share the samples of one tile (split), minimising
// rounds x step time x samples per unit.  GPNERF_WAVES / GPNERF_SPLIT override (diagnostics).
constexpr int GPNERF_MAX_SPLIT = 8;     // waves that may share one tile's samples
constexpr size_t QUEUE_BYTES = 256;     // head of the workspace: 8 tile-queue counters (one per XCD), padded
// frame-level deferral (colour_units_kernel): per launch slot its entry count, and room for an entry and a result per SAMPLE (a
// frame whose every weight is non-zero); frames beyond 2^26 samples (1024 x 1024 x 64: 2.1 GB) keep the tile-level passes
size_t align256(size_t v);
bool gdef_fits(int64_t n_rays, int32_t n_samples) { return n_samples <= 256 && n_rays * (int64_t)n_samples <= ((int64_t)1 << 26); }
constexpr size_t GDEF_HEAD_BYTES = 256;        // gd_ctrl: the unit queue's counters, the entry / unit count, the tickets, the wavefronts that are done listing
// entries: one per sample + a unit per VISIT of a work unit (the unified form pads what a visit leaves to a whole unit; a 32-ray tile
// is one visit, or up to eight where it runs as units of several samples per step); flags: one word per unit
size_t gdef_entries(int64_t n_rays, int32_t n_samples) { return (size_t)n_rays * n_samples + ((size_t)((n_rays + 31) / 32) * 8 + 64) * 32; }
size_t gdef_flag_bytes(int64_t n_rays, int32_t n_samples) { return align256((gdef_entries(n_rays, n_samples) / 32 + 64) * sizeof(unsigned)); }
size_t gdef_bytes(int64_t n_rays, int32_t n_samples) {
    return GDEF_HEAD_BYTES + align256((size_t)n_rays * sizeof(int)) + gdef_flag_bytes(n_rays, n_samples) + gdef_entries(n_rays, n_samples) * sizeof(uint4) +
           (size_t)n_rays * n_samples * sizeof(f32x4);
}
// Early termination walks the samples in segments of chain_len(), one launch per segment over the rays still alive (see
// gpnerf_render_fused); the workspace then holds a control block (per segment: 8 queue counters + the length of its output
// list), two ray lists (written and read alternately) and 16 floats of parked state per ray.
// Experiment knobs (dbg_int / dbg_env, tools/*.sh A/B runs): hook points of gpnerf_diag.h.  The product's version returns the
// default, always -- it has no getenv; the diagnostic libraries of csrc/diag/ read the environment under GPNERF_DEBUG=1, clamped.
// One launch of the fused kernel: which arithmetic (`sel`), whether the colour branch is deferred sample by sample (render_tile), and
// the sample loop (chained segments / culled) as template arguments.  Dynamic LDS = the form's head image (+ the split form's guard
// slots) + the wavefronts' colour queues.
enum { SEL_REF = 0, SEL_FOLD = 1, SEL_SPLIT = 2, SEL_GUARD = 3 };
// Every form defers the colour branch (round 6; rounds 1-5 kept it in the step for the split-precision forms).  Round 5 built the
// split forms' deferral (bench frame 7.3 -> 6.4 ms) and did not ship it: ONE build's unguarded instantiation gave colour passes
// 10-30 % off, the same wrong values on every box, cured by any change that moved the schedule (an opaque copy of the regathered
// inputs, `volatile` on lo_pair's asm, lo_pair written without asm) and not by waits around the pass.  The mechanism class is
// now demonstrated on the hardware (tools/micro/asm_producer_hazards.hip, profiles/r06/i_asm_producer_hazards.txt): gfx950 does
// not interlock a VALU write of a VGPR with an MFMA that reads it as a source operand in the next issue slot -- the MFMA gets the
// register's previous contents, in every lane alike -- LLVM pads the producers it can see, and lo_pair's v_fma_mixlo/hi_f16 are
// inline assembly it cannot: whether one lands directly in front of its MFMA is the scheduler's accident, which is exactly how
// the failure came and went with unrelated edits.  (That build itself can no longer be reproduced -- revision c9a579c without the
// opaque copy compiles to a schedule with one instruction in between and is bit-exact today, profiles/r06/i_split_defer_rebuild.txt
// -- so the instance stays inferred; the class is measured.)  Since round 6 every fragment that becomes an MFMA operand passes
// through settle_operand(), which carries the wait state itself, and tools/isa_mfma_hazards.py fails the CPU suite on any
// inline-asm producer closer to its MFMA than the measured requirement (tests/test_abi.py).
constexpr bool SPLIT_DEFERS = true;
template <int FORM> constexpr bool form_defers() { return SPLIT_DEFERS || (FORM != FORM_SPLIT && FORM != FORM_SPLIT_GUARD); }
template <int FORM, bool CHAIN, bool CULL>
void launch_form(bool deferred, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const KArgs& ka) {
    if constexpr ((FORM == FORM_F32 || FORM == FORM_F32_FOLD) && !CULL) {
        // (the unified form's wavefronts wait for each other -- every one reports before any leaves -- which assumes the grid becomes
        //  resident without depending on another tenant that waits the same way: GPNERF_FLAG_SHARED_DEVICE selects the second kernel.
        //  As a cooperative launch, which guarantees residency, it cost 0.04 ms per call and, with the next frame's producers on a
        //  second stream, the whole overlap of the pipelined evaluation loop: 7.0 -> 8.5 ms per frame)
        if (deferred && ka.gd_ent && ka.gd_flag) { hipLaunchKernelGGL((render_fused_kernel<FORM, CHAIN, false, true, true, true>), grid, block, lds, stream, ka); return; }
        if (deferred && ka.gd_ent) { hipLaunchKernelGGL((render_fused_kernel<FORM, CHAIN, false, true, true>), grid, block, lds, stream, ka); return; }
    }
    if constexpr (form_defers<FORM>()) {
        if (deferred) { hipLaunchKernelGGL((render_fused_kernel<FORM, CHAIN, CULL, true>), grid, block, lds, stream, ka); return; }
    }
    hipLaunchKernelGGL((render_fused_kernel<FORM, CHAIN, CULL, false>), grid, block, lds, stream, ka);
}
template <bool CHAIN, bool CULL>
void launch_render(int sel, bool deferred, dim3 grid, dim3 block, hipStream_t stream, const KArgs& ka) {
    const size_t f32 = sizeof(float) * gpl::BLOB_FLOATS + DEFER_LDS_BYTES, split = sizeof(unsigned) * gph::BLOB_WORDS + DEFER_LDS_BYTES;
    switch (sel) {
        case SEL_GUARD: launch_form<FORM_SPLIT_GUARD, CHAIN, CULL>(deferred, grid, block, split + GUARD_LDS_SLOTS * 8, stream, ka); break;
        case SEL_SPLIT: launch_form<FORM_SPLIT, CHAIN, CULL>(deferred, grid, block, split, stream, ka); break;
        case SEL_FOLD:  launch_form<FORM_F32_FOLD, CHAIN, CULL>(deferred, grid, block, f32, stream, ka); break;
        default:        launch_form<FORM_F32, CHAIN, CULL>(deferred, grid, block, f32, stream, ka); break;
    }
}

constexpr int CHAIN_SEG = 16;
constexpr int CHAIN_MAX_SEGS = 64;
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
int chain_len(int S) {
    static int f_seg = -1;
    if (f_seg < 0) f_seg = dbg_int("GPNERF_CHAIN_SEG", CHAIN_SEG, 1, 256);
    const int least = (S + CHAIN_MAX_SEGS - 1) / CHAIN_MAX_SEGS;
    return f_seg > least ? f_seg : least;
}
// GPNERF_CHAIN_SCHEDULE="16,8,8,8,16" (experiment knob, under GPNERF_DEBUG=1): the launches' segment lengths, the last one repeated
// to the end of the ray; at most CHAIN_MAX_SEGS entries.  Empty (default): equal segments.
const int* chain_custom(int* n) {
    static int f_n = -1, f_len[CHAIN_MAX_SEGS];
    if (f_n < 0) {
        f_n = 0;
        const char* e = dbg_env("GPNERF_CHAIN_SCHEDULE");
        while (e && *e && f_n < CHAIN_MAX_SEGS) {
            const int v = atoi(e);
            if (v < 1) break;
            f_len[f_n++] = v > 256 ? 256 : v;
            while (*e && *e != ',') ++e;
            if (*e == ',') ++e;
        }
    }
    *n = f_n;
    return f_len;
}
// The launches' sample ranges.  Default: three segments of chain_len(), one twice as long, then segments three times as long.  GPNERF_CHAIN_MERGE_AFTER=k (experiment knob, under
// GPNERF_DEBUG=1): from sample k on every segment is as long as all the samples before it ([0,16) .. [48,64), [64,128), ...).
// Measured on the 512x512x128 bench frame, where 5 % of the rays are alive after 64 samples: 8.82 ms with equal segments,
// 8.89 ms with the four tail launches merged into one -- the tail levels already run several samples of a ray per step
// (chain_plan's P), so their launches cost little, and inside a long segment an opaque ray idles to the segment's end.
// begins[] gets n + 1 entries; returns n <= chain_segs(S).
int chain_schedule(int S, int* begins) {
    static int f_merge = -1;
    if (f_merge < 0) f_merge = dbg_int("GPNERF_CHAIN_MERGE_AFTER", 0, 0, 1 << 20);       // 0 (default): equal segments throughout
    int nc = 0;
    const int* custom = chain_custom(&nc);
    const int len = chain_len(S);
    int n = 0, k = 0, cur = len;
    while (k < S && n < CHAIN_MAX_SEGS) {
        begins[n] = k;
        if (nc > 0) cur = custom[n < nc ? n : nc - 1];
        else if (f_merge > 0 && k >= f_merge) { cur = k; }            // from here on every segment is as long as all before it
        else if (f_merge == 0 && n >= 3) cur = n == 3 ? 2 * len : 3 * len;   // default: three segments of `len`, one twice, then three
                                                                      // times as long (512x512x128: 16,16,16,32,48: the few rays
                                                                      // alive past sample 48 need fewer, fuller launches.  Round 5
                                                                      // had 16,16,16,16,32,32; with the colour work out of the
                                                                      // segment launches, three boxes: 6.93 -> 6.86 ms, with the
                                                                      // two-sample tail units below 6.84)
        ++n;
        k += cur;
        if (n == CHAIN_MAX_SEGS - 1 && k < S) { begins[n++] = k; k = S; }     // (the schedule ran out of launches: one last segment to the end)
    }
    begins[n] = S;
    return n;
}
int chain_segs(int S) {               // upper bound on the launches (workspace sizing)
    int begins[CHAIN_MAX_SEGS + 2];
    const int n = chain_schedule(S, begins), len = chain_len(S), eq = (S + len - 1) / len;
    return n > eq ? n : eq;
}
size_t chain_chunks(int64_t n_rays) { return (size_t)((n_rays + 2047) / 2048); }       // LIST_CHUNK entries each
// control block: per segment 8 queue counters, the length of its output list, and one survivor counter per LIST_CHUNK input entries
size_t chain_ctrl_bytes(int n_seg, int64_t n_rays) { return align256((size_t)n_seg * (9 + chain_chunks(n_rays)) * sizeof(unsigned)); }
// range guard of the split form: header (flag count, the fix-up launch's queue counters) + one word per tile, at the workspace's end
size_t guard_bytes(int64_t n_rays) {
    return align256((GUARD_HEADER_WORDS + (size_t)((n_rays + RAYS_PER_WAVE - 1) / RAYS_PER_WAVE)) * sizeof(unsigned));
}
// occupancy culling: keep bits per ray, then per tile its step count and its place in the longest-first order
size_t cull_tiles(int64_t n_rays) { return (size_t)((n_rays + RAYS_PER_WAVE - 1) / RAYS_PER_WAVE); }
size_t cull_mask_bytes(int64_t n_rays) {
    return align256((size_t)n_rays * 2 * sizeof(unsigned long long)) + 2 * align256(cull_tiles(n_rays) * sizeof(int));
}
size_t chain_bytes(int64_t n_rays, int S) {
    if (chain_segs(S) < 2 || n_rays >= ((int64_t)1 << 31)) return 0;
    return chain_ctrl_bytes(chain_segs(S), n_rays) + 3 * align256((size_t)n_rays * sizeof(int)) + (size_t)n_rays * 16 * sizeof(float);
}
struct Geometry { int waves, split; };

Geometry choose_geometry(int64_t tiles, int S, bool may_split, size_t ws_bytes, int64_t n_rays, int n_cus) {
    static int f_waves = -1, f_split = -1;
    if (f_waves < 0) {
        f_waves = dbg_int("GPNERF_WAVES", 0, 0, 8);
        f_split = dbg_int("GPNERF_SPLIT", 0, 0, 8);
    }
    const int64_t cus = n_cus > 0 ? n_cus : 256;
    Geometry best{8, 1};
    double best_t = 1e300;
    for (int split = 1; split <= GPNERF_MAX_SPLIT; split *= 2) {
        if (split > 1 && (!may_split || S / split < 8 || ws_bytes < (size_t)n_rays * split * 16 * sizeof(float))) continue;
        if (f_split > 0 && split != f_split && !(split == 1 && f_split > 1 && !may_split)) continue;
        for (int w = GPNERF_MAX_WAVES; w >= 1; --w) {
            if (f_waves > 0 && w != f_waves) continue;
            const int64_t blocks = (tiles * split + w - 1) / w;
            const int64_t rounds = (blocks + cus - 1) / cus;
            const double step = w <= 4 ? 82.0 : (w <= 8 ? 133.0 : 190.0);  // kilo-cycles per 32-sample step (measured)
            const double t = (double)rounds * step * ((double)S / split) + (split > 1 ? 60.0 : 0.0);
            if (t < best_t * 0.97) { best_t = t; best = Geometry{w, split}; }   // ties: wider workgroup, no split
        }
    }
    return best;
}

hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
// The counters a launch sequence starts from (tile queues, list lengths, the guard's flags) are zeroed by a kernel of the library's
// own, not by hipMemsetAsync: captured into a HIP graph, the memset node was not reliably ordered before the kernel node that
// follows it (replays after the first found the previous replay's counters -- exhausted queues -- and rendered nothing;
// tests/test_gpu_guard.py, tools/graph_probe2.py), kernel after kernel is.
__global__ void zero_words_kernel(unsigned* __restrict__ p, const long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ void zero_bytes_kernel(uint8_t* __restrict__ p, const long n) {      // p: 4-byte aligned
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) *reinterpret_cast<unsigned*>(p + i) = 0u;
        else for (long j = i; j < n; ++j) p[j] = 0;
    }
}
bool zero_async(void* p, size_t bytes, void* stream) {          // bytes: a multiple of 4
    const long n = (long)(bytes / sizeof(unsigned));
    const long wgs = (n + 255) / 256;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(wgs < 1024 ? (wgs > 0 ? wgs : 1) : 1024)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<unsigned*>(p), n);
    return hipGetLastError() == hipSuccess;
}

// Per-device facts and one-time setup, keyed by the CURRENT device of the calling thread: the architecture check
// (GPNERF_E_DEVICE on anything but gfx950), the CU count the launch geometry balances over, and the opt-in to > 64 KB of
// dynamic LDS, which is a per-device function attribute (a process that renders on a second GPU needs it there too).
struct DeviceState { bool probed = false, ok = false; int cus = 0; };
constexpr int MAX_DEVICES = 64;
DeviceState g_devices[MAX_DEVICES];
std::mutex g_devices_mutex;


// returns GPNERF_OK and fills *cus, or GPNERF_E_DEVICE
int device_ready(int* cus) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return GPNERF_E_DEVICE;
    std::lock_guard<std::mutex> lock(g_devices_mutex);
    DeviceState& d = g_devices[dev];
    if (!d.probed) {
        d.probed = true;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) {
            d.cus = prop.multiProcessorCount;
            const size_t lds_bytes = sizeof(float) * gpl::BLOB_FLOATS + DEFER_LDS_BYTES, lds_split = sizeof(unsigned) * gph::BLOB_WORDS;
            auto lds_ok = [](const void* fn, size_t bytes) {
                return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
            };
            d.ok = true;
            // every instantiation the launches below can pick: form x (plain, chained, culled) x (colour branch in the step, deferred)
            auto all_of = [&](auto form_tag, size_t bytes) {
                constexpr int F = decltype(form_tag)::value;
                d.ok = d.ok && lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<F, false, false, false>), bytes) &&
                       lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<F, true, false, false>), bytes) &&
                       lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<F, false, true, false>), bytes);
                if constexpr (form_defers<F>())
                    d.ok = d.ok && lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<F, false, false, true>), bytes) &&
                           lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<F, true, false, true>), bytes) &&
                           lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<F, false, true, true>), bytes);
            };
            all_of(std::integral_constant<int, FORM_F32>{}, lds_bytes);
            all_of(std::integral_constant<int, FORM_F32_FOLD>{}, lds_bytes);
            // frame-level deferral (fp32 forms)
            d.ok = d.ok && lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32, false, false, true, true>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32_FOLD, false, false, true, true>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32, false, false, true, true, true>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32_FOLD, false, false, true, true, true>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32, true, false, true, true, true>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32_FOLD, true, false, true, true, true>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32, true, false, true, true>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32_FOLD, true, false, true, true>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&colour_units_kernel<FORM_F32>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&colour_units_kernel<FORM_F32_FOLD>), lds_bytes);
            all_of(std::integral_constant<int, FORM_SPLIT>{}, lds_split + DEFER_LDS_BYTES);
            all_of(std::integral_constant<int, FORM_SPLIT_GUARD>{}, lds_split + GUARD_LDS_SLOTS * 8 + DEFER_LDS_BYTES);
            d.ok = d.ok && lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32_FIXUP, false>), lds_bytes) &&
                   lds_ok(reinterpret_cast<const void*>(&render_fused_kernel<FORM_F32_FIXUP, false, true>), lds_bytes) &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&head_forward_kernel<FUSED_WAVES, 0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&head_forward_kernel<FUSED_WAVES, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&head_forward_kernel<FUSED_WAVES, 2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess;
        }
        (void)hipGetLastError();
    }
    if (!d.ok) return GPNERF_E_DEVICE;
    if (cus) *cus = d.cus;
    return GPNERF_OK;
}

int launch_status() { return hipGetLastError() == hipSuccess ? GPNERF_OK : GPNERF_E_LAUNCH; }

// GpnerfFrame -> kernel argument; need_vol / need_img say which tensors the launch will touch
bool to_framek(const GpnerfFrame* f, FrameK& k, bool need_vol, bool need_img) {
    memset(&k, 0, sizeof(k));
    // tap addresses are 32-bit byte offsets built from 24-bit factors inside the kernels (gather_volume / gather_view)
    const int64_t lim_bytes = (int64_t)1 << 32, lim24 = (int64_t)1 << 24;
    for (int l = 0; l < GPNERF_LEVELS; ++l) {
        if (!f->vol[l]) continue;
        const int64_t D = f->vol_dhw[l][0], H = f->vol_dhw[l][1], W = f->vol_dhw[l][2];
        if (D * H * W * GPNERF_CH * 4 >= lim_bytes || D * H >= lim24 || W * GPNERF_CH * 4 >= lim24) return false;
    }
    if ((int64_t)f->img_h * f->img_w * 16 >= lim_bytes || (int64_t)f->img_w * 16 >= lim24 || f->img_h >= lim24) return false;
    if ((int64_t)f->feat_h * f->feat_w * GPNERF_CH * 4 >= lim_bytes || (int64_t)f->feat_w * GPNERF_CH * 4 >= lim24 || f->feat_h >= lim24) return false;
    if (need_vol)
        for (int l = 0; l < GPNERF_LEVELS; ++l)
            if (!f->vol[l] || f->vol_dhw[l][0] < 1 || f->vol_dhw[l][1] < 1 || f->vol_dhw[l][2] < 1) return false;
    if (need_img && (!f->featmaps || !f->imgs || f->feat_h < 1 || f->feat_w < 1 || f->img_h < 1 || f->img_w < 1)) return false;
    for (int l = 0; l < GPNERF_LEVELS; ++l) {
        k.vol[l] = f->vol[l];
        for (int a = 0; a < 3; ++a) k.vol_dhw[l][a] = f->vol_dhw[l][a];
    }
    k.featmaps = f->featmaps; k.feat_h = f->feat_h; k.feat_w = f->feat_w;
    k.imgs = f->imgs; k.img_h = f->img_h; k.img_w = f->img_w;
    memcpy(k.proj, f->proj, sizeof(k.proj));
    memcpy(k.Rh, f->Rh, sizeof(k.Rh));
    memcpy(k.Th, f->Th, sizeof(k.Th));
    memcpy(k.bounds_min, f->bounds_min, sizeof(k.bounds_min));
    memcpy(k.voxel, f->voxel, sizeof(k.voxel));
    for (int a = 0; a < 3; ++a) k.out_sh[a] = (float)f->out_sh[a];
    k.head_blob = f->head_blob;
    k.head_blob_split = f->head_blob_split;
    k.head_blob_ref = f->head_blob_ref;
    k.occ = f->occ;
    // folded volumes: all four levels or none; 64 values per voxel must still be addressable with 32-bit byte offsets
    bool fold = true;
    for (int l = GPNERF_FOLD_FIRST_LEVEL; l < GPNERF_LEVELS; ++l) {
        const int64_t D = f->vol_dhw[l][0], H = f->vol_dhw[l][1], W = f->vol_dhw[l][2];
        fold = fold && f->vol_folded[l] && D * H * W * GPNERF_CH * 8 < lim_bytes && W * GPNERF_CH * 8 < lim24;
    }
    for (int l = GPNERF_FOLD_FIRST_LEVEL; l < GPNERF_LEVELS; ++l) k.vol_fold[l] = fold ? f->vol_folded[l] : nullptr;
    return true;
}

OutK to_outk(const GpnerfOutputs* o, const int32_t* order = nullptr) {
    OutK k;
    k.order = order;
    k.rgb = o->rgb; k.depth = o->depth; k.acc = o->acc; k.disp = o->disp; k.weights = o->weights;
    k.z_vals = o->z_vals; k.rgb_in = o->rgb_in; k.raw = o->raw; k.ray_mask = o->ray_mask; k.samples_done = o->samples_done; k.step_stats = o->step_stats;
    return k;
}

template <int MODE>
int launch_head(const float* head_blob, const float* a, const float* rgb_feat, const float* mask, int64_t n_points, float* out,
                float* out2, void* stream) {
    if (device_ready(nullptr) != GPNERF_OK) return GPNERF_E_DEVICE;
    const size_t lds_bytes = sizeof(float) * gpl::BLOB_FLOATS;
    const int64_t tiles = (n_points + 31) / 32;
    int64_t blocks = (tiles + FUSED_WAVES - 1) / FUSED_WAVES;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL((head_forward_kernel<FUSED_WAVES, MODE>), dim3((unsigned)blocks), dim3(FUSED_WAVES * 64), lds_bytes, S_(stream),
                       head_blob, a, rgb_feat, mask, (long)n_points, out, out2);
    return launch_status();
}

// ---- patch-major order of a frame's kept pixels (frame.patch_order on the device, three launches) --------------------------------
// order[q] = raster rank (among the kept pixels) of the q-th kept pixel when the image is walked patch by patch (patch_w x patch_h
// pixels, patches in raster order, pixels inside a patch in raster order).  A workgroup owns a band of patch_h image rows, one
// wavefront walks a row 64 pixels at a time with ballots (prefix inside the row = popcount below the lane + the row's running sum):
//   1. po_count_kernel: kept pixels per image row and per patch;
//   2. po_scan_kernel (one workgroup): both count arrays -> exclusive prefixes; the grand total is compared with the caller's n;
//   3. po_scatter_kernel: rank = row base + prefix in the row; position = patch base + kept pixels of the patch's earlier rows +
//      prefix in the patch's row; order[position] = rank.  A mask that does not keep exactly n pixels: order = 0 .. n - 1.
// scratch (int32): rows[H] | patches[npy * npx] | flag[1]
constexpr int PO_MAX_PH = 32;
template <bool SCATTER>
__global__ void __launch_bounds__(256) po_band_kernel(const uint8_t* __restrict__ mask, const int H, const int W, const int pw, const int ph,
                                                      int* __restrict__ rows, int* __restrict__ patches, const int n, int* __restrict__ order) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, band = blockIdx.x, npx = (W + pw - 1) / pw;
    const int* const flag = patches + (size_t)gridDim.x * npx;
    if (SCATTER && *flag != n) {                         // uniform: the mask does not match the ray list
        for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < n; i += (int)gridDim.x * 256) order[i] = i;
        return;
    }
    if (!SCATTER) {
        // rows of the band dealt to the four wavefronts; a row's total, and its patches' totals added up by atomics in LDS-free form:
        // every patch of the band is touched by ph rows, so the patch counts go through global atomics on zeroed words (integer adds
        // commute: deterministic)
        for (int yy = wave; yy < ph; yy += 4) {
            const int y = band * ph + yy;
            if (y >= H) break;
            int run = 0;
            for (int x0 = 0; x0 < W; x0 += 64) {
                const int x = x0 + lane;
                const bool m = x < W && mask[(size_t)y * W + x] != 0;
                const unsigned long long b = __ballot(m);
                // the 64 pixels span at most 64 / pw + 1 patches: lanes that START a patch segment add that segment's count
                const int k = x / pw, xe = min(min((k + 1) * pw, W), x0 + 64);          // end of this lane's segment inside the window
                if (x < W && (x % pw == 0 || lane == 0)) {
                    const unsigned long long seg = (xe - x >= 64 ? ~0ull : ((1ull << (xe - x)) - 1ull)) << lane;
                    const int c = __popcll(b & seg);
                    if (c) atomicAdd(patches + (size_t)band * npx + k, c);
                }
                run += __popcll(b);
            }
            if (lane == 0) rows[y] = run;
        }
        return;
    } else {
        // patch by patch would leave most lanes idle for narrow patches; instead row by row, with the earlier rows' counts of every
        // patch kept in LDS: pass A counts (patch, row) pairs, pass B scatters
        extern __shared__ int pr[];                      // [ph][npx] kept pixels of patch k in band row yy
        for (int i = threadIdx.x; i < ph * npx; i += 256) pr[i] = 0;
        __syncthreads();
        for (int yy = wave; yy < ph; yy += 4) {
            const int y = band * ph + yy;
            if (y >= H) break;
            for (int x0 = 0; x0 < W; x0 += 64) {
                const int x = x0 + lane;
                const bool m = x < W && mask[(size_t)y * W + x] != 0;
                const unsigned long long b = __ballot(m);
                const int k = x / pw, xe = min(min((k + 1) * pw, W), x0 + 64);
                if (x < W && (x % pw == 0 || lane == 0)) {
                    const unsigned long long seg = (xe - x >= 64 ? ~0ull : ((1ull << (xe - x)) - 1ull)) << lane;
                    const int c = __popcll(b & seg);
                    if (c) atomicAdd(pr + yy * npx + k, c);
                }
            }
        }
        __syncthreads();
        for (int yy = wave; yy < ph; yy += 4) {
            const int y = band * ph + yy;
            if (y >= H) break;
            int run = rows[y];                           // raster rank of the row's first kept pixel
            for (int x0 = 0; x0 < W; x0 += 64) {
                const int x = x0 + lane;
                const bool m = x < W && mask[(size_t)y * W + x] != 0;
                const unsigned long long b = __ballot(m);
                if (m) {
                    const int k = x / pw;
                    int pos = patches[(size_t)band * npx + k];                        // kept pixels in earlier patches
                    for (int r = 0; r < yy; ++r) pos += pr[r * npx + k];              // ... in this patch's earlier rows
                    // ... in this row of the patch before x: pixels of the patch's row segment left of x, counted in this window and,
                    // when the segment started in an earlier window, in those (pw <= 64 keeps that to one earlier window at most)
                    const int xs = k * pw;
                    int before;
                    if (xs >= x0) before = __popcll(b & (((1ull << lane) - 1ull) & ~((1ull << (xs - x0)) - 1ull)));
                    else {
                        before = __popcll(b & ((1ull << lane) - 1ull));
                        for (int xx = xs; xx < x0; ++xx) before += mask[(size_t)y * W + xx] != 0;
                    }
                    order[pos + before] = run + __popcll(b & ((1ull << lane) - 1ull));
                }
                run += __popcll(b);
            }
        }
    }
}
__global__ void __launch_bounds__(1024) po_scan_kernel(int* __restrict__ rows, const int H, int* __restrict__ patches, const int NP) {
    // exclusive prefixes of both arrays in place, one after the other; patches[NP] receives the grand total
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int pass = 0; pass < 2; ++pass) {
        int* a = pass ? patches : rows;
        const int n = pass ? NP : H;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (int base = 0; base < n; base += 1024) {
            const int i = base + (int)threadIdx.x;
            const int v = i < n ? a[i] : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            int woff = carry;
            for (int w = 0; w < wave; ++w) woff += wsum[w];
            if (i < n) a[i] = woff + inc - v;
            __syncthreads();
            if (threadIdx.x == 1023) carry = woff + inc;
            __syncthreads();
        }
        if (pass && threadIdx.x == 0) patches[NP] = carry;
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int64_t gpnerf_head_blob_floats(void) { return gpl::BLOB_FLOATS; }
int32_t gpnerf_rays_per_tile(void) { return RAYS_PER_WAVE; }
const char* gpnerf_build_info(void) { return "gpnerf-hip gfx950 fp32-mfma32x32x2 waves<=8"; }

GPNERF_DIAG_EXPORTS        /* nothing in the product (csrc/nodiag/gpnerf_diag.h) */

int gpnerf_head_layout(int32_t* table) {
    if (!table) return GPNERF_E_ARG;
    for (int l = 0; l < gpl::NLAYER; ++l) {
        table[4 * l + 0] = gpl::NT[l]; table[4 * l + 1] = gpl::MT[l];
    }
    table[4 * gpl::GEO + 2] = gpl::w_off(gpl::GEO); table[4 * gpl::GEO + 3] = gpl::b_off(gpl::GEO);
    table[4 * gpl::D1 + 2] = gpl::w_off(gpl::D1);   table[4 * gpl::D1 + 3] = gpl::b_off(gpl::D1);
    table[4 * gpl::D2 + 2] = gpl::w_off(gpl::D2);   table[4 * gpl::D2 + 3] = gpl::b_off(gpl::D2);
    table[4 * gpl::D3 + 2] = gpl::w_off(gpl::D3);   table[4 * gpl::D3 + 3] = gpl::b_off(gpl::D3);
    table[4 * gpl::BS + 2] = gpl::w_off(gpl::BS);   table[4 * gpl::BS + 3] = gpl::b_off(gpl::BS);
    table[4 * gpl::BV + 2] = gpl::w_off(gpl::BV);   table[4 * gpl::BV + 3] = gpl::b_off(gpl::BV);
    table[4 * gpl::B2 + 2] = gpl::w_off(gpl::B2);   table[4 * gpl::B2 + 3] = gpl::b_off(gpl::B2);
    table[4 * gpl::V1 + 2] = gpl::w_off(gpl::V1);   table[4 * gpl::V1 + 3] = gpl::b_off(gpl::V1);
    table[4 * gpl::V2 + 2] = gpl::w_off(gpl::V2);   table[4 * gpl::V2 + 3] = gpl::b_off(gpl::V2);
    table[4 * gpl::R1 + 2] = gpl::w_off(gpl::R1);   table[4 * gpl::R1 + 3] = gpl::b_off(gpl::R1);
    table[4 * gpl::R2 + 2] = gpl::w_off(gpl::R2);   table[4 * gpl::R2 + 3] = gpl::b_off(gpl::R2);
    table[4 * gpl::NLAYER + 0] = gpl::D4_W; table[4 * gpl::NLAYER + 1] = gpl::D4_B;
    table[4 * gpl::NLAYER + 2] = gpl::R3_W; table[4 * gpl::NLAYER + 3] = gpl::R3_B;
    return GPNERF_OK;
}

const char* gpnerf_strerror(int code) {
    switch (code) {
        case GPNERF_OK: return "ok";
        case GPNERF_E_ARG: return "invalid argument";
        case GPNERF_E_LAUNCH: return "kernel launch failed";
        case GPNERF_E_DEVICE: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

int gpnerf_pack_head(const GpnerfHeadParams* p, float* blob) {
    if (!p || !blob) return GPNERF_E_ARG;
    const float* const* all = reinterpret_cast<const float* const*>(p);
    for (size_t i = 0; i < sizeof(GpnerfHeadParams) / sizeof(float*); ++i)
        if (!all[i]) return GPNERF_E_ARG;
    memset(blob, 0, sizeof(float) * gpl::BLOB_FLOATS);
    pack_layer(gpl::GEO, p->geo_w, p->geo_b, 64, 128, blob);
    pack_layer(gpl::D1, p->d1_w, p->d1_b, 64, 134, blob);
    pack_layer(gpl::D2, p->d2_w, p->d2_b, 32, 64, blob);
    pack_layer(gpl::D3, p->d3_w, p->d3_b, 16, 32, blob);
    pack_layer(gpl::BS, p->b1_w, p->b1_b, 64, 105, blob);     // [mean,var] columns + the layer's bias
    pack_layer(gpl::BV, p->b1_w, nullptr, 64, 105, blob);     // per-view columns, accumulates onto BS
    pack_layer(gpl::B2, p->b2_w, p->b2_b, 32, 64, blob);
    pack_layer(gpl::V1, p->v1_w, p->v1_b, 32, 32, blob);
    pack_layer(gpl::V2, p->v2_w, p->v2_b, 32, 32, blob);
    pack_layer(gpl::R1, p->r1_w, p->r1_b, 32, 96, blob);
    pack_layer(gpl::R2, p->r2_w, p->r2_b, 16, 32, blob);
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 8; ++r) {
            blob[gpl::D4_W + h * 8 + r] = p->d4_w[gpl::ft(r, h)] * PACK_LN2;
            for (int o = 0; o < 3; ++o) blob[gpl::R3_W + o * 16 + h * 8 + r] = p->r3_w[o * 16 + gpl::ft(r, h)] * PACK_LN2;
        }
    blob[gpl::D4_B] = p->d4_b[0];
    for (int o = 0; o < 3; ++o) blob[gpl::R3_B + o] = p->r3_b[o];
    return GPNERF_OK;
}

int gpnerf_pack_head_ref(const GpnerfHeadParams* p, float* blob) {
    if (!p || !blob) return GPNERF_E_ARG;
    const float* const* all = reinterpret_cast<const float* const*>(p);
    for (size_t i = 0; i < sizeof(GpnerfHeadParams) / sizeof(float*); ++i)
        if (!all[i]) return GPNERF_E_ARG;
    memset(blob, 0, sizeof(float) * gpl::BLOB_FLOATS);
    pack_layer_ref(gpl::GEO, p->geo_w, p->geo_b, 64, 128, blob);
    pack_layer_ref(gpl::D1, p->d1_w, p->d1_b, 64, 134, blob);
    pack_layer_ref(gpl::D2, p->d2_w, p->d2_b, 32, 64, blob);
    pack_layer_ref(gpl::D3, p->d3_w, p->d3_b, 16, 32, blob);
    pack_layer_ref(gpl::BS, p->b1_w, p->b1_b, 64, 105, blob);     // [mean, var] columns; the bias tile is added after BV's chain
    pack_layer_ref(gpl::BV, p->b1_w, nullptr, 64, 105, blob);
    pack_layer_ref(gpl::B2, p->b2_w, p->b2_b, 32, 64, blob);
    pack_layer_ref(gpl::V1, p->v1_w, p->v1_b, 32, 32, blob);
    pack_layer_ref(gpl::V2, p->v2_w, p->v2_b, 32, 32, blob);
    pack_layer_ref(gpl::R1, p->r1_w, p->r1_b, 32, 96, blob);
    pack_layer_ref(gpl::R2, p->r2_w, p->r2_b, 16, 32, blob);
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 8; ++r) {
            blob[gpl::D4_W + h * 8 + r] = p->d4_w[2 * r + h];
            for (int o = 0; o < 3; ++o) blob[gpl::R3_W + o * 16 + h * 8 + r] = p->r3_w[o * 16 + 2 * r + h];
        }
    blob[gpl::D4_B] = p->d4_b[0];
    for (int o = 0; o < 3; ++o) blob[gpl::R3_B + o] = p->r3_b[o];
    return GPNERF_OK;
}

int64_t gpnerf_head_blob_split_floats(void) { return gph::BLOB_WORDS; }

int gpnerf_pack_head_split(const GpnerfHeadParams* p, float* blob) {
    if (!p || !blob) return GPNERF_E_ARG;
    const float* const* all = reinterpret_cast<const float* const*>(p);
    for (size_t i = 0; i < sizeof(GpnerfHeadParams) / sizeof(float*); ++i)
        if (!all[i]) return GPNERF_E_ARG;
    memset(blob, 0, sizeof(float) * gph::BLOB_WORDS);
    struct Spec { int L; const float *W, *b; int n_out, n_in; };
    const Spec specs[] = {{gpl::GEO, p->geo_w, p->geo_b, 64, 128}, {gpl::D1, p->d1_w, p->d1_b, 64, 134}, {gpl::D2, p->d2_w, p->d2_b, 32, 64},
                          {gpl::D3, p->d3_w, p->d3_b, 16, 32},    {gpl::BS, p->b1_w, p->b1_b, 64, 105}, {gpl::BV, p->b1_w, nullptr, 64, 105},
                          {gpl::B2, p->b2_w, p->b2_b, 32, 64},    {gpl::V1, p->v1_w, p->v1_b, 32, 32},  {gpl::V2, p->v2_w, p->v2_b, 32, 32},
                          {gpl::R1, p->r1_w, p->r1_b, 32, 96},    {gpl::R2, p->r2_w, p->r2_b, 16, 32}};
    uint16_t* halfs = reinterpret_cast<uint16_t*>(blob);
    for (const Spec& sp : specs) {
        for (int m = 0; m < gph::MT[sp.L]; ++m) {
            for (int st = 0; st < gph::NS[sp.L]; ++st) {
                const size_t base = ((size_t)gph::w_off(sp.L) + (size_t)(m * gph::NS[sp.L] + st) * gph::STEP_WORDS) * 2;   // in halfs
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = 32 * m + (lane & 31), h = lane >> 5, c = gph::col_of(sp.L, st, j, h);
                        const float v = (row < sp.n_out && c >= 0 && c < sp.n_in) ? sp.W[(size_t)row * sp.n_in + c] * pack_scale(sp.L, c) : 0.f;
                        const _Float16 hi = f16_rtz(v);
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        uint16_t hb, lb;
                        memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
                        halfs[base + lane * 8 + j] = hb;
                        halfs[base + 512 + lane * 8 + j] = lb;
                    }
            }
            float* bt = blob + gph::b_off(sp.L) + m * 32;
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * m + gpl::ft(r, h);
                    bt[h * 16 + r] = (sp.b && row < sp.n_out) ? sp.b[row] * PACK_LOG2E : 0.f;
                }
        }
    }
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 8; ++r) {
            blob[gph::D4_W + h * 8 + r] = p->d4_w[gpl::ft(r, h)] * PACK_LN2;
            for (int o = 0; o < 3; ++o) blob[gph::R3_W + o * 16 + h * 8 + r] = p->r3_w[o * 16 + gpl::ft(r, h)] * PACK_LN2;
        }
    blob[gph::D4_B] = p->d4_b[0];
    for (int o = 0; o < 3; ++o) blob[gph::R3_B + o] = p->r3_b[o];
    return GPNERF_OK;
}

int gpnerf_fold_volumes(const GpnerfFrame* f, float* const* out, void* stream) {
    if (!f || !out || !f->head_blob) return GPNERF_E_ARG;
    for (int l = FOLD_FROM; l < GPNERF_LEVELS; ++l)
        if (!f->vol[l] || !out[l] || f->vol_dhw[l][0] < 1 || f->vol_dhw[l][1] < 1 || f->vol_dhw[l][2] < 1) return GPNERF_E_ARG;
    if (device_ready(nullptr) != GPNERF_OK) return GPNERF_E_DEVICE;
    FoldArgs a;
    a.tiles_before[0] = 0;
    for (int l = FOLD_FROM; l < GPNERF_LEVELS; ++l) {
        const int i = l - FOLD_FROM;
        a.vol[i] = f->vol[l]; a.out[i] = out[l];
        a.n_vox[i] = (long)f->vol_dhw[l][0] * f->vol_dhw[l][1] * f->vol_dhw[l][2];
        a.tiles_before[i + 1] = a.tiles_before[i] + (a.n_vox[i] + 31) / 32;
    }
    const long wgs = (a.tiles_before[GPNERF_LEVELS - FOLD_FROM] + 3) / 4;
    hipLaunchKernelGGL(fold_volume_kernel, dim3((unsigned)(wgs < 8192 ? wgs : 8192)), dim3(256), 0, S_(stream), f->head_blob, a);
    return launch_status();
}

int gpnerf_render_fused(const GpnerfFrame* f, const float* rays, int64_t n_rays, int32_t n_samples, uint32_t flags,
                        float term_eps, const int32_t* ray_order, const GpnerfOutputs* out, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (n_rays == 0) return GPNERF_OK;          // empty ray list: nothing to do (pointers may be null)
    if (!f || !rays || !out || n_rays < 0 || n_samples < 1) return GPNERF_E_ARG;
    if (!out->rgb || !out->depth || !out->acc || !out->disp) return GPNERF_E_ARG;
    FrameK k;
    if (!to_framek(f, k, true, true)) return GPNERF_E_ARG;
    if (n_rays >= ((int64_t)1 << 31)) return GPNERF_E_ARG;      // output rows are 32-bit values inside the kernel
    const int64_t tiles = (n_rays + RAYS_PER_WAVE - 1) / RAYS_PER_WAVE;
    const size_t lds_bytes = sizeof(float) * gpl::BLOB_FLOATS + DEFER_LDS_BYTES;       // head image + the wavefronts' colour queues (render_tile)
    const bool split16 = (flags & GPNERF_FLAG_SPLIT_F16) != 0;
    // the fp32 form: reference order (FORM_F32, head_blob_ref) unless the frame carries folded volumes and the caller did not ask
    // for the reference's order; the guarded split form's fix-up launch is the reference-order form too
    const bool folded = !split16 && k.vol_fold[GPNERF_LEVELS - 1] != nullptr && !(flags & GPNERF_FLAG_REF_ORDER);
    if (split16 && !f->head_blob_split) return GPNERF_E_ARG;
    if (!split16 && !folded && !f->head_blob_ref) return GPNERF_E_ARG;
    if (folded && !f->head_blob) return GPNERF_E_ARG;
    if (split16 && (flags & GPNERF_FLAG_SPLIT_GUARD) && !f->head_blob_ref) return GPNERF_E_ARG;
    // GPNERF_FLAG_SPLIT_GUARD: the split form records the tiles in which an MFMA operand reached the f16 range, and a second
    // launch renders exactly those again in the fp32 form (it returns at once when there are none).  The flags live in the
    // last guard_bytes() of the workspace.
    const bool guard = split16 && (flags & GPNERF_FLAG_SPLIT_GUARD) != 0;
    unsigned* guard_words = nullptr;
    if (guard) {
        const size_t gb = guard_bytes(n_rays);
        if (!workspace || workspace_bytes < QUEUE_BYTES + gb) return GPNERF_E_ARG;
        workspace_bytes = (workspace_bytes - gb) & ~(size_t)255;
        guard_words = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + workspace_bytes);
        if (!zero_async(guard_words, gb, stream)) return GPNERF_E_LAUNCH;
    }
    int n_cus = 0;
    if (device_ready(&n_cus) != GPNERF_OK) return GPNERF_E_DEVICE;
    {   // GPNERF_FLAG_RESERVE_CUS(n): plan the launch for n fewer compute units (whole XCD rounds of 8, at least 8 stay), so that
        // kernels of OTHER streams -- the next frame's encoder and volume builder -- find free CUs while the persistent workgroups run
        const int reserve = (int)((flags >> 24) & 0xffu) & ~7;
        if (reserve > 0) n_cus = n_cus - reserve >= 8 ? n_cus - reserve : (n_cus >= 8 ? 8 : n_cus);
        flags &= 0x00ffffffu;
    }
    const bool culling = (flags & GPNERF_FLAG_OCC_CULL) != 0 && f->occ != nullptr;
    if (flags & GPNERF_FLAG_OCC_CULL) k.voxel[0] = k.voxel[1] = k.voxel[2] = 0.005f;   // demo_render.py:91 `xyz / 0.005`
    // occupancy culling: the keep bits of every sample in one pass before the launch (occupancy_mask_kernel), in the last
    // cull_mask_bytes() of the workspace (before the guard's block); outputs that need every step written keep the in-loop test
    unsigned long long* cull_mask = nullptr;
    static int f_mask = -1;
    if (f_mask < 0) f_mask = dbg_int("GPNERF_CULL_MASK", 3, 0, 3);       // experiments: 1 = keep bits, 2 = + tile order
    if ((f_mask & 1) && culling && workspace && n_samples <= 128 && !out->weights && !out->raw && workspace_bytes >= QUEUE_BYTES + cull_mask_bytes(n_rays)) {
        workspace_bytes = (workspace_bytes - cull_mask_bytes(n_rays)) & ~(size_t)255;
        cull_mask = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + workspace_bytes);
    }
    // workspace layout: [0, QUEUE_BYTES) the tile queue's counters, then the per-segment partial composites
    const size_t seg_bytes = workspace && workspace_bytes > QUEUE_BYTES ? workspace_bytes - QUEUE_BYTES : 0;
    float* const seg_part = seg_bytes ? reinterpret_cast<float*>(static_cast<char*>(workspace) + QUEUE_BYTES) : nullptr;
    // A frame of more than ~1.25 rounds of wavefronts whose launch can list its colour work (frame-level deferral, below) keeps whole
    // tiles on the queue: the sample loop's tiles then cost the same and the listed colour work is balanced by construction, which
    // is what splitting a tile's samples bought (384 x 384 x 64: 7.06 ms split in two, 6.17 ms whole; 300 / 320 / 448: the same
    // either way; 272 x 272, 1.13 rounds, still gains from the split: 3.86 against 4.24) -- and the maps stay bit-identical to the
    // same rays' in any other launch.
    // (decided by the frame and the workspace alone -- not by GPNERF_FLAG_NO_EXITS or a `raw` output, whose launches list nothing: a
    //  launch and its diagnostic twin must cut the frame the same way to be compared bit for bit)
    const bool can_list = workspace && !split16 && !(flags & (GPNERF_FLAG_OCC_CULL | GPNERF_FLAG_EARLY_TERM)) &&
                          gdef_fits(n_rays, n_samples) && workspace_bytes >= QUEUE_BYTES + gdef_bytes(n_rays, n_samples);
    const bool may_split = seg_bytes && !(flags & GPNERF_FLAG_EARLY_TERM) && !out->samples_done &&
                           !(can_list && tiles * 4 >= (int64_t)n_cus * GPNERF_MAX_WAVES * 5);
    Geometry g = choose_geometry(tiles, n_samples, may_split, seg_bytes, n_rays, n_cus);
    // whole rounds of full workgroups + a remainder launch (below) when the frame is that shape
    static int f_rem = -1;
    if (f_rem < 0) f_rem = dbg_int("GPNERF_REMAINDER", 2, 0, 2);     // 1: two launches (whole rounds, then the remainder); 2: one launch (below)
    const int64_t slots = (int64_t)n_cus * GPNERF_MAX_WAVES;
    const int64_t rem_tiles = tiles % slots;
    const bool remainder = f_rem && workspace && workspace_bytes >= QUEUE_BYTES && !(flags & (GPNERF_FLAG_EARLY_TERM | GPNERF_FLAG_OCC_CULL)) &&
                           n_cus >= 8 && tiles > slots && rem_tiles > 0 && rem_tiles * 8 <= slots && n_samples >= 8 && !dbg_env("GPNERF_WAVES");
    if (remainder) { g.waves = GPNERF_MAX_WAVES; g.split = 1; }
    // EXPERIMENT, not the default (GPNERF_DEBUG=1 GPNERF_QSPLIT=2 / 4 / 8): a frame of one to two rounds of wavefronts (2 048 < tiles
    // <= 4 096 on 256 CUs: the 74 k-ray ZJU-sized frame, 272^2 ... 360^2 crops) ends with every SIMD's last whole tile running alone
    // (profiles/r04/l_survey_frame_analysis.md).  Here the persistent workgroups pull (tile, sample segment) UNITS of
    // S / n samples from the queue instead of whole tiles (same 32 rays per wavefront, same cost per step), each unit parks its
    // partial composite and combine_segments_kernel merges them as it does for small frames.  Measured, reference-order form
    // (profiles/r05/l_qsplit_sweep.txt): n = 8: 272^2 4.83 -> 4.40 ms, 320^2 6.39 -> 5.79, 360^2 7.49 -> 7.31, the survey frame
    // 4.61 -> 4.52 (n = 4: 4.34), exactly one round (256^2) 3.83 -> 3.90; segment-major against tile-major unit order: no
    // difference.  NOT adopted: the merge associates the transmittance product per segment (~1e-7 relative), so a frame in that
    // range would no longer be bit-identical to its own shards or to the same rays inside a larger launch -- the invariant the
    // strong-scaling path is tested on (tests/test_gpu_configs.py: the 8-rank plan of the 1024^2 frame has shares of exactly 4 096
    // tiles) -- for 2 % on the frame size that matters.  A bit-exact version needs the units of a tile to resume each other's
    // state in order (the chained form's parked state inside one launch).
    static int f_dynamic = -1;
    if (f_dynamic < 0) f_dynamic = dbg_int("GPNERF_DYNAMIC", 1, 0, 1);
    static int f_qsplit = -1;
    if (f_qsplit < 0) f_qsplit = dbg_int("GPNERF_QSPLIT", 0, 0, 8);
    bool qsplit = false;
    {
        int q = f_qsplit > 1 ? f_qsplit : 0;
        while (q > 1 && n_samples / q < 8) q >>= 1;
        if (q > 1 && may_split && !culling && seg_bytes >= (size_t)n_rays * q * 16 * sizeof(float) && tiles > n_cus && f_dynamic) {
            g.waves = GPNERF_MAX_WAVES; g.split = q; qsplit = true;
        }
    }
    int64_t blocks = (tiles * g.split + g.waves - 1) / g.waves;
    // more than one round of workgroups and nothing split: persistent workgroups + tile queue (see render_fused_kernel)
    const bool dynamic = f_dynamic && workspace && workspace_bytes >= QUEUE_BYTES && (g.split == 1 || qsplit) && blocks > n_cus;
    if (dynamic) {
        if (!zero_async(workspace, QUEUE_BYTES, stream)) return GPNERF_E_LAUNCH;
        blocks = n_cus;
    }
    const bool do_remainder = remainder && dynamic && !qsplit;
    const OutK ok = to_outk(out, ray_order);
    KArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.fr = k; ka.rays = rays; ka.n_rays = (long)n_rays; ka.S = (int)n_samples; ka.flags = (unsigned)flags; ka.term_eps = term_eps;
    ka.out = ok; ka.split = g.split; ka.part = seg_part;
    ka.dynamic = dynamic ? 1 : 0; ka.queue = static_cast<unsigned*>(workspace);
    ka.guard = guard_words;
    const dim3 full_block(GPNERF_MAX_WAVES * 64);
    // the fp32 form over the tiles the guarded split form flagged: persistent workgroups on a queue of their own
    auto fixup = [&]() -> int {
        if (!guard) return launch_status();
        if (hipGetLastError() != hipSuccess) return GPNERF_E_LAUNCH;
        KArgs kf = ka;
        kf.flags = (unsigned)flags & ~(GPNERF_FLAG_SPLIT_F16 | GPNERF_FLAG_SPLIT_GUARD);
        kf.split = 1; kf.part = nullptr; kf.dynamic = 1; kf.queue = guard_words + 8; kf.wave_cap = 0;
        kf.chain = 0; kf.seg = 0; kf.list_in = nullptr; kf.count_in = nullptr; kf.list_out = nullptr; kf.count_out = nullptr;
        const int64_t wg = (tiles + GPNERF_MAX_WAVES - 1) / GPNERF_MAX_WAVES;
        if (kf.cull_mask)
            hipLaunchKernelGGL((render_fused_kernel<FORM_F32_FIXUP, false, true>), dim3((unsigned)(wg < n_cus ? wg : n_cus)), full_block, lds_bytes,
                               S_(stream), kf);
        else
            hipLaunchKernelGGL((render_fused_kernel<FORM_F32_FIXUP, false>), dim3((unsigned)(wg < n_cus ? wg : n_cus)), full_block, lds_bytes,
                               S_(stream), kf);
        return launch_status();
    };
    static int f_cap = -1;
    if (f_cap < 0) f_cap = dbg_int("GPNERF_WAVE_CAP", 0, 0, 8);      // experiments: waves per CU that pull tiles
    ka.wave_cap = f_cap;
    ka.skip = (flags & GPNERF_FLAG_NO_EXITS) ? 0 : 3;
    // every form defers the colour branch sample by sample (render_tile) unless the exits are off or `raw` wants every rgb
    static int f_defer = -1;
    if (f_defer < 0) f_defer = dbg_int("GPNERF_DEFER", 1, 0, 1);
    const bool deferred = f_defer && (ka.skip & 2) && !out->raw;
    const int sel = guard ? SEL_GUARD : (split16 ? SEL_SPLIT : (folded ? SEL_FOLD : SEL_REF));
    // Frame-level deferral (fp32 forms, persistent launches of whole tiles): the sample loop only LISTS the samples whose weight is
    // not zero; the list is evaluated by the launch's own wavefronts (`unify`, below) or by colour_units_kernel, and
    // colour_accumulate_kernel adds every ray's terms in order (see there).
    static int f_gdef = -1;
    if (f_gdef < 0) f_gdef = dbg_int("GPNERF_FRAME_DEFER", 1, 0, 1);
    const bool gdef_ok = f_gdef && deferred && !culling && !cull_mask && (sel == SEL_REF || sel == SEL_FOLD) && !(flags & GPNERF_FLAG_OCC_CULL) &&
                         gdef_fits(n_rays, n_samples) && workspace != nullptr;
    unsigned* gd_flags = nullptr;
    // the block sits behind what the launch's own form keeps in the workspace (`behind` bytes); false: no room, the wavefronts keep their passes
    auto gdef_setup = [&](KArgs& kx, size_t behind) -> bool {
        behind = align256(behind);
        if (!gdef_ok || workspace_bytes < behind + gdef_bytes(n_rays, n_samples)) return false;
        char* const b = static_cast<char*>(workspace) + behind;
        if (!zero_async(b, GDEF_HEAD_BYTES, stream)) return false;
        const size_t cnt_bytes = align256((size_t)n_rays * sizeof(int)), flag_bytes = gdef_flag_bytes(n_rays, n_samples);
        kx.gd_ctrl = reinterpret_cast<unsigned*>(b);
        kx.gd_cnt = reinterpret_cast<int*>(b + GDEF_HEAD_BYTES);
        gd_flags = reinterpret_cast<unsigned*>(b + GDEF_HEAD_BYTES + cnt_bytes);
        kx.gd_ent = reinterpret_cast<uint4*>(b + GDEF_HEAD_BYTES + cnt_bytes + flag_bytes);
        kx.gd_rgbw = reinterpret_cast<f32x4*>(b + GDEF_HEAD_BYTES + cnt_bytes + flag_bytes + gdef_entries(n_rays, n_samples) * sizeof(uint4));
        return true;
    };
    // unified form: the listing launch's own wavefronts evaluate the list once they have no tile left (render_fused_kernel, UNI)
    static int f_uni = -1, f_budget = -1;
    // (uni_budget = units a wavefront may evaluate between two tiles: measured 0 / 4 / 24 / 100 -> 9.72 / 9.75 / 9.82 / 11.8 ms on the
    //  bench frame -- colour work between tiles buys no overlap and unbalances the tile queue; the list is evaluated when a wavefront
    //  has no tile left, which is what fills the end of a launch whose tiles differ in cost)
    if (f_uni < 0) { f_uni = dbg_int("GPNERF_UNIFIED", 3, 0, 3); f_budget = dbg_int("GPNERF_UNI_BUDGET", 0, 0, 4096); }
    auto unify = [&](KArgs& kx, long waves) -> bool {
        if (!kx.gd_ent || !gd_flags || (flags & GPNERF_FLAG_SHARED_DEVICE)) return false;
        if (!zero_async(gd_flags, gdef_flag_bytes(n_rays, n_samples), stream)) return false;
        kx.gd_flag = gd_flags;
        kx.gd_waves = (int)waves;
        kx.uni_budget = f_budget;
        return true;
    };
    // the list's evaluation and the colour map of launch slots [0, n_slots) (behind the launches that listed the entries)
    auto colour_phase = [&](const KArgs& kx, long n_slots) -> bool {
        if (hipGetLastError() != hipSuccess) return false;
        if (kx.gd_flag) {       // unified form: the launch has evaluated its list itself
            hipLaunchKernelGGL(colour_accumulate_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, S_(stream), (const int*)kx.gd_cnt,
                               (const f32x4*)kx.gd_rgbw, n_slots, (int)n_samples, ok.order, ok.rgb);
            return hipGetLastError() == hipSuccess;
        }
        KArgs kc = kx;
        static int f_uchunk = -1;
        if (f_uchunk < 0) f_uchunk = dbg_int("GPNERF_UNIT_CHUNK", 64, 1, 4096);
        kc.chunk = f_uchunk;
        if (sel == SEL_FOLD) hipLaunchKernelGGL((colour_units_kernel<FORM_F32_FOLD>), dim3((unsigned)n_cus), dim3(64 * UNIT_WAVES), lds_bytes, S_(stream), kc);
        else hipLaunchKernelGGL((colour_units_kernel<FORM_F32>), dim3((unsigned)n_cus), dim3(64 * UNIT_WAVES), lds_bytes, S_(stream), kc);
        if (hipGetLastError() != hipSuccess) return false;
        hipLaunchKernelGGL(colour_accumulate_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, S_(stream), (const int*)kx.gd_cnt,
                           (const f32x4*)kx.gd_rgbw, n_slots, (int)n_samples, ok.order, ok.rgb);
        return hipGetLastError() == hipSuccess;
    };
    static int f_segmajor = -1;
    if (f_segmajor < 0) f_segmajor = dbg_int("GPNERF_QSPLIT_SEGMAJOR", 0, 0, 1);
    ka.seg_major = f_segmajor;
    static int f_stagger = -1;
    if (f_stagger < 0) f_stagger = dbg_int("GPNERF_STAGGER", 0, 0, 4096);
    ka.stagger = f_stagger;
    static int f_chunk = -1;
    if (f_chunk < 0) f_chunk = dbg_int("GPNERF_QUEUE_CHUNK", 64, 1, 4096);
    ka.chunk = f_chunk > 0 ? f_chunk : (int)((tiles + 7) / 8);        // 0: one contiguous run per XCD
    static int f_tail = -1;
    if (f_tail < 0) f_tail = dbg_int("GPNERF_CHAIN_TAILP", 2, 1, 8);
    ka.tail_p = f_tail;
    // ONE launch for a frame of whole rounds + a few tiles: the segmented form's kernel over all S samples as a single segment
    // (term_eps = 0: nothing is ever frozen), whose work units are the whole rounds' 32-ray tiles at one sample per step AND the
    // remaining tiles at eight samples of a ray per step (chain_plan), all on one tile queue.  In a lone round a SIMD's older
    // wavefront is through its tile after ~2.7 ms and the younger after ~4.0 ms: the short remainder units fill exactly that
    // gap, where a second launch (f_rem = 1) had to wait for the first to drain.  73 689-ray frame: 4.58 -> 4.36 ms.
    if (do_remainder && f_rem == 2 && tiles < 2 * slots && !out->samples_done) {      // (several whole rounds: the plain kernel's loop is ~2 % faster than the segmented form's, two launches win: 576x576x64 17.08 against 17.46 ms)
        KArgs ku = ka;
        ku.split = 1; ku.dynamic = 1; ku.part = nullptr;
        ku.chain = (int)n_samples; ku.seg = 0; ku.term_eps = 0.f;
        ku.k_begin = 0; ku.k_end = (int)n_samples;
        ku.first_slot = 0; ku.first_items = (long)n_rays;
        ku.list_in = nullptr; ku.count_in = nullptr; ku.list_out = nullptr; ku.count_out = nullptr; ku.chunk_cnt = nullptr;
        ku.p_cap = (long)(slots * RAYS_PER_WAVE);
        const bool listed = gdef_setup(ku, QUEUE_BYTES);
        if (listed && (f_uni & 2)) unify(ku, (long)n_cus * GPNERF_MAX_WAVES);
        launch_render<true, false>(sel, deferred, dim3((unsigned)n_cus), full_block, S_(stream), ku);
        if (listed && !colour_phase(ku, (long)n_rays)) return GPNERF_E_LAUNCH;
        return fixup();
    }
    // Early termination on frames of at least one round of wavefronts: the samples are walked in segments of chain_len(), one
    // persistent-queue launch per segment.  A ray that is opaque stops (per ray, not per tile); the rays that go on park 16
    // floats and are appended to the next launch's list, so every launch packs the survivors 32 to a wavefront again: on the
    // bench frame a ray needs 20 % of its samples, a fixed 32-ray tile 35-43 % (until its last ray is opaque).  The launches
    // are enqueued unconditionally -- one that finds its list empty returns before it stages anything.
    // (frames of less than one round of waves gain nothing from it, and every XCD's queue needs workgroups of its own)
    const size_t need_chain = (flags & GPNERF_FLAG_EARLY_TERM) && !culling && f_dynamic && tiles >= (int64_t)n_cus * GPNERF_MAX_WAVES && n_cus >= 8
                                  ? chain_bytes(n_rays, n_samples) : 0;
    if (need_chain && workspace && workspace_bytes >= need_chain) {
        const int n_seg = chain_segs(n_samples);
        char* const base = static_cast<char*>(workspace);
        unsigned* const ctrl = reinterpret_cast<unsigned*>(base);      // [n_seg][8] queue counters, [n_seg] list lengths, [n_seg][chunks] survivor counters
        const size_t list_bytes = align256((size_t)n_rays * sizeof(int)), ctrl_bytes = chain_ctrl_bytes(n_seg, n_rays);
        const size_t n_chunks = chain_chunks(n_rays);
        int* const lists[2] = {reinterpret_cast<int*>(base + ctrl_bytes), reinterpret_cast<int*>(base + ctrl_bytes + list_bytes)};
        int* const sparse = reinterpret_cast<int*>(base + ctrl_bytes + 2 * list_bytes);
        if (!zero_async(base, ctrl_bytes, stream)) return GPNERF_E_LAUNCH;
        ka.split = 1; ka.dynamic = 1; ka.chain = chain_len(n_samples);
        ka.part = reinterpret_cast<float*>(base + ctrl_bytes + 3 * list_bytes);
        const int64_t wg = (tiles + GPNERF_MAX_WAVES - 1) / GPNERF_MAX_WAVES;
        const unsigned grid = (unsigned)(wg < n_cus ? wg : n_cus);
        static float f_fill = -1.f;
        if (f_fill < 0.f) { const char* e = dbg_env("GPNERF_CHAIN_PFILL"); f_fill = e ? fminf(fmaxf((float)atof(e), 0.f), 8.f) : 1.f; }
        ka.p_cap = (long)((double)grid * GPNERF_MAX_WAVES * RAYS_PER_WAVE * f_fill);
        ka.first_slot = 0; ka.first_items = (long)n_rays;
        const bool listed = gdef_setup(ka, need_chain);
        int begins[CHAIN_MAX_SEGS + 2];
        const int n_launch = chain_schedule((int)n_samples, begins);
        // (the unified form here -- every segment launch evaluating what is listed so far, the next launch's tickets starting at the
        //  list's end -- measured 6.95 -> 7.86 ms on configs[2]: every 16-step visit pads its last unit and six launches each wait
        //  for their lists.  The list is evaluated once, by colour_units_kernel behind the last launch.)
        for (int sg = 0; sg < n_launch; ++sg) {
            ka.seg = sg;
            ka.k_begin = begins[sg]; ka.k_end = begins[sg + 1];
            ka.queue = ctrl + 8 * sg;
            ka.list_in = sg ? lists[(sg - 1) & 1] : nullptr;
            ka.count_in = sg ? ctrl + 8 * n_seg + (sg - 1) : nullptr;
            const bool last = sg + 1 == n_launch;
            ka.list_out = last ? nullptr : sparse;
            ka.count_out = ctrl + 8 * n_seg + sg;
            ka.chunk_cnt = ctrl + 9 * n_seg + (size_t)sg * n_chunks;
            launch_render<true, false>(sel, deferred, dim3(grid), full_block, S_(stream), ka);
            if (!last)     // close the gaps of the sparse list, in order: the next launch's dense input
                hipLaunchKernelGGL(compact_list_kernel, dim3((unsigned)n_chunks), dim3(256), 0, S_(stream), (const int*)sparse,
                                   (const unsigned*)ka.chunk_cnt, (const unsigned*)ka.count_in, ka.first_items, lists[sg & 1], ka.count_out);
            if (hipGetLastError() != hipSuccess) return GPNERF_E_LAUNCH;
        }
        if (listed && !colour_phase(ka, (long)n_rays)) return GPNERF_E_LAUNCH;
        return fixup();
    }
    // A frame of q whole rounds of wavefronts plus a FEW tiles (at most an eighth of a round) ends with those few running alone,
    // S dependent steps at one wave per CU.  The persistent launch then takes the whole rounds, and the remainder goes to the
    // segmented form's kernel as ONE segment of all S samples: its work units put 8 samples of a ray side by side (render_tile's
    // P), S / 8 steps each, with the plain form's arithmetic per ray (term_eps = 0: nothing is ever frozen).  Bit-identical
    // results; 576x576x64: 18.8 -> 18.0 ms.  (A larger remainder is better left to the queue: CUs with few waves step faster.)
    if (do_remainder) ka.n_rays = (long)((tiles - rem_tiles) * RAYS_PER_WAVE);
    const bool gdef = dynamic && !qsplit && !(flags & GPNERF_FLAG_EARLY_TERM) && gdef_setup(ka, QUEUE_BYTES);
    // unified form, unless a remainder launch follows (with both launches unified -- the second's tickets starting where the first's
    // list ends -- 576 squared measured 12.67 -> 12.89 ms, 370 squared 5.70 -> 6.01):
    // 300 / 320 / 340 / 384 squared: 4.42 / 5.04 / 5.13 / 6.15 ms with the second kernel, 4.17 / 4.51 / 5.00 / 5.62 unified
    if (gdef && (f_uni & 1) && !do_remainder && !cull_mask) unify(ka, (long)blocks * g.waves);
    if (cull_mask) {
        ka.cull_mask = cull_mask;
        {
            hipLaunchKernelGGL(occupancy_mask_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, S_(stream), k, rays, ok.order, (long)n_rays,
                               (int)n_samples, (flags & GPNERF_FLAG_FLIP_SAMPLES) ? 1 : 0, cull_mask);
            if (hipGetLastError() != hipSuccess) return GPNERF_E_LAUNCH;
            if (dynamic && (f_mask & 2)) {      // longest tiles first (see tile_steps_kernel)
                char* const after = reinterpret_cast<char*>(cull_mask) + align256((size_t)n_rays * 2 * sizeof(unsigned long long));
                int* const steps = reinterpret_cast<int*>(after);
                int* const order = reinterpret_cast<int*>(after + align256(cull_tiles(n_rays) * sizeof(int)));
                hipLaunchKernelGGL(tile_steps_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, S_(stream), (const unsigned long long*)cull_mask,
                                   (long)n_rays, steps);
                hipLaunchKernelGGL(tile_sort_kernel, dim3(1), dim3(SORT_THREADS), 0, S_(stream), (const int*)steps, (int)tiles, (int)n_samples, order);
                if (hipGetLastError() != hipSuccess) return GPNERF_E_LAUNCH;
                ka.tile_order = order;
            }
        }
        launch_render<false, true>(sel, deferred, dim3((unsigned)blocks), dim3(g.waves * 64), S_(stream), ka);
    } else
        launch_render<false, false>(sel, deferred, dim3((unsigned)blocks), dim3(g.waves * 64), S_(stream), ka);
    if (do_remainder) {
        if (hipGetLastError() != hipSuccess) return GPNERF_E_LAUNCH;
        KArgs kr = ka;
        kr.n_rays = (long)n_rays;
        kr.queue = static_cast<unsigned*>(workspace) + 8;             // the second set of queue counters of the QUEUE_BYTES block
        kr.chain = (int)n_samples; kr.seg = 0; kr.term_eps = 0.f;
        kr.k_begin = 0; kr.k_end = (int)n_samples;
        kr.first_slot = ka.n_rays; kr.first_items = (long)n_rays - ka.n_rays;
        kr.list_in = nullptr; kr.count_in = nullptr;
        kr.list_out = nullptr;                                        // nothing goes on after segment 0 of 1
        kr.count_out = nullptr; kr.chunk_cnt = nullptr;
        kr.part = nullptr;
        kr.p_cap = (long)(slots * RAYS_PER_WAVE);
        launch_render<true, false>(sel, deferred, dim3((unsigned)n_cus), full_block, S_(stream), kr);
    }
    if (gdef && !colour_phase(ka, (long)n_rays)) return GPNERF_E_LAUNCH;     // (the remainder launch lists into the same block)
    if (g.split > 1) {
        if (hipGetLastError() != hipSuccess) return GPNERF_E_LAUNCH;
        hipLaunchKernelGGL(combine_segments_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, S_(stream),
                           (const float*)seg_part, (long)n_rays, (int)n_samples, g.split, ok);
    }
    return fixup();
}

size_t gpnerf_render_workspace_bytes(int64_t n_rays, int32_t n_samples) {
    if (n_rays <= 0) return 0;
    // the tile queue's counters, plus -- only for frames small enough to profit from splitting -- room for GPNERF_MAX_SPLIT
    // sample segments per ray, 16 floats each; or what the chained segments of an early-terminating launch need
    const size_t plain = QUEUE_BYTES + (n_rays <= 131072 ? (size_t)n_rays * GPNERF_MAX_SPLIT * 16 * sizeof(float) : 0);
    const size_t chain = chain_bytes(n_rays, n_samples);
    // + the keep bits of occupancy culling + the split form's range-guard flags
    // + the entry list of the frame-level colour deferral, behind whichever of the two the launch uses
    const size_t gdef = gdef_fits(n_rays, n_samples) && n_rays > 32 * 8 * 8 ? gdef_bytes(n_rays, n_samples) : 0;
    return align256(plain > chain ? plain : chain) + gdef + cull_mask_bytes(n_rays) + guard_bytes(n_rays);
}

size_t gpnerf_render_guard_bytes(int64_t n_rays) { return n_rays > 0 ? guard_bytes(n_rays) : 0; }

int gpnerf_sample_points(const GpnerfFrame* f, const float* rays, int64_t n_rays, int32_t n_samples, float* pts,
                         float* z_vals, float* grid, void* stream) {
    if (n_rays == 0) return GPNERF_OK;
    if (!f || !rays || n_rays < 0 || n_samples < 1) return GPNERF_E_ARG;
    FrameK k;
    if (!to_framek(f, k, false, false)) return GPNERF_E_ARG;
    const long n = (long)n_rays * n_samples;
    hipLaunchKernelGGL(stage_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), k, rays, (long)n_rays,
                       (int)n_samples, pts, z_vals, grid);
    return launch_status();
}

int gpnerf_sample_volume(const GpnerfFrame* f, const float* grid, int64_t n_points, float* vol_feat, void* stream) {
    if (n_points == 0) return GPNERF_OK;
    if (!f || !grid || !vol_feat || n_points < 0) return GPNERF_E_ARG;
    FrameK k;
    if (!to_framek(f, k, true, false)) return GPNERF_E_ARG;
    const long n = (long)n_points * 2;
    hipLaunchKernelGGL(stage_volume_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), k, grid, (long)n_points,
                       vol_feat);
    return launch_status();
}

int gpnerf_project_gather(const GpnerfFrame* f, const float* pts, int64_t n_points, int32_t neg_ray, float* rgb_feat,
                          float* mask, void* stream) {
    if (n_points == 0) return GPNERF_OK;
    if (!f || !pts || !rgb_feat || !mask || n_points < 0) return GPNERF_E_ARG;
    FrameK k;
    if (!to_framek(f, k, false, true)) return GPNERF_E_ARG;
    const long n = (long)n_points * 2;
    hipLaunchKernelGGL(stage_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), k, pts, (long)n_points,
                       (int)neg_ray, rgb_feat, mask);
    return launch_status();
}

int gpnerf_head_forward(const float* head_blob, const float* vol_feat, const float* rgb_feat, const float* mask,
                        int64_t n_points, float* raw, void* stream) {
    if (n_points == 0) return GPNERF_OK;
    if (!head_blob || !vol_feat || !rgb_feat || !mask || !raw || n_points < 0) return GPNERF_E_ARG;
    return launch_head<0>(head_blob, vol_feat, rgb_feat, mask, n_points, raw, nullptr, stream);
}

int gpnerf_sigma_features(const float* head_blob, const float* vol_feat, const float* rgb_feat, int64_t n_points,
                          float* sigma_feat, float* globalfeat, void* stream) {
    if (n_points == 0) return GPNERF_OK;
    if (!head_blob || !vol_feat || !rgb_feat || !sigma_feat || !globalfeat || n_points < 0) return GPNERF_E_ARG;
    return launch_head<1>(head_blob, vol_feat, rgb_feat, nullptr, n_points, sigma_feat, globalfeat, stream);
}

int gpnerf_rgb_head_forward(const float* head_blob, const float* sigma_feat, const float* rgb_feat, const float* mask,
                            int64_t n_points, float* raw, void* stream) {
    if (n_points == 0) return GPNERF_OK;
    if (!head_blob || !sigma_feat || !rgb_feat || !mask || !raw || n_points < 0) return GPNERF_E_ARG;
    return launch_head<2>(head_blob, sigma_feat, rgb_feat, mask, n_points, raw, nullptr, stream);
}

int gpnerf_composite(const float* raw, const float* z_vals, const float* nvalid, int64_t n_rays, int32_t n_samples,
                     int32_t neg, const GpnerfOutputs* out, void* stream) {
    if (n_rays == 0) return GPNERF_OK;
    if (!raw || !z_vals || !out || !out->rgb || !out->depth || !out->acc || !out->disp || n_rays < 0 || n_samples < 1)
        return GPNERF_E_ARG;
    const int bs = 64;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((n_rays + bs - 1) / bs)), dim3(bs), 0, S_(stream), raw, z_vals, nvalid,
                       (long)n_rays, (int)n_samples, (int)neg, to_outk(out));
    return launch_status();
}

int gpnerf_make_rays(int32_t H, int32_t W, const double* Kinv, const double* Rinv, const double* cam_o, const float* bounds,
                     float* rays, uint8_t* hit, void* stream) {
    if (!Kinv || !Rinv || !cam_o || !bounds || !rays || !hit || H < 1 || W < 1) return GPNERF_E_ARG;
    RayCam c;
    memset(&c, 0, sizeof(c));
    memcpy(c.Kinv_d, Kinv, sizeof(c.Kinv_d));
    memcpy(c.Rinv_d, Rinv, sizeof(c.Rinv_d));
    memcpy(c.o_d, cam_o, sizeof(c.o_d));
    for (int a = 0; a < 3; ++a) {   // bounds + [-0.01, 0.01] in float64 (data_utils.py:98)
        c.bmin_d[a] = (double)bounds[a] + -0.01;
        c.bmax_d[a] = (double)bounds[3 + a] + 0.01;
    }
    const int n = H * W, bs = 256;
    hipLaunchKernelGGL(make_rays_kernel, dim3((n + bs - 1) / bs), dim3(bs), 0, S_(stream), (int)H, (int)W, c, 0, 0,
                       (const uint8_t*)nullptr, (const int*)nullptr, rays, hit);
    return launch_status();
}

int gpnerf_select_pixels(const float* occ, int32_t D, int32_t H, int32_t W, float threshold, const float* voxel_xyz,
                         const float* bounds_min, const float* Rh, const float* Th, const float* pose, const float* K,
                         int32_t img_h, int32_t img_w, uint8_t* pixel_sel, int32_t* world_minmax, void* stream) {
    if (!occ || !voxel_xyz || !bounds_min || !Rh || !Th || !pose || !K || !pixel_sel || !world_minmax || D < 1 || H < 1 ||
        W < 1 || img_h < 1 || img_w < 1)
        return GPNERF_E_ARG;
    SelGeom g;
    memcpy(g.voxel, voxel_xyz, sizeof(g.voxel));
    memcpy(g.bmin, bounds_min, sizeof(g.bmin));
    memcpy(g.Rh, Rh, sizeof(g.Rh));
    memcpy(g.Th, Th, sizeof(g.Th));
    memcpy(g.pose, pose, sizeof(g.pose));
    memcpy(g.K, K, sizeof(g.K));
    {
        const long nb = (long)img_h * img_w, wgs = (nb / 4 + 255) / 256;
        hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)(wgs < 1 ? 1 : (wgs > 2048 ? 2048 : wgs))), dim3(256), 0, S_(stream), pixel_sel, nb);
        if (hipGetLastError() != hipSuccess) return GPNERF_E_LAUNCH;
    }
    hipLaunchKernelGGL(init_minmax_kernel, dim3(1), dim3(64), 0, S_(stream), (int*)world_minmax);
    const long n = (long)D * H * W;
    const long want_blocks = (n + 255) / 256;
    hipLaunchKernelGGL(select_pixels_kernel, dim3((unsigned)(want_blocks < 512 ? want_blocks : 512)), dim3(256), 0, S_(stream), occ, (int)D, (int)H,
                       (int)W, threshold, g, (int)img_h, (int)img_w, pixel_sel, (int*)world_minmax);
    return launch_status();
}

int gpnerf_make_rays_demo(int32_t H, int32_t W, const float* Kinv, const float* pose, const float* bounds,
                          const int32_t* world_minmax_dev, int32_t neg_ray, const uint8_t* pixel_sel, float* rays, uint8_t* hit,
                          void* stream) {
    if (!Kinv || !pose || (!bounds && !world_minmax_dev) || !rays || !hit || H < 1 || W < 1) return GPNERF_E_ARG;
    RayCam c;
    memset(&c, 0, sizeof(c));
    memcpy(c.Kinv, Kinv, sizeof(c.Kinv));
    for (int a = 0; a < 3; ++a) {
        for (int k = 0; k < 3; ++k) c.Rinv[a * 3 + k] = pose[k * 4 + a];              // column a of R: (x @ R)[a] = sum_k x[k] R[k][a]
        c.T[a] = pose[a * 4 + 3];
    }
    // ori_rays_o = (-R^T) @ T, a [3,3] @ [3,1] product accumulated like every CPU `@` (demo_render.py:203)
    for (int a = 0; a < 3; ++a) c.o[a] = fmaf(-pose[2 * 4 + a], pose[11], fmaf(-pose[1 * 4 + a], pose[7], (-pose[0 * 4 + a]) * pose[3]));
    if (bounds)
        for (int a = 0; a < 3; ++a) { c.bmin[a] = bounds[a]; c.bmax[a] = bounds[3 + a]; }   // used as given (demo_render.py:215)
    const int n = H * W, bs = 256;
    hipLaunchKernelGGL(make_rays_kernel, dim3((n + bs - 1) / bs), dim3(bs), 0, S_(stream), (int)H, (int)W, c, 1, (int)neg_ray,
                       pixel_sel, (const int*)world_minmax_dev, rays, hit);
    return launch_status();
}

int gpnerf_build_occupancy(const GpnerfFrame* f, float* occ, void* stream) {
    if (!f || !occ) return GPNERF_E_ARG;
    FrameK k;
    if (!to_framek(f, k, true, false)) return GPNERF_E_ARG;
    const long n = (long)k.vol_dhw[0][0] * k.vol_dhw[0][1] * k.vol_dhw[0][2];
    hipLaunchKernelGGL(occupancy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), k, occ);
    return launch_status();
}

int gpnerf_relayout_volume(const float* ncdhw, float* ndhwc, int32_t D, int32_t H, int32_t W, void* stream) {
    if (!ncdhw || !ndhwc || D < 1 || H < 1 || W < 1) return GPNERF_E_ARG;
    const long P = (long)D * H * W;
    hipLaunchKernelGGL(cfirst_to_clast32_kernel, dim3((unsigned)((P + 63) / 64), 1), dim3(256), 0, S_(stream), ncdhw, ndhwc, P,
                       0L, 0L);
    return launch_status();
}

int gpnerf_relayout_featmaps(const float* nchw, float* nhwc, int32_t V, int32_t H, int32_t W, void* stream) {
    if (!nchw || !nhwc || V < 1 || H < 1 || W < 1) return GPNERF_E_ARG;
    const long P = (long)H * W;
    hipLaunchKernelGGL(cfirst_to_clast32_kernel, dim3((unsigned)((P + 63) / 64), (unsigned)V), dim3(256), 0, S_(stream), nchw,
                       nhwc, P, 32 * P, 32 * P);
    return launch_status();
}

int64_t gpnerf_patch_order_scratch_bytes(int32_t H, int32_t W, int32_t patch_w, int32_t patch_h) {
    if (H < 1 || W < 1 || patch_w < 1 || patch_h < 1) return 0;
    return (int64_t)sizeof(int32_t) * ((int64_t)H + (int64_t)((H + patch_h - 1) / patch_h) * ((W + patch_w - 1) / patch_w) + 1);
}

int gpnerf_patch_order(const uint8_t* mask, int32_t H, int32_t W, int32_t patch_w, int32_t patch_h, int32_t n, int32_t* scratch,
                       int32_t* order, void* stream) {
    if (n == 0) return GPNERF_OK;
    if (!mask || !scratch || !order || H < 1 || W < 1 || patch_w < 1 || patch_w > 64 || patch_h < 1 || patch_h > PO_MAX_PH || n < 0) return GPNERF_E_ARG;
    const int npy = (H + patch_h - 1) / patch_h, npx = (W + patch_w - 1) / patch_w;
    int* const rows = scratch;
    int* const patches = scratch + H;
    if (!zero_async(patches, sizeof(int) * ((size_t)npy * npx + 1), stream)) return GPNERF_E_LAUNCH;
    hipLaunchKernelGGL((po_band_kernel<false>), dim3((unsigned)npy), dim3(256), 0, S_(stream), mask, (int)H, (int)W, (int)patch_w, (int)patch_h, rows,
                       patches, (int)n, order);
    hipLaunchKernelGGL(po_scan_kernel, dim3(1), dim3(1024), 0, S_(stream), rows, (int)H, patches, npy * npx);
    hipLaunchKernelGGL((po_band_kernel<true>), dim3((unsigned)npy), dim3(256), sizeof(int) * (size_t)patch_h * npx, S_(stream), mask, (int)H, (int)W,
                       (int)patch_w, (int)patch_h, rows, patches, (int)n, order);
    return launch_status();
}

int gpnerf_relayout_images(const float* nchw, float* nhwc4, int32_t V, int32_t H, int32_t W, void* stream) {
    if (!nchw || !nhwc4 || V < 1 || H < 1 || W < 1) return GPNERF_E_ARG;
    const long HW = (long)H * W, n = HW * V;
    hipLaunchKernelGGL(images_to_nhwc4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), nchw, nhwc4, HW, (int)V);
    return launch_status();
}

}  // extern "C"
