// gpnerf_conv.hip -- the image encoder's convolutions and their glue on channels-last (NHWC) activations, per frame
// (libs/encoders/UNet.py:17-53,107-131,154-234: reflect-padded 7x7/2 stem, 3x3 stride-1/2 residual units, 1x1 shortcuts and
// output conv, affine InstanceNorm + ReLU / ELU, bilinear x2 upsampling).
//
// Convolution = implicit GEMM on the matrix cores in the render kernel's transposed form,
//     out[co][pixel] += W[co][(tap, ci)] * in[(tap, ci)][pixel],
// with every fp32 operand written as f16 hi + lo (hi = f16(x) to nearest, lo = f16(x - hi)) and three
// v_mfma_f32_32x32x16_f16 per 16-deep k-step (Wlo.xhi + Whi.xlo + Whi.xhi, f32 accumulation): ~23 significant bits per
// operand at 3/16 of the fp32-MFMA cost -- the same arithmetic as GPNERF_FLAG_SPLIT_F16 (head_layout.h, namespace gph), whose
// precondition (operands below the f16 range) holds here by construction for InstanceNorm'd / ReLU'd activations of images.
//   A operand (weights): packed once per parameter version by pack_conv_weight_kernel into [chunk = (tap, 16 input channels)]
//     [32-row output tile][hi 64 lanes x 8 halfs | lo 64 lanes x 8 halfs]; a workgroup streams the chunks of its output tiles
//     through a double-buffered LDS window (one __syncthreads per chunk), every wave reads them as ds_read_b128;
//   B operand (activations): lane (pixel = lane & 31, half = lane >> 5) loads 8 consecutive input channels of its pixel at the
//     tap's (reflected) position -- two dwordx4 from the NHWC tensor -- and converts them to hi / lo in registers; the next
//     chunk's loads are issued before the current chunk's MFMAs;
//   a wave owns 2 x 32 output pixels x COT x 32 output channels (weights read from LDS are used for both pixel tiles).
// Reflection padding is index arithmetic (no padded copy of the input, no pad launches).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/gpnerf_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DEV __device__ __forceinline__
#include "gpnerf_diag.h"       // the lab's hook points, empty in the product (csrc/nodiag/)

constexpr int STEP_BYTES = 2048;     // one (chunk, output tile): 64 lanes x 8 halfs hi (1 KB) + the same for lo
constexpr int PT = 2;                // 32-pixel tiles per wave
constexpr int WAVES = 4;             // waves per workgroup -> 256 output pixels per workgroup

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// hi = f16(x) rounded to NEAREST (v_cvt_pk_f16_f32, new on gfx950): |x - hi| <= 2^-12 |x|, so lo = f16(x - hi) leaves
// |x - hi - lo| <= 2^-23 |x| and the dropped lo.lo product is <= 2^-22 of the full one.  (Rounding hi toward zero, as
// v_cvt_pkrtz does and round 2 shipped, doubles the residual: 4x the dropped product, 2x the operand error -- measured on the
// 512x512 reference vector: 4.2e-5 -> see DESIGN.md 4.4.)
DEV unsigned pk_hi(float a, float b) { const h2 r = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, r); }
DEV unsigned lo_pair(unsigned w, float x0, float x1) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w), "v"(x1));
    return r;
}
// Operand scales (exact powers of two, undone on the accumulators): x = hi + lo carries x to 2^-23 only while lo is a NORMAL f16,
// i.e. |x| >= 2^-14 * 2^12 = 0.25; below that lo is a subnormal with a fixed 2^-24 spacing -- for a typical deep-layer weight of
// 0.03 that is a RELATIVE error of 1e-6, sixteen times the 2^-23 aimed at, and it was the encoder's largest error term (gpu vs
// the exact float64 result 1.7x the reference's own float32 error before).  Weights are packed as 2^12 w
// (range |w| < 16, checked at pack time; full precision down to |w| = 6e-5), activations are staged as 2^4 x (range |x| < 4094:
// an InstanceNorm'd value is bounded by sqrt(h w) |gamma|; full precision down to 0.016, absolute error 2e-9 below).
constexpr float W_SCALE = 4096.f, X_SCALE = 16.f, ACC_UNSCALE = 1.f / (4096.f * 16.f);
struct Frag { h8 hi, lo; };
DEV Frag make_frag(f32x4 a, f32x4 b) {
    a *= X_SCALE;
    b *= X_SCALE;
    u32x4 H, Lo;
    H[0] = pk_hi(a[0], a[1]); Lo[0] = lo_pair(H[0], a[0], a[1]);
    H[1] = pk_hi(a[2], a[3]); Lo[1] = lo_pair(H[1], a[2], a[3]);
    H[2] = pk_hi(b[0], b[1]); Lo[2] = lo_pair(H[2], b[0], b[1]);
    H[3] = pk_hi(b[2], b[3]); Lo[3] = lo_pair(H[3], b[2], b[3]);
    // (this fragment goes straight into an MFMA: one wait state behind the last inline-asm conversion, carried by a statement every
    // reader depends on -- gfx950 does not interlock a VALU write with an MFMA read in the next slot, and LLVM cannot see into
    // lo_pair: tools/micro/asm_producer_hazards.hip, gpnerf_kernels.hip settle_operand.  The 3x3 and stem kernels' conversions
    // go to LDS first and need none.)
    asm("s_nop 0" : "+v"(Lo));
    Frag f;
    f.hi = __builtin_bit_cast(h8, H);
    f.lo = __builtin_bit_cast(h8, Lo);
    return f;
}

// The EXACT form of every convolution kernel below (template argument EXACT; `ResUNet.precision = "fp32"`, the default): the same
// implicit GEMM, tiles, staging, fused InstanceNorm tables and launch chain, with the operands as they are -- fp32 on
// v_mfma_f32_32x32x2_f32, every dot product an fp32 FMA chain over (channel block, tap, channel) like the reference's own,
// no operand range, no flag.  A lane's eight values of a 16-deep k-step (channels 8 half + j, j = 0..7) are two float4 --
// byte for byte where the split form keeps its eight hi and eight lo halfs, so the weight image, the LDS patch and the register
// budget have the same shape -- and a k-step is eight MFMAs pairing channel j of half 0 with channel 8 + j of half 1
// (fma(a1, b1, fma(a0, b0, c))): 8 x 64 cycles where the split form's three f16 MFMAs take 3 x 32.
struct FragX { f32x4 q0, q1; };
DEV void mfma_exact(f32x16& acc, const f32x4& w0, const f32x4& w1, const FragX& x) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[0], x.q0[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[1], x.q0[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[2], x.q0[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[3], x.q0[3], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[0], x.q1[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[1], x.q1[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[2], x.q1[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[3], x.q1[3], acc, 0, 0, 0);
}

// feature index held by accumulator register r of lane-half h (32x32 C/D layout), as in head_layout.h
DEV int ft(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// nn.Conv2d(padding_mode='reflect'): index -1 -> 1, n -> n - 2
DEV int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// ---- weight packing -----------------------------------------------------------------------------------------------------
// w [Cout][Cin][KS][KS] fp32 (PyTorch) -> packed[(tap * CB + cb) * CT + ct][hi: lane][8] | [lo: lane][8] halfs,
// lane l: row co = 32 ct + (l & 31), k = 8 (l >> 5) + j  <->  ci = 16 cb + k   (zero beyond Cin / Cout).
// flat (inputs with fewer than 8 channels, the 3-channel stem): the K dimension is (tap, ci) flattened, k = tap * Cin + ci, cut
// into chunks of 16 -- 10 chunks for the 7x7x3 stem instead of 49 mostly empty ones; packed[chunk * CT + ct][...].
// exact: the fp32 image of the EXACT form -- the same steps, [64 lanes][4 floats: j = 0..3] | [64 lanes][4 floats: j = 4..7], unscaled.
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, const int Cout, const int Cin, const int KS, const int CB,
                                        const int CT, const int flat, const int exact, uint16_t* __restrict__ packed) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one (step, lane, j)
    const long nstep = flat ? (long)CB * CT : (long)KS * KS * CB * CT;  // flat: CB = number of 16-deep chunks of K
    if (i >= nstep * 512) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long step = i >> 9;
    const int ct = (int)(step % CT), cb = (int)((step / CT) % CB);
    int tap = (int)(step / ((long)CT * CB));
    const int co = 32 * ct + (lane & 31);
    int ci = 16 * cb + 8 * (lane >> 5) + j;
    bool in_k = ci < Cin;
    if (flat) {
        const int k = 16 * cb + 8 * (lane >> 5) + j;
        in_k = k < KS * KS * Cin;
        tap = k / Cin;
        ci = k % Cin;
    }
    const float wv = (co < Cout && in_k) ? w[((long)co * Cin + ci) * KS * KS + tap] : 0.f;
    if (exact) {
        reinterpret_cast<float*>(packed)[step * (STEP_BYTES / 4) + (j >> 2) * 256 + lane * 4 + (j & 3)] = wv;
        return;
    }
    const float v = W_SCALE * wv;
    const _Float16 hi = (_Float16)v;                          // round to nearest, as the activations' hi
    const unsigned hw = __builtin_bit_cast(uint16_t, hi);
    const _Float16 lo = (_Float16)(v - (float)hi);
    uint16_t* dst = packed + step * (STEP_BYTES / 2);
    dst[lane * 8 + j] = (uint16_t)(hw & 0xFFFFu);
    dst[512 + lane * 8 + j] = __builtin_bit_cast(uint16_t, lo);
}

// The 7x7 stride-2 stem on <= 4 input channels (conv7x7_s2_stem_kernel): K = (ky, kx, c) with every kernel row padded to
// 8 columns x 4 channels, 14 chunks of 16: chunk (ky, p), lane half h, j -> kx = 4 p + 2 h + (j >> 2), c = j & 3 -- so that a
// lane's 8 values of a chunk are two NEIGHBOURING patch pixels x 4 channels, one aligned 16-byte LDS read.  [chunk][ct][hi | lo].
constexpr int STEM_CHUNKS = 14;
// exact: fp32, [64 lanes][4] | [64 lanes][4] as above, with the lane's eight values ordered (pixel 0: c0 c1, pixel 1: c0 c1 | pixel 0: c2
// c3, pixel 1: c2 c3) -- the two planes the exact stem keeps its patch in (conv7x7_s2_stem_kernel).
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, const int Cout, const int Cin, const int CT, const int exact, uint16_t* __restrict__ packed) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one (step, lane, j)
    if (i >= (long)STEM_CHUNKS * CT * 512) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long step = i >> 9;
    const int ct = (int)(step % CT), ch = (int)(step / CT);
    const int ky = ch >> 1, pq = ch & 1;
    const int co = 32 * ct + (lane & 31);
    if (exact) {        // value j of the lane: plane (j >> 2) holds channels 2 (j >> 2), 2 (j >> 2) + 1; pixel (j >> 1) & 1; channel within the pair j & 1
        const int kxe = 4 * pq + 2 * (lane >> 5) + ((j >> 1) & 1), ce = 2 * (j >> 2) + (j & 1);
        reinterpret_cast<float*>(packed)[step * (STEP_BYTES / 4) + (j >> 2) * 256 + lane * 4 + (j & 3)] =
            (co < Cout && kxe < 7 && ce < Cin) ? w[((long)co * Cin + ce) * 49 + ky * 7 + kxe] : 0.f;
        return;
    }
    const int kx = 4 * pq + 2 * (lane >> 5) + (j >> 2), c = j & 3;
    const float v = W_SCALE * ((co < Cout && kx < 7 && c < Cin) ? w[((long)co * Cin + c) * 49 + ky * 7 + kx] : 0.f);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    uint16_t* dst = packed + step * (STEP_BYTES / 2);
    dst[lane * 8 + j] = __builtin_bit_cast(uint16_t, hi);
    dst[512 + lane * 8 + j] = __builtin_bit_cast(uint16_t, lo);
}

// ---- the convolution ----------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;           // [N][H][W][Cin]
    const uint16_t* packed;
    const float* bias;        // [Cout] or nullptr
    float* y;                 // [N][Ho][Wo][Cout]
    float* stats;             // [N][tiles][Cout][3] per-workgroup-tile sum, sum of squares, and M2 = sum of (y - tile mean)^2 of the outputs (for the InstanceNorm behind), or nullptr
    int H, W, Cin, Ho, Wo, Cout, CB, CT;
    // the InstanceNorm in FRONT of the convolution, applied while the input is staged (3x3 stride-1 kernel only):
    const float* in_tab;      // [N][3][Cin] mean / scale / beta, or nullptr
    int in_act;               // 0 none, 1 ReLU
    // the InstanceNorm BEHIND it: the last workgroup of an (image, channel group) turns the tile sums into the table
    float* out_tab;           // [N][3][Cout] mean / gamma * rstd / beta, or nullptr (needs stats)
    const float* gamma;
    const float* beta;
    float eps;
    unsigned* counters;       // [N][gridDim.z], zero before the launch, left zero
    // a channel concatenation read in place (3x3 stride-1 kernel only): channels 0 .. CinA of x ([N][H][W][CinA]), the rest of x2
    // ([N][H][W][Cin - CinA]); x2 nullptr: CinA = Cin
    const float* x2;
    int CinA;
    // Range flag of the split-f16 form, or nullptr: ONE word that is set to 1 when an operand left the f16 range.  An activation
    // with |16 x| >= 65520 (or a weight with |4096 w| >= 65520) becomes an f16 infinity when it is split, its lo half the opposite
    // infinity, and every accumulator it meets ends up NaN -- nothing is silently wrong, the result is loudly non-finite.  That is
    // looked for where it costs nothing: the InstanceNorm table's sums (finalize_if_last: a non-finite sum or sum of squares of any
    // channel), and the accumulators themselves in a convolution with no norm behind it.  The host then runs the layer chain
    // again in the exact fp32 form (exact = 1).  A non-finite INPUT sets the flag too (the exact form then returns
    // what fp32 arithmetic makes of it).  The word is only ever written with 1; the caller zeroes it.
    unsigned* flag;
    int tile_h, tile_w;       // the kernel's workgroup tile of output pixels (tile_h x tile_w; tile_h = 0: tile_w consecutive pixels): how many
                              // pixels tile k's statistics cover (tile_pixels), set by the launcher
    int exact;                // 1: the EXACT form (fp32 operands on the fp32 MFMA; `packed` is the fp32 image); launch dispatch only
};

// valid output pixels of workgroup tile k (what its statistics cover)
DEV int tile_pixels(const ConvArgs& a, const int k) {
    if (a.tile_h == 0) return min(a.tile_w, a.Ho * a.Wo - k * a.tile_w);
    const int tiles_x = (a.Wo + a.tile_w - 1) / a.tile_w, ty = k / tiles_x, tx = k - ty * tiles_x;
    return min(a.tile_h, a.Ho - ty * a.tile_h) * min(a.tile_w, a.Wo - tx * a.tile_w);
}
// (count, mean, M2) of two disjoint sets -> of their union (Chan et al.), in double
struct Moments { double n, mean, m2; };
DEV Moments merge(const Moments& x, const Moments& y) {
    if (y.n == 0) return x;
    if (x.n == 0) return y;
    const double n = x.n + y.n, d = y.mean - x.mean;
    return Moments{n, x.mean + d * (y.n / n), x.m2 + y.m2 + d * d * (x.n * y.n / n)};
}

// Last-arriving workgroup of (image n, channel group ct0 .. ct0 + COT): every workgroup has written its tile's (sum, sum of squares,
// M2); the one whose atomic ticket is the last adds the tiles of its COT * 32 channels in double, in tile order (so the result does
// not depend on which workgroup that is), and writes mean / gamma * rstd / beta.  256 threads: 256 / (COT * 32) tile lanes per channel.
// The variance is E[y^2] - mean^2 from the float32 tile sums where that is well conditioned (var >= 1e-3 mean^2: at most three of
// float32's seven digits cancel), and otherwise assembled from the tiles' sums of squares ABOUT THEIR MEANS (tile_stats) by Chan's
// pairwise merge: on a channel that is nearly constant over the image (a one-hot source image behind a large InstanceNorm scale)
// the difference cancels to a few roundings of E[y^2] -- measured 3e-4 of the output range on such a frame, 1e-6 with the merge.
template <int COT>
DEV void finalize_if_last(const ConvArgs& a, const int n, const int ct0, const int ntiles, double* red /* [3][256] doubles of LDS */) {
    // No agent-scope fence here: a release fence would write back this XCD's whole L2 (the convolution's output has just
    // dirtied it; measured: the encoder 1.35 -> 1.9 ms).  The tile sums were stored with agent-scope atomic stores (sc1:
    // written through to the device's coherence point).  What orders them before the ticket is an EXPLICIT s_waitcnt vmcnt(0)
    // in every wave, in front of the barrier: on gfx9 a workgroup-scope release fence only waits on lgkmcnt (round 3 relied on
    // it and the built code had its first vmcnt(0) behind the ticket atomic, so the ticket could become visible while another
    // wave's write-through stores were still in flight to their L2 channel).  With the wait, every wave's stores are
    // acknowledged before it enters the barrier, the ticket (a relaxed agent-scope atomic) is issued behind the barrier, and
    // the last workgroup reads the sums back with agent-scope atomic loads (past its own L2).  tools/isa_ticket_release.py
    // checks the built ISA for exactly this order (tests/test_abi.py).
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&a.counters[(size_t)n * gridDim.z + blockIdx.z], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev + 1u == gridDim.x ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    constexpr int NCH = COT * 32, LANES = 256 / NCH;
    const int ch = threadIdx.x % NCH, kl = threadIdx.x / NCH, co = 32 * ct0 + ch;
    double ts = 0, tq = 0;
    if (co < a.Cout)
        for (int k0 = kl; k0 < ntiles; k0 += 8 * LANES) {     // eight tiles' loads in flight, added in tile order
            float vs[8], vq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j * LANES;
                const float* o = a.stats + (((size_t)n * ntiles + (k < ntiles ? k : kl)) * a.Cout + co) * 3;
                vs[j] = __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vq[j] = __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j * LANES < ntiles) { ts += (double)vs[j]; tq += (double)vq[j]; }
        }
    red[threadIdx.x] = ts;
    red[256 + threadIdx.x] = tq;
    __syncthreads();
    __shared__ unsigned s_ill;
    if (threadIdx.x == 0) s_ill = 0u;
    const double hw = (double)a.Ho * (double)a.Wo;
    double mean = 0, var = 0, sum_s = 0, sum_q = 0;
    if (kl == 0 && co < a.Cout) {
#pragma unroll
        for (int k = 0; k < LANES; ++k) { sum_s += red[k * NCH + ch]; sum_q += red[256 + k * NCH + ch]; }
        mean = sum_s / hw;
        var = sum_q / hw - mean * mean;                         // biased variance, as InstanceNorm2d normalises with
    }
    __syncthreads();
    const bool ill = kl == 0 && co < a.Cout && !(var >= 1e-3 * mean * mean);
    if (ill) s_ill = 1u;                                        // (every writer writes the same value)
    __syncthreads();
    if (s_ill) {
        // some channel of the group is ill-conditioned (or not finite): the tiles' M2 merged as Chan et al., in tile order -- a second
        // pass over the tile rows that ordinary frames never take
        Moments mine{0, 0, 0};
        if (co < a.Cout)
            for (int k = kl; k < ntiles; k += LANES) {
                const float* o = a.stats + (((size_t)n * ntiles + k) * a.Cout + co) * 3;
                const double cnt = (double)tile_pixels(a, k);
                const double sk = (double)__hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mine = merge(mine, Moments{cnt, cnt > 0 ? sk / cnt : 0.0, (double)__hip_atomic_load(o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)});
            }
        red[threadIdx.x] = mine.n;
        red[256 + threadIdx.x] = mine.mean;
        red[512 + threadIdx.x] = mine.m2;
        __syncthreads();
        if (ill) {
            Moments all{0, 0, 0};
#pragma unroll
            for (int k = 0; k < LANES; ++k) all = merge(all, Moments{red[k * NCH + ch], red[256 + k * NCH + ch], red[512 + k * NCH + ch]});
            var = all.m2 / hw;
        }
    }
    if (kl == 0 && co < a.Cout) {
        if (!(var > 0)) var = 0;
        const float g = a.gamma[co] / sqrtf((float)var + a.eps);
        a.out_tab[((size_t)n * 3 + 0) * a.Cout + co] = (float)mean;
        a.out_tab[((size_t)n * 3 + 1) * a.Cout + co] = g;
        a.out_tab[((size_t)n * 3 + 2) * a.Cout + co] = a.beta[co];
        // an operand beyond the f16 range (ConvArgs::flag): the channel's sums are non-finite
        if (a.flag && !(fabs(sum_s) < __builtin_inf() && fabs(sum_q) < __builtin_inf())) __hip_atomic_store(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) __hip_atomic_store(&a.counters[(size_t)n * gridDim.z + blockIdx.z], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

// Per-channel sum, sum of squares and M2 (sum of squares about the tile's mean) of a workgroup's output tile, from the accumulators (the InstanceNorm
// that follows every convolution but the last would otherwise re-read the whole tensor for them).  Accumulator register r of half h holds channel
// 32 c + ft(r, h) of the lane's pixel.  The wave's PT pixel tiles are added per lane; the 32 lanes of a half are summed with
// DPP moves (rotations by 8, 4, 2, 1 inside each 16-lane row, then row_bcast15 carries row 0's total into row 1 and row 2's
// into row 3: five VALU adds per register, no LDS traffic); lanes 16 and 48 then hold the two halves' totals, and the four
// waves meet in LDS in a fixed order.  red: [WAVES][COT * 32][2] floats.
template <int CTRL, int ROW_MASK>
DEV float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
DEV float half_sum(float v) {             // valid in lanes 16..31 (half 0) and 48..63 (half 1)
    v = dpp_add<0x128, 0xF>(v);           // row_ror:8
    v = dpp_add<0x124, 0xF>(v);           // row_ror:4
    v = dpp_add<0x122, 0xF>(v);           // row_ror:2
    v = dpp_add<0x121, 0xF>(v);           // row_ror:1  -> every lane holds its row's total
    return dpp_add<0x142, 0xA>(v);        // row_bcast:15 into rows 1 and 3
}

template <int COT, int NP = 2>
DEV void tile_stats(const f32x16 (&acc)[NP][COT], const bool (&valid)[NP], float* red, const ConvArgs& a, const int n, const int tile,
                    const int ntiles, const int ct0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    // the wave's valid pixels (the same in both lane halves): its mean per channel, and the squares ABOUT that mean (see finalize_if_last)
    const int nw = __popc((unsigned)__ballot(valid[0])) + (NP > 1 ? __popc((unsigned)__ballot(valid[NP - 1])) : 0);
    const float inv = nw > 0 ? 1.f / (float)nw : 0.f;
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v0 = valid[0] ? acc[0][c][r] : 0.f, v1 = (NP > 1 && valid[NP - 1]) ? acc[NP - 1][c][r] : 0.f;
            const float s = half_sum(v0 + v1), q = half_sum(fmaf(v1, v1, v0 * v0));
            const float s_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 16));
            const float s_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 48));
            const float m = (half ? s_hi : s_lo) * inv;
            const float d0 = valid[0] ? acc[0][c][r] - m : 0.f, d1 = (NP > 1 && valid[NP - 1]) ? acc[NP - 1][c][r] - m : 0.f;
            const float m2 = half_sum(fmaf(d1, d1, d0 * d0));
            if ((lane & 31) == 16) {
                const int ch = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * half;
                red[(wave * COT * 32 + ch) * 3 + 0] = s;
                red[(wave * COT * 32 + ch) * 3 + 1] = q;
                red[(wave * COT * 32 + ch) * 3 + 2] = m2;
            }
        }
    int* const cnt = reinterpret_cast<int*>(red + WAVES * COT * 32 * 3);
    if (lane == 0) cnt[wave] = nw;
    __syncthreads();
    for (int ch = threadIdx.x; ch < COT * 32; ch += WAVES * 64) {
        float s = 0.f, q = 0.f, ntot = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { s += red[(w * COT * 32 + ch) * 3]; q += red[(w * COT * 32 + ch) * 3 + 1]; ntot += (float)cnt[w]; }
        // M2 of the tile = sum over the waves of M2_w + n_w (mean_w - mean)^2   (float32: every term is a sum about a nearby mean)
        const float mean = ntot > 0.f ? s / ntot : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float cw = (float)cnt[w];
            const float d = cw > 0.f ? red[(w * COT * 32 + ch) * 3] / cw - mean : 0.f;
            m2 += red[(w * COT * 32 + ch) * 3 + 2] + cw * d * d;
        }
        const int co = 32 * ct0 + ch;
        if (co < a.Cout) {
            // agent-scope stores: written through to the device's coherence point, so that the last workgroup (possibly on
            // another XCD, behind another L2) can read them without any cache being flushed -- see finalize_if_last
            float* o = a.stats + (((size_t)n * ntiles + tile) * a.Cout + co) * 3;
            __hip_atomic_store(o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 2, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// A convolution with no InstanceNorm behind it (the encoder's output convolution): the range flag (ConvArgs::flag) is raised
// from the accumulators themselves -- 0 * v is NaN exactly when v is not finite.  Pixels beyond the image are copies of real
// ones (clamped positions), so they raise no false flag.
template <int COT, int NP>
DEV void flag_nonfinite(const f32x16 (&acc)[NP][COT], const ConvArgs& a) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) t = fmaf(acc[p][c][r], 0.f, t);
    if (t != t) __hip_atomic_store(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// this lane's 8 input channels (16 cb + 8 half ..) of pixel (iy, ix)
DEV void load_px(const ConvArgs& a, const int n, const int iy, const int ix, const int cb, const int half, f32x4& v0, f32x4& v1) {
    const f32x4* q = reinterpret_cast<const f32x4*>(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.Cin + 16 * cb + 8 * half);
    v0 = q[0];
    v1 = q[1];
}

// narrow inputs (flat K, see pack_conv_weight_kernel): this lane's 8 values k = 16 chunk + 8 half + j -> (tap, ci) of the output
// pixel (oy, ox); eight scalar loads
template <int KS, int STRIDE>
DEV void load_flat(const ConvArgs& a, const int n, const int oy, const int ox, const int chunk, const int half, f32x4& v0, f32x4& v1) {
    constexpr int PAD = KS / 2;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * chunk + 8 * half + j;
        const int tap = k / a.Cin, ci = k - tap * a.Cin;
        const int ky = tap / KS, kx = tap - ky * KS;
        v[j] = 0.f;
        if (tap < KS * KS)
            v[j] = a.x[(((size_t)n * a.H + reflect(oy * STRIDE + ky - PAD, a.H)) * a.W + reflect(ox * STRIDE + kx - PAD, a.W)) * a.Cin + ci];
    }
    v0 = f32x4{v[0], v[1], v[2], v[3]};
    v1 = f32x4{v[4], v[5], v[6], v[7]};
}

template <int KS, int STRIDE, int COT, bool NARROW, bool EXACT = false>
__global__ void __launch_bounds__(WAVES * 64) conv2d_nhwc_kernel(const ConvArgs a) {
    constexpr int PAD = KS / 2;
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[2][COT * STEP_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, px = lane & 31, half = lane >> 5;
    const int n = blockIdx.y, ct0 = blockIdx.z * COT;
    const int howo = a.Ho * a.Wo;
    int oy[PT], ox[PT];
    bool valid[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int p = ((int)blockIdx.x * WAVES + wave) * (PT * 32) + t * 32 + px;
        valid[t] = p < howo;
        const int pc = valid[t] ? p : howo - 1;
        oy[t] = pc / a.Wo;
        ox[t] = pc % a.Wo;
    }
    f32x16 acc[PT][COT];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

    const int nchunk = NARROW ? a.CB : KS * KS * a.CB;            // narrow: CB counts the chunks of the flattened K
    // every thread moves 16 bytes of the next chunk's weights per 256-thread pass: COT * 2 KB = COT * 128 pieces
    auto wsrc = [&](int q, int piece) {
        return reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.packed) + ((size_t)q * a.CT + ct0) * STEP_BYTES) + piece;
    };
    constexpr int PIECES = COT * (STEP_BYTES / 16);
    constexpr int WPASS = (PIECES + WAVES * 64 - 1) / (WAVES * 64);
    u32x4 wreg[WPASS];
    f32x4 xin[PT][2];
    // an InstanceNorm (+ ReLU) in FRONT of the convolution (a.in_tab; wide inputs only): the chunk's 8 channels of mean / scale / beta
    // travel with its pixels and are applied where the pixels are split, as nhwc_norm_apply_kernel computes it
    f32x4 tin[3][2];
    const float act_floor = (a.in_tab && a.in_act == 1) ? 0.f : -__builtin_inff();
    auto fetch = [&](int q) {
#pragma unroll
        for (int s = 0; s < WPASS; ++s) {
            const int piece = s * (WAVES * 64) + (int)threadIdx.x;
            if (piece < PIECES) wreg[s] = *wsrc(q, piece);
        }
        if constexpr (NARROW) {
#pragma unroll
            for (int t = 0; t < PT; ++t) load_flat<KS, STRIDE>(a, n, oy[t], ox[t], q, half, xin[t][0], xin[t][1]);
        } else {
            const int tap = q / a.CB, cb = q - tap * a.CB;
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int t = 0; t < PT; ++t)
                load_px(a, n, reflect(oy[t] * STRIDE + ky - PAD, a.H), reflect(ox[t] * STRIDE + kx - PAD, a.W), cb, half, xin[t][0], xin[t][1]);
            if (a.in_tab) {
                const float* tb = a.in_tab + (size_t)n * 3 * a.Cin + 16 * cb + 8 * half;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    tin[j][0] = *reinterpret_cast<const f32x4*>(tb + (size_t)j * a.Cin);
                    tin[j][1] = *reinterpret_cast<const f32x4*>(tb + (size_t)j * a.Cin + 4);
                }
            }
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int s = 0; s < WPASS; ++s) {
            const int piece = s * (WAVES * 64) + (int)threadIdx.x;
            if (piece < PIECES) reinterpret_cast<u32x4*>(wbuf[buf])[piece] = wreg[s];
        }
    };
    fetch(0);
    park(0);
    __syncthreads();
    for (int q = 0; q < nchunk; ++q) {
        const int buf = q & 1;
        typename std::conditional<EXACT, FragX, Frag>::type b[PT];
        if constexpr (!NARROW) {
            if (a.in_tab) {
#pragma unroll
                for (int t = 0; t < PT; ++t)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            xin[t][h2][k] = fmaxf(fmaf(xin[t][h2][k] - tin[0][h2][k], tin[1][h2][k], tin[2][h2][k]), act_floor);
            }
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            if constexpr (EXACT) { b[t].q0 = xin[t][0]; b[t].q1 = xin[t][1]; }
            else b[t] = make_frag(xin[t][0], xin[t][1]);
        }
        if (q + 1 < nchunk) fetch(q + 1);                 // next chunk's loads fly while this chunk's MFMAs run
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            const u32x4* wl = reinterpret_cast<const u32x4*>(wbuf[buf] + c * STEP_BYTES);
            if constexpr (EXACT) {
                const f32x4 w0 = __builtin_bit_cast(f32x4, wl[lane]), w1 = __builtin_bit_cast(f32x4, wl[64 + lane]);
#pragma unroll
                for (int t = 0; t < PT; ++t) mfma_exact(acc[t][c], w0, w1, b[t]);
            } else {
                const h8 wh = __builtin_bit_cast(h8, wl[lane]), wlo = __builtin_bit_cast(h8, wl[64 + lane]);
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, b[t].hi, acc[t][c], 0, 0, 0);
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b[t].lo, acc[t][c], 0, 0, 0);
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b[t].hi, acc[t][c], 0, 0, 0);
                }
            }
        }
        if (q + 1 < nchunk) park(buf ^ 1);
        __syncthreads();
    }
    // epilogue: accumulator register r of half h holds output channel 32 ct + ft(r, h): four runs of 4 consecutive channels
    if constexpr (!EXACT) {
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int c = 0; c < COT; ++c) acc[t][c] *= ACC_UNSCALE;
    }
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * (ct0 + c) + ft(r, half);
                const float bv = co < a.Cout ? a.bias[co] : 0.f;
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[t][c][r] += bv;
            }
    }
    if (a.stats) tile_stats<COT>(acc, valid, reinterpret_cast<float*>(wbuf[0]), a, n, (int)blockIdx.x, (int)gridDim.x, ct0);
    if (a.out_tab) {
        __shared__ double fin_red[768];
        finalize_if_last<COT>(a, n, ct0, (int)gridDim.x, fin_red);
    } else if (a.flag) flag_nonfinite<COT, PT>(acc, a);
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (!valid[t]) continue;
        float* yp = a.y + ((size_t)n * howo + (size_t)oy[t] * a.Wo + ox[t]) * a.Cout;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            const int co0 = 32 * (ct0 + c);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + 8 * g + 4 * half;
                if (co + 4 > a.Cout) continue;                    // Cout is a multiple of 4; the last tile may be partial
                const f32x4 v = {acc[t][c][4 * g + 0], acc[t][c][4 * g + 1], acc[t][c][4 * g + 2], acc[t][c][4 * g + 3]};
                *reinterpret_cast<f32x4*>(yp + co) = v;
            }
        }
    }
}

// ---- 3x3, stride 1: the bulk of the encoder (27 of its 35 convolutions) ------------------------------------------------
// The direct kernel above pays one global-memory round trip per (tap, 16 channels) chunk for 6 MFMAs of work.  Here a
// workgroup owns an 8-row x 32-column output tile; per 16-channel block it stages the (8+2) x (32+2) input patch ONCE into
// LDS, already split into f16 hi / lo (so the nine taps read their B operand straight from LDS, no conversion in the loop),
// together with the nine taps' weights, double-buffered against the MFMAs of the previous block: 54 x COT MFMAs per wave
// between two barriers, global loads 9x fewer and all in flight together.
constexpr int TW = 32, PW = TW + 2;                         // output tile columns; input patch columns (one-pixel halo)
// rows per wave RW: 2 (an 8-row tile per workgroup) or 1 (4 rows): small images take the 4-row tile, which doubles the workgroups of
// a layer that cannot fill the chip anyway and halves every workgroup's serial chain (conv3x3_rows())
__host__ __device__ constexpr int tile_rows(int rw) { return WAVES * rw; }
constexpr int PPX = 80;                                     // bytes per patch pixel: 16 hi halfs | 16 lo halfs | 16 pad (bank spread)
// input patch of a tile: (STRIDE * rows + 3 - STRIDE) x (STRIDE * 32 + 3 - STRIDE) pixels.  Stride 2 keeps the even and the odd
// patch columns in separate planes ([row][parity][33 columns]), so that a tap's 32 lanes -- which read every second column --
// still walk consecutive pixels of one plane (80-byte pitch, conflict-free) instead of every second pixel of one row
__host__ __device__ constexpr int patch_rows(int rw, int stride) { return stride * tile_rows(rw) + 3 - stride; }
__host__ __device__ constexpr int patch_cols(int stride) { return stride == 1 ? PW : 66; }              // stride 2: 65 columns in 2 planes of 33
__host__ __device__ constexpr int patch_bytes(int rw, int stride = 1) { return patch_rows(rw, stride) * patch_cols(stride) * PPX; }    // 27 200 / 16 320 (stride 1: 8 / 4 rows), 47 520 (stride 2, 4 rows)

// KSPLIT = 2: eight waves, two halves of four; half kh runs the channel blocks kh, kh + 2, ... through its own pair of patch buffers
// and the halves' accumulators meet in LDS at the end (lower + upper, a fixed order).  For the layers whose grid cannot fill the chip
// (32 x 32 images: 192 workgroups): the serial chain of a wave halves, and every SIMD has a second wave to issue from while the
// first waits.
template <int COT, int RW, int STRIDE = 1, int KSPLIT = 1, bool CAT = false, bool EXACT = false>
__global__ void __launch_bounds__(WAVES * KSPLIT * 64) conv3x3_s1_nhwc_kernel(const ConvArgs a) {
    constexpr int TH = tile_rows(RW), PH = patch_rows(RW, STRIDE), PATCH_BYTES = patch_bytes(RW, STRIDE), PT = RW;     // (PT shadows the direct kernel's pixel-tile count)
    constexpr int PCOLS = STRIDE == 1 ? PW : 65;             // patch columns actually staged
    // LDS position of patch pixel (row, col)
    auto ppos = [](int row, int col) { return STRIDE == 1 ? row * PW + col : (row * 2 + (col & 1)) * 33 + (col >> 1); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int kh = KSPLIT == 1 ? 0 : (int)(threadIdx.x >> 8);                  // K half; ht: thread index within the half
    const int ht = (int)threadIdx.x & (WAVES * 64 - 1);
    unsigned char* const patch0 = smem + kh * 2 * PATCH_BYTES;                 // [KSPLIT][2][PATCH_BYTES]
    float* const itab = reinterpret_cast<float*>(smem + KSPLIT * 2 * PATCH_BYTES);      // [3][Cin] when a.in_tab
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & (WAVES - 1), px = lane & 31, half = lane >> 5;
    const int tiles_x = (a.Wo + TW - 1) / TW;
    const int ty0 = ((int)blockIdx.x / tiles_x) * TH, tx0 = ((int)blockIdx.x % tiles_x) * TW;
    const int n = blockIdx.y, ct0 = blockIdx.z * COT;

    // EXACT: BLOCKED summation.  One fp32 FMA chain over all K = 9 Cin terms (2 304 at 256 channels) drifts ~K / sqrt(2) roundings
    // (of a term's size) from the exact sum; the reference's CPU convolutions block their sums, and what the head behind the encoder
    // amplifies is the DISTANCE between the two roundings.  So every 16-channel block's 144 terms are summed from zero in a block
    // accumulator and added to the running total once per block: by the same count ~K / sqrt(2 CB) + sqrt(K CB / 2) roundings, a
    // quarter of the single chain's at 256 channels, for 16 registers per accumulator tile and 16 adds per 72 MFMAs.  Measured on
    // the 3 x 512 x 512 reference vector and the config-5-sized end-to-end fixture (profiles/r06/e_encoder_summation_variants.txt):
    // one chain 4.0e-6 mean from float64 and 1.7e-4 from the reference on depth behind the head; four chains by tap 2.9e-6 / 9.7e-5;
    // this 2.6e-6 / 7.7e-5 -- the reference's own float32 sits 2.6e-6 from float64 -- and it is the fastest of them (fewest registers).
    f32x16 acc[PT][COT], blk_acc[PT][COT];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

    // The A operand (weights) goes from global memory (the layer's packed image: <= 2.4 MB, L2-resident, every workgroup reads
    // the same bytes) straight into registers: a lane's hi and lo quads of every (tap, output tile) of one 16-channel block,
    // 9 x COT x 2 x 4 registers.  A tap's registers are re-loaded with the NEXT block's weights right after the tap's MFMAs
    // have been issued, so every load has eight taps of matrix work (~a block) to arrive.  (Round 2 staged the weights through
    // LDS like the patch: with COT = 1 a tap then cost 6 ds_read_b128 for 6 MFMAs, and four waves x 8 LDS cycles per read fill
    // the 32 cycles an MFMA gives them -- the loop ran at 2.3 us per block against 0.7 us of MFMAs.  Now LDS carries the
    // B operand only: 4 reads per tap.)
    u32x4 wreg[9][COT][2];
    auto wload = [&](int tap, int cb) {
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            const u32x4* wl = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.packed) +
                                                             (((size_t)tap * a.CB + cb) * a.CT + ct0 + c) * STEP_BYTES);
            wreg[tap][c][0] = wl[lane];
            wreg[tap][c][1] = wl[64 + lane];
        }
    };
    // staging work of one thread per channel block: PPASS patch items (pixel, 4-channel quad).  TWO register sets: while block
    // cb is in its MFMAs, block cb + 1's values sit in one set (loaded a block ago) and are converted and written to LDS ONE ITEM
    // AFTER EACH TAP's MFMAs -- the ~25 vector instructions of an item run in the shadow of the tap's 6 x COT matrix
    // instructions (the f16 MFMA leaves the issue port free) -- while block cb + 2's loads fill the other set.  (Measured per
    // 256-channel 32x32 convolution, graph-timed: 30.7 us with the whole conversion behind the MFMAs, of which 15.1 us matrix work
    // + LDS reads, 7.6 us conversion + LDS writes, 1.4 us issuing loads, 1.7 us barriers; tools/probes/conv_layer_time.py.)
    constexpr int PITEMS = PH * PCOLS * 4, PPASS = (PITEMS + WAVES * 64 - 1) / (WAVES * 64);
    static_assert(PPASS <= 18, "at most two patch items per tap");
    f32x4 preg[2][PPASS];
    auto fetch = [&](int cb, auto SET) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int s = 0; s < PPASS; ++s) {
            const int item = s * (WAVES * 64) + ht;
            if (item < PITEMS) {
                const int pp = item >> 2, qd = item & 3;
                const int iy = reflect(STRIDE * ty0 + pp / PCOLS - 1, a.H), ix = reflect(STRIDE * tx0 + pp % PCOLS - 1, a.W);
                // tiles may hang far over the image, where the reflection itself leaves it: clamp (those outputs are never written)
                const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
                if constexpr (CAT) {
                // the decoder's [up, skip] concatenations are never materialised: a block's channels come from one tensor or the
                // other (an instantiation of its own: the pixel stride changes with the block, which costs the other layers' loops
                // an address multiplication per item)
                const bool second = 16 * cb >= a.CinA;
                const float* const src = second ? a.x2 : a.x;
                const int cs = second ? a.Cin - a.CinA : a.CinA, c0 = 16 * cb - (second ? a.CinA : 0);
                preg[S][s] = *reinterpret_cast<const f32x4*>(src + (((size_t)n * a.H + cy) * a.W + cx) * cs + c0 + 4 * qd);
            } else {
                preg[S][s] = *reinterpret_cast<const f32x4*>(a.x + (((size_t)n * a.H + cy) * a.W + cx) * a.Cin + 16 * cb + 4 * qd);
            }
            }
        }
    };
    // item s of block pcb: normalise (optional), scale, split into hi / lo, write to patch buffer `buf`
    auto park_item = [&](int s, int buf, int pcb, auto SET) {
        constexpr int S = decltype(SET)::value;
        unsigned char* const pb = patch0 + buf * PATCH_BYTES;
        const int item = s * (WAVES * 64) + ht;
        if (item < PITEMS) {
            const int pp = item >> 2, qd = item & 3;
            f32x4 v = preg[S][s];
            if (a.in_tab) {                                  // the InstanceNorm (+ ReLU) in front of this convolution, as nhwc_norm_apply_kernel computes it
                const int c0 = 16 * pcb + 4 * qd;
                const f32x4 mu = *reinterpret_cast<const f32x4*>(itab + c0), sc = *reinterpret_cast<const f32x4*>(itab + a.Cin + c0),
                            be = *reinterpret_cast<const f32x4*>(itab + 2 * a.Cin + c0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t = fmaf(v[k] - mu[k], sc[k], be[k]);
                    v[k] = a.in_act == 1 ? fmaxf(t, 0.f) : t;
                }
            }
            if constexpr (EXACT) {       // the pixel's 16 channels as they are: 64 bytes, quad qd at 16 qd
                *reinterpret_cast<f32x4*>(pb + ppos(pp / PCOLS, pp % PCOLS) * PPX + qd * 16) = v;
            } else {
                v *= X_SCALE;
                const unsigned h0 = pk_hi(v[0], v[1]), h1 = pk_hi(v[2], v[3]);
                unsigned* d = reinterpret_cast<unsigned*>(pb + ppos(pp / PCOLS, pp % PCOLS) * PPX + qd * 8);
                d[0] = h0; d[1] = h1;
                d[8] = lo_pair(h0, v[0], v[1]); d[9] = lo_pair(h1, v[2], v[3]);           // lo block starts 32 bytes in
            }
        }
    };
    // the 54 x COT MFMAs of the block staged in `buf`: B operands of tap 0, then per tap: read the next tap's B operands, run this
    // tap's MFMAs (the three products of one accumulator are issued COT * PT MFMAs apart: back-to-back they would wait on each
    // other), fetch this tap's weights of block cb_next into the registers just used, and park one item of block cb_next
    auto compute = [&](int buf, int cb_next, auto SET) {
        const unsigned char* const pb = patch0 + buf * PATCH_BYTES;
        typename std::conditional<EXACT, FragX, Frag>::type b[2][PT];
        auto read_tap = [&](int tap, int slot) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                if constexpr (EXACT) {      // this half's eight channels: 32 contiguous bytes of the pixel's 64
                    const unsigned char* q = pb + ppos(STRIDE * (RW * wave + t) + ky, STRIDE * px + kx) * PPX + half * 32;
                    b[slot][t].q0 = *reinterpret_cast<const f32x4*>(q);
                    b[slot][t].q1 = *reinterpret_cast<const f32x4*>(q + 16);
                } else {
                    const unsigned char* q = pb + ppos(STRIDE * (RW * wave + t) + ky, STRIDE * px + kx) * PPX + half * 16;
                    b[slot][t].hi = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(q));
                    b[slot][t].lo = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(q + 32));
                }
            }
        };
        read_tap(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < 9) read_tap(tap + 1, cur ^ 1);
            if constexpr (EXACT) {
#pragma unroll
                for (int c = 0; c < COT; ++c) {
                    const f32x4 w0 = __builtin_bit_cast(f32x4, wreg[tap][c][0]), w1 = __builtin_bit_cast(f32x4, wreg[tap][c][1]);
#pragma unroll
                    for (int t = 0; t < PT; ++t) {
                        if (tap == 0) {         // the block's chain starts from zero
#pragma unroll
                            for (int r = 0; r < 16; ++r) blk_acc[t][c][r] = 0.f;
                        }
                        mfma_exact(blk_acc[t][c], w0, w1, b[cur][t]);
                    }
                }
                if (tap == 8) {                 // the block's sums into the running total
#pragma unroll
                    for (int c = 0; c < COT; ++c)
#pragma unroll
                        for (int t = 0; t < PT; ++t) acc[t][c] += blk_acc[t][c];
                }
            } else {
                h8 wh[COT], wlo[COT];
#pragma unroll
                for (int c = 0; c < COT; ++c) { wh[c] = __builtin_bit_cast(h8, wreg[tap][c][0]); wlo[c] = __builtin_bit_cast(h8, wreg[tap][c][1]); }
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int t = 0; t < PT; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[c], b[cur][t].hi, acc[t][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int t = 0; t < PT; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[c], b[cur][t].lo, acc[t][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int t = 0; t < PT; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[c], b[cur][t].hi, acc[t][c], 0, 0, 0);
            }
            if (cb_next >= 0) {
                wload(tap, cb_next);
                if (tap < PPASS) park_item(tap, buf ^ 1, cb_next, SET);
                if (9 + tap < PPASS) park_item(9 + tap, buf ^ 1, cb_next, SET);
            }
        }
    };
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    // this half's channel blocks: blk(i) = kh + KSPLIT * i, i < nblk (the launcher takes KSPLIT = 2 only for an even CB)
    const int nblk = a.CB / KSPLIT;
    auto blk = [&](int i) { return kh + KSPLIT * i; };
    fetch(blk(0), S0);
    if (nblk > 1) fetch(blk(1), S1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wload(tap, blk(0));
    if (a.in_tab) {
        for (int i = threadIdx.x; i < 3 * a.Cin; i += WAVES * KSPLIT * 64) itab[i] = a.in_tab[(size_t)n * 3 * a.Cin + i];
        __syncthreads();
    }
#pragma unroll
    for (int s2 = 0; s2 < PPASS; ++s2) park_item(s2, 0, blk(0), S0);
    __syncthreads();
    for (int i = 0; i < nblk; i += 2) {
        // even block: staged in buffer 0; set 1 holds block i + 1 (in flight since the block before); set 0 is free for block i + 2
        if (i + 2 < nblk) fetch(blk(i + 2), S0);
        compute(0, i + 1 < nblk ? blk(i + 1) : -1, S1);
        __syncthreads();
        if (i + 1 >= nblk) break;
        if (i + 3 < nblk) fetch(blk(i + 3), S1);
        compute(1, i + 2 < nblk ? blk(i + 2) : -1, S0);
        __syncthreads();
    }
    if constexpr (KSPLIT == 2) {
        // the upper half hands its sums over ([wave][t][c][r][lane] floats, past what the epilogue's reductions use) and is done;
        // barriers from here on count the four surviving waves only (s_barrier ignores waves that have ended)
        float* const xch = reinterpret_cast<float*>(smem + 16384) + (size_t)wave * (PT * COT * 16 * 64);
        if (kh == 1) {
#pragma unroll
            for (int t = 0; t < PT; ++t)
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((t * COT + c) * 16 + r) * 64 + lane] = acc[t][c][r];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][c][r] += xch[((t * COT + c) * 16 + r) * 64 + lane];
    }
    const int ox = tx0 + px;
    bool valid[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) valid[t] = (ty0 + RW * wave + t) < a.Ho && ox < a.Wo;
    if constexpr (!EXACT) {
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int c = 0; c < COT; ++c) acc[t][c] *= ACC_UNSCALE;
    }
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * (ct0 + c) + ft(r, half);
                const float bv = co < a.Cout ? a.bias[co] : 0.f;
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[t][c][r] += bv;
            }
    }
    if (a.stats) tile_stats<COT, RW>(acc, valid, reinterpret_cast<float*>(smem), a, n, (int)blockIdx.x, (int)gridDim.x, ct0);
    if (a.out_tab) finalize_if_last<COT>(a, n, ct0, (int)gridDim.x, reinterpret_cast<double*>(smem + 8192));
    else if (a.flag) flag_nonfinite<COT, PT>(acc, a);
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (!valid[t]) continue;
        float* yp = a.y + (((size_t)n * a.Ho + ty0 + RW * wave + t) * a.Wo + ox) * a.Cout;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            const int co0 = 32 * (ct0 + c);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + 8 * g + 4 * half;
                if (co + 4 > a.Cout) continue;
                const f32x4 v = {acc[t][c][4 * g + 0], acc[t][c][4 * g + 1], acc[t][c][4 * g + 2], acc[t][c][4 * g + 3]};
                *reinterpret_cast<f32x4*>(yp + co) = v;
            }
        }
    }
}

// ---- 7x7, stride 2, <= 4 input channels: the stem ------------------------------------------------------------------------
// The direct kernel gathers this layer's operands with eight 4-byte loads per lane and chunk from pixels 24 bytes apart (72 us at
// 3 x 512 x 512 for 5 us of matrix work).  Here a workgroup owns an 8 x 32 output tile x COT x 32 output channels: its
// (2*8+5) x (2*32+5)-pixel input patch is loaded once, pixel by pixel (coalesced), scaled and split into hi / lo planes of 8 bytes
// per pixel (4 channels, the 4th zero); the whole packed weight image of its output tiles (14 chunks, 28 KB per tile) is copied
// into LDS beside it; then 14 chunks x 3 MFMAs per (pixel tile, output tile), every operand one aligned ds_read_b128
// (pack_stem_weight_kernel's K order).  No loop over channel blocks, one barrier.
constexpr int STEM_TH = 8, STEM_PH = 2 * STEM_TH + 5, STEM_PCOLS = 2 * TW + 5, STEM_PITCH = 72;     // patch: 21 rows x 69 (+3 zero) columns
constexpr int STEM_PLANE = STEM_PH * STEM_PITCH * 8;                                                 // bytes of one plane (hi or lo)
// EXACT: the two planes hold fp32 pairs instead -- plane 0 channels (c0, c1), plane 1 (c2, c3) of every patch pixel, 8 bytes each --
// so a lane's two neighbouring pixels are again one aligned 16-byte read per plane (pack_stem_weight_kernel orders the weights to match).
template <int COT, bool EXACT = false>
__global__ void __launch_bounds__(WAVES * 64) conv7x7_s2_stem_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const phi = smem;                               // [21][72] x 8 bytes: hi halves of (c0, c1, c2, 0)
    unsigned char* const plo = smem + STEM_PLANE;
    unsigned char* const wl = smem + 2 * STEM_PLANE;               // [14][COT][2048]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, px = lane & 31, half = lane >> 5;
    const int tiles_x = (a.Wo + TW - 1) / TW;
    const int ty0 = ((int)blockIdx.x / tiles_x) * STEM_TH, tx0 = ((int)blockIdx.x % tiles_x) * TW;
    const int n = blockIdx.y, ct0 = blockIdx.z * COT;
    // Both staging loops are unrolled with every load issued before the first use: left as loops, each of their 14 + 6 passes waited
    // out a memory round trip (the kernel ran 51 us against 5 us of matrix work).
    // weights: the chunks of this workgroup's output tiles, 16 bytes per thread and pass
    constexpr int WITEMS = STEM_CHUNKS * COT * (STEP_BYTES / 16), WPASS = (WITEMS + WAVES * 64 - 1) / (WAVES * 64);
    u32x4 wv[WPASS];
#pragma unroll
    for (int s = 0; s < WPASS; ++s) {
        const int i = min(s * (WAVES * 64) + (int)threadIdx.x, WITEMS - 1);
        const int within = i & 127, c = (i >> 7) % COT, ch = (i >> 7) / COT;
        wv[s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.packed) + ((size_t)ch * a.CT + ct0 + c) * STEP_BYTES + within * 16);
    }
    // patch: one pixel per thread and pass; the columns past the patch are zero (they meet zero weights, but must be finite)
    constexpr int PPASS = (STEM_PH * STEM_PITCH + WAVES * 64 - 1) / (WAVES * 64);
    float pv[PPASS][4];
#pragma unroll
    for (int s = 0; s < PPASS; ++s) {
        const int i = min(s * (WAVES * 64) + (int)threadIdx.x, STEM_PH * STEM_PITCH - 1);
        const int row = i / STEM_PITCH, col = min(i % STEM_PITCH, STEM_PCOLS - 1);
        const int iy = reflect(2 * ty0 + row - 3, a.H), ix = reflect(2 * tx0 + col - 3, a.W);
        // tiles may hang far over the image, where the reflection itself leaves it: clamp (those outputs are never written)
        const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
        const float* q = a.x + (((size_t)n * a.H + cy) * a.W + cx) * a.Cin;
        // (a channel the input does not have repeats the last one: it meets zero weights, and a conditional load would put a
        // branch and a full wait behind every one of the 18 loads)
        pv[s][0] = q[0]; pv[s][1] = q[min(1, a.Cin - 1)]; pv[s][2] = q[min(2, a.Cin - 1)]; pv[s][3] = q[min(3, a.Cin - 1)];
    }
#pragma unroll
    for (int s = 0; s < WPASS; ++s) {
        const int i = min(s * (WAVES * 64) + (int)threadIdx.x, WITEMS - 1);
        *reinterpret_cast<u32x4*>(wl + i * 16) = wv[s];
    }
#pragma unroll
    for (int s = 0; s < PPASS; ++s) {
        const int i = min(s * (WAVES * 64) + (int)threadIdx.x, STEM_PH * STEM_PITCH - 1);
        const bool real = i % STEM_PITCH < STEM_PCOLS;
        if constexpr (EXACT) {
            float* dh = reinterpret_cast<float*>(phi + i * 8);
            float* dl = reinterpret_cast<float*>(plo + i * 8);
            dh[0] = real ? pv[s][0] : 0.f; dh[1] = real ? pv[s][1] : 0.f; dl[0] = real ? pv[s][2] : 0.f; dl[1] = real ? pv[s][3] : 0.f;
        } else {
            const float v0 = real ? X_SCALE * pv[s][0] : 0.f, v1 = real ? X_SCALE * pv[s][1] : 0.f, v2 = real ? X_SCALE * pv[s][2] : 0.f,
                        v3 = real ? X_SCALE * pv[s][3] : 0.f;
            const unsigned h0 = pk_hi(v0, v1), h1 = pk_hi(v2, v3);
            unsigned* dh = reinterpret_cast<unsigned*>(phi + i * 8);
            unsigned* dl = reinterpret_cast<unsigned*>(plo + i * 8);
            dh[0] = h0; dh[1] = h1; dl[0] = lo_pair(h0, v0, v1); dl[1] = lo_pair(h1, v2, v3);
        }
    }
    __syncthreads();

    f32x16 acc[PT][COT];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;
    // lane (px, half) of chunk (ky, p): patch pixels (2 oy + ky, 2 ox + 4 p + 2 half) and its right neighbour
    Frag b[2][PT], w[2][COT];
    auto read_chunk = [&](int ch, int slot) {
        const int ky = ch >> 1, pq = ch & 1;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            const unsigned char* q = wl + (ch * COT + c) * STEP_BYTES + lane * 16;
            w[slot][c].hi = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(q));
            w[slot][c].lo = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(q + 1024));
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int off = ((2 * (PT * wave + t) + ky) * STEM_PITCH + 2 * px + 4 * pq + 2 * half) * 8;
            b[slot][t].hi = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(phi + off));
            b[slot][t].lo = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(plo + off));
        }
    };
    read_chunk(0, 0);
#pragma unroll
    for (int ch = 0; ch < STEM_CHUNKS; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < STEM_CHUNKS) read_chunk(ch + 1, cur ^ 1);
        if constexpr (EXACT) {          // (.hi / .lo here = plane 0 / plane 1: the same 16 bytes per lane, read as floats)
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    FragX xb;
                    xb.q0 = __builtin_bit_cast(f32x4, b[cur][t].hi); xb.q1 = __builtin_bit_cast(f32x4, b[cur][t].lo);
                    mfma_exact(acc[t][c], __builtin_bit_cast(f32x4, w[cur][c].hi), __builtin_bit_cast(f32x4, w[cur][c].lo), xb);
                }
        } else {
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][c].lo, b[cur][t].hi, acc[t][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][c].hi, b[cur][t].lo, acc[t][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][c].hi, b[cur][t].hi, acc[t][c], 0, 0, 0);
        }
    }
    __syncthreads();                                             // the epilogue's reductions reuse the LDS
    const int ox = tx0 + px;
    bool valid[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) valid[t] = (ty0 + PT * wave + t) < a.Ho && ox < a.Wo;
    if constexpr (!EXACT) {
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int c = 0; c < COT; ++c) acc[t][c] *= ACC_UNSCALE;
    }
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * (ct0 + c) + ft(r, half);
                const float bv = co < a.Cout ? a.bias[co] : 0.f;
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[t][c][r] += bv;
            }
    }
    if (a.stats) tile_stats<COT, PT>(acc, valid, reinterpret_cast<float*>(smem), a, n, (int)blockIdx.x, (int)gridDim.x, ct0);
    if (a.out_tab) finalize_if_last<COT>(a, n, ct0, (int)gridDim.x, reinterpret_cast<double*>(smem + 8192));
    else if (a.flag) flag_nonfinite<COT, PT>(acc, a);
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (!valid[t]) continue;
        float* yp = a.y + (((size_t)n * a.Ho + ty0 + PT * wave + t) * a.Wo + ox) * a.Cout;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            const int co0 = 32 * (ct0 + c);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + 8 * g + 4 * half;
                if (co + 4 > a.Cout) continue;
                const f32x4 v = {acc[t][c][4 * g + 0], acc[t][c][4 * g + 1], acc[t][c][4 * g + 2], acc[t][c][4 * g + 3]};
                *reinterpret_cast<f32x4*>(yp + co) = v;
            }
        }
    }
}

// ---- InstanceNorm + residual + activation on NHWC -------------------------------------------------------------------------
// three launches: (1) per (image, 256-pixel chunk, channel) sum and sum of squares in double; (2) per (image, channel) the
// chunks added in a fixed order (deterministic) -> scale = gamma * rstd, shift = beta - mean * scale; (3) elementwise
// y = act(x * scale + shift [+ residual]).
constexpr int PCH = 256;

__global__ void __launch_bounds__(256) nhwc_stats_kernel(const float* __restrict__ x, const long hw, const int C, const int nchunks,
                                                         double* __restrict__ partial) {
    __shared__ double red[2][256];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int c4 = C >> 2;                                  // float4 groups per pixel (C <= 1024, a multiple of 4)
    const int lanes_per_px = c4 < 256 ? c4 : 256;
    const int px_par = 256 / lanes_per_px;                  // pixels handled in parallel
    const int g = threadIdx.x % lanes_per_px, pr = threadIdx.x / lanes_per_px;
    const long p0 = (long)chunk * PCH, p1 = p0 + PCH < hw ? p0 + PCH : hw;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (pr < px_par)
        for (long p = p0 + pr; p < p1; p += px_par) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)n * hw + p) * C + 4 * g);
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += (double)v[k]; q[k] += (double)v[k] * (double)v[k]; }
        }
    for (int k = 0; k < 4; ++k) {                           // over the px_par pixel lanes of a channel group, fixed order
        __syncthreads();
        red[0][threadIdx.x] = s[k];
        red[1][threadIdx.x] = q[k];
        __syncthreads();
        if (pr == 0 && g < lanes_per_px) {
            double ts = 0, tq = 0;
            for (int r = 0; r < px_par; ++r) { ts += red[0][r * lanes_per_px + g]; tq += red[1][r * lanes_per_px + g]; }
            double* o = partial + (((size_t)n * nchunks + chunk) * C + 4 * g + k) * 2;
            o[0] = ts;
            o[1] = tq;
        }
    }
}

// one workgroup per (image, 32 channels): 8 chunk lanes per channel add every 8th chunk in double, then the 8 partial sums in a
// fixed order; T = double (nhwc_stats_kernel's partials)
template <class T>
__global__ void __launch_bounds__(256) nhwc_norm_finalize_kernel(const T* __restrict__ partial, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const long hw, const int C,
                                                                 const int nchunks, const float eps,
                                                                 float* __restrict__ scale_shift /* [N][3][C]: mean, scale, beta */) {
    __shared__ double red[2][256];
    const int n = blockIdx.y, cl = threadIdx.x & 31, kl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double ts = 0, tq = 0;
    if (c < C)
        for (int k = kl; k < nchunks; k += 8) {
            const T* o = partial + (((size_t)n * nchunks + k) * C + c) * 2;
            ts += (double)o[0];
            tq += (double)o[1];
        }
    red[0][threadIdx.x] = ts;
    red[1][threadIdx.x] = tq;
    __syncthreads();
    if (kl != 0 || c >= C) return;
    ts = 0; tq = 0;
    for (int k = 0; k < 8; ++k) { ts += red[0][k * 32 + cl]; tq += red[1][k * 32 + cl]; }
    const double mean = ts / (double)hw;
    double var = tq / (double)hw - mean * mean;             // biased variance, as InstanceNorm2d normalises with
    if (var < 0) var = 0;
    // y = (x - mean) * scale + beta, the reference's order (F.instance_norm subtracts first): folding the mean into a shift
    // (x * scale + (beta - mean * scale)) cancels two large terms when |mean| >> sigma and loses |mean| / sigma ulps
    const float g = gamma[c] / sqrtf((float)var + eps);
    scale_shift[((size_t)n * 3 + 0) * C + c] = (float)mean;
    scale_shift[((size_t)n * 3 + 1) * C + c] = g;
    scale_shift[((size_t)n * 3 + 2) * C + c] = beta[c];
}

__global__ void __launch_bounds__(256) nhwc_norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale_shift,
                                                              const float* __restrict__ residual, const float* __restrict__ res_tab,
                                                              const long hw, const int C, const int act, float* __restrict__ out) {
    const int n = blockIdx.y, c4 = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;           // float4 element of image n
    if (i >= hw * c4) return;
    const int g4 = (int)(i % c4);
    const size_t off = (size_t)n * hw * C + (size_t)i * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(scale_shift + ((size_t)n * 3 + 0) * C + 4 * g4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale_shift + ((size_t)n * 3 + 1) * C + 4 * g4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(scale_shift + ((size_t)n * 3 + 2) * C + 4 * g4);
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (residual) r = *reinterpret_cast<const f32x4*>(residual + off);
    if (res_tab) {                                            // the shortcut's own InstanceNorm (no activation), applied on the fly
        const f32x4 rm = *reinterpret_cast<const f32x4*>(res_tab + ((size_t)n * 3 + 0) * C + 4 * g4);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(res_tab + ((size_t)n * 3 + 1) * C + 4 * g4);
        const f32x4 rb = *reinterpret_cast<const f32x4*>(res_tab + ((size_t)n * 3 + 2) * C + 4 * g4);
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = fmaf(r[k] - rm[k], rs[k], rb[k]);
    }
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t = fmaf(v[k] - mu[k], sc[k], sh[k]) + r[k];
        if (act == 1) t = fmaxf(t, 0.f);
        else if (act == 2) t = t > 0.f ? t : expm1f(t);
        y[k] = t;
    }
    *reinterpret_cast<f32x4*>(out + off) = y;
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) on [N][H][W][C]: src = dst * (in - 1) / (out - 1)
__global__ void upsample2x_nhwc_kernel(const float* __restrict__ x, const int N, const int H, const int W, const int C,
                                       float* __restrict__ out) {
    const int OH = 2 * H, OW = 2 * W, c4 = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * OH * OW * c4) return;
    const int g = (int)(i % c4);
    const long p = i / c4;
    const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), n = (int)(p / ((long)OW * OH));
    const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ty = fy - (float)y0, tx = fx - (float)x0;
    auto at = [&](int yy, int xx) { return *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + yy) * W + xx) * C + 4 * g); };
    const f32x4 a = at(y0, x0), b = at(y0, x1), c = at(y1, x0), d = at(y1, x1);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float top = a[k] * (1.f - tx) + b[k] * tx, bot = c[k] * (1.f - tx) + d[k] * tx;
        o[k] = top * (1.f - ty) + bot * ty;
    }
    *reinterpret_cast<f32x4*>(out + (size_t)p * C + 4 * g) = o;
}

// ---- the exact form: fp32 operands on v_mfma_f32_32x32x2_f32 -----------------------------------------------------------------
// What the encoder falls back to when the split-f16 form raised its range flag (an activation of 4 095 or more in magnitude, a
// weight of 16 or more): the same implicit GEMM with the operands as they are, every dot product an fp32 FMA chain over
// (tap, input channel) -- the reference's own arithmetic up to the order of the sum, no range to respect.  Reads the PyTorch
// weight [Cout][Cin][KS][KS] directly (nothing is packed) and any KS / stride / channel count.  A wave owns 32 output pixels x
// 32 output channels and issues one MFMA per two input channels with two scalar loads per lane in front of it: ~20x the time of
// the split form, which does not matter on a path that exists so that no checkpoint is refused.
struct ExactArgs {
    const float* x; const float* w; const float* bias; float* y;
    int H, W, Cin, Ho, Wo, Cout, KS, stride;
};

__global__ void __launch_bounds__(WAVES * 64) conv2d_exact_kernel(const ExactArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, px = lane & 31, half = lane >> 5;
    const int n = blockIdx.y, ct = blockIdx.z;
    const int howo = a.Ho * a.Wo, pad = a.KS / 2, kk = a.KS * a.KS;
    const int p = ((int)blockIdx.x * WAVES + wave) * 32 + px;
    const bool valid = p < howo;
    const int pc = valid ? p : howo - 1;
    const int oy = pc / a.Wo, ox = pc % a.Wo;
    const int co_a = 32 * ct + px;                                  // the A operand's row of this lane: lane (row, k = half)
    const bool row_ok = co_a < a.Cout;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ky = 0; ky < a.KS; ++ky)
        for (int kx = 0; kx < a.KS; ++kx) {
            const int iy = reflect(oy * a.stride + ky - pad, a.H), ix = reflect(ox * a.stride + kx - pad, a.W);
            const float* xp = a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.Cin;
            const float* wp = a.w + (size_t)(row_ok ? co_a : 0) * a.Cin * kk + ky * a.KS + kx;
            for (int ci = 0; ci < a.Cin; ci += 2) {
                const int k = ci + half;
                const bool k_ok = k < a.Cin;
                const float bv = k_ok ? xp[k] : 0.f;
                const float av = (k_ok && row_ok) ? wp[(size_t)k * kk] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
    if (!valid) return;
    float* yp = a.y + ((size_t)n * howo + p) * a.Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * ct + ft(r, half);
        if (co < a.Cout) yp[co] = acc[r] + (a.bias ? a.bias[co] : 0.f);
    }
}

hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
int status() { return hipGetLastError() == hipSuccess ? GPNERF_OK : GPNERF_E_LAUNCH; }

template <int KS, int STRIDE, bool NARROW>
int launch_conv(const ConvArgs& a, int N, void* stream) {
    const int tiles = (a.Ho * a.Wo + WAVES * PT * 32 - 1) / (WAVES * PT * 32);
    // output tiles per workgroup: as many as still leave the chip a full round of workgroups (weights read once per COT tiles)
    int cot = 4;
    while (cot > 1 && (a.CT % cot != 0 || (long)tiles * N * (a.CT / cot) < 256)) cot >>= 1;
    const dim3 grid((unsigned)tiles, (unsigned)N, (unsigned)(a.CT / cot));
    ConvArgs g = a;
    g.tile_h = 0; g.tile_w = WAVES * PT * 32;
    const ConvArgs& a2 = g;
    if (a.exact) {
        if (cot == 4) hipLaunchKernelGGL((conv2d_nhwc_kernel<KS, STRIDE, 4, NARROW, true>), grid, dim3(WAVES * 64), 0, S_(stream), a2);
        else if (cot == 2) hipLaunchKernelGGL((conv2d_nhwc_kernel<KS, STRIDE, 2, NARROW, true>), grid, dim3(WAVES * 64), 0, S_(stream), a2);
        else hipLaunchKernelGGL((conv2d_nhwc_kernel<KS, STRIDE, 1, NARROW, true>), grid, dim3(WAVES * 64), 0, S_(stream), a2);
        return status();
    }
    if (cot == 4) hipLaunchKernelGGL((conv2d_nhwc_kernel<KS, STRIDE, 4, NARROW>), grid, dim3(WAVES * 64), 0, S_(stream), a2);
    else if (cot == 2) hipLaunchKernelGGL((conv2d_nhwc_kernel<KS, STRIDE, 2, NARROW>), grid, dim3(WAVES * 64), 0, S_(stream), a2);
    else hipLaunchKernelGGL((conv2d_nhwc_kernel<KS, STRIDE, 1, NARROW>), grid, dim3(WAVES * 64), 0, S_(stream), a2);
    return status();
}

// rows per wave of the 3x3 kernel for an (Ho x Wo) output: images of at most sixteen 8-row tiles (64 x 64 and smaller) take 4-row tiles (threshold swept: 8: 1.165, 16: 1.147, 64: 1.203 ms per frame).
// A function of the output size ALONE, so that gpnerf_conv_out_tiles() can tell the caller how many tile rows the statistics have.
int conv3x3_rows(int ho, int wo) {
    static int f_rows = -1, f_max = -1;
    if (f_rows < 0) {                      // experiment knobs (gpnerf_diag.h: the product takes the defaults)
        f_rows = dbg_int("GPNERF_CONV_ROWS", 0, 0, 2);
        f_max = dbg_int("GPNERF_CONV_ROWS_MAXTILES", 16, 0, 1 << 20);
    }
    if (f_rows) return f_rows;
    return ((ho + 7) / 8) * ((wo + TW - 1) / TW) <= f_max && ho > 4 ? 1 : 2;
}

template <int COT, int RW, int STRIDE = 1, int KSPLIT = 1, bool CAT = false, bool EXACT = false>
int launch_conv3x3_as(const ConvArgs& a, int N, int tiles, void* stream) {
    if constexpr (!EXACT) {
        if (a.exact) return launch_conv3x3_as<COT, RW, STRIDE, KSPLIT, CAT, true>(a, N, tiles, stream);
    }
    // two patch buffers per K half (+ the input norm's table); never less than what the epilogue's reductions use (tile sums,
    // finalize: 16 KB; the halves' exchange: RW * COT * 16 KB behind them)
    size_t lds = KSPLIT * 2 * (size_t)patch_bytes(RW, STRIDE) + (a.in_tab ? 3 * (size_t)a.Cin * sizeof(float) : 0);
    const size_t floor_ = 16384 + (KSPLIT == 2 ? (size_t)RW * COT * 16384 : 0);
    if (lds < floor_) lds = floor_;
    const void* fn = reinterpret_cast<const void*>(&conv3x3_s1_nhwc_kernel<COT, RW, STRIDE, KSPLIT, CAT, EXACT>);
    // > 64 KB of dynamic LDS is an opt-in per device; setting it is cheap, so it is simply set before every launch
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GPNERF_E_DEVICE;
    const dim3 grid((unsigned)tiles, (unsigned)N, (unsigned)(a.CT / COT));
    ConvArgs g = a;
    g.tile_h = tile_rows(RW); g.tile_w = TW;
    hipLaunchKernelGGL((conv3x3_s1_nhwc_kernel<COT, RW, STRIDE, KSPLIT, CAT, EXACT>), grid, dim3(WAVES * KSPLIT * 64), lds, S_(stream), g);
    return status();
}

int launch_conv3x3(const ConvArgs& a, int N, void* stream) {
    const int rw = conv3x3_rows(a.Ho, a.Wo), th = tile_rows(rw);
    const int tiles = ((a.Ho + th - 1) / th) * ((a.Wo + TW - 1) / TW);
    static int f_cot = -1;
    if (f_cot < 0) f_cot = dbg_int("GPNERF_CONV_COT", 0, 0, 2);      // experiment knob (gpnerf_diag.h)
    // one 32-channel output tile per workgroup everywhere: with the weights in registers and 54 KB of LDS two workgroups share a CU,
    // and one's staging runs under the other's MFMAs (two tiles per workgroup, GPNERF_CONV_COT=2: 1.19 -> 1.38 ms per frame)
    const int cot = (f_cot == 2 && a.CT % 2 == 0) ? 2 : 1;
    if (a.x2) return rw == 1 ? launch_conv3x3_as<1, 1, 1, 1, true>(a, N, tiles, stream) : launch_conv3x3_as<1, 2, 1, 1, true>(a, N, tiles, stream);
    // a grid of at most one workgroup per CU (the 32 x 32 stage: 192): the channel blocks are split over the two halves of an
    // eight-wave workgroup (encoder 1.097 -> 1.030 ms; at <= 400 workgroups, which takes in the 64 x 64 stage: 1.087)
    static int f_ksplit = -1;
    if (f_ksplit < 0) f_ksplit = dbg_int("GPNERF_CONV_KSPLIT_MAXWG", 256, 0, 1 << 20);     // experiment knob (gpnerf_diag.h)
    if (cot == 1 && rw == 1 && a.CB % 2 == 0 && a.CB >= 4 && (long)tiles * N * a.CT <= f_ksplit)
        return launch_conv3x3_as<1, 1, 1, 2>(a, N, tiles, stream);
    if (rw == 1) return cot == 2 ? launch_conv3x3_as<2, 1>(a, N, tiles, stream) : launch_conv3x3_as<1, 1>(a, N, tiles, stream);
    return cot == 2 ? launch_conv3x3_as<2, 2>(a, N, tiles, stream) : launch_conv3x3_as<1, 2>(a, N, tiles, stream);
}

// 3x3 stride 2 (the entry of every residual stage): the same kernel on 4-row output tiles, whose input patch is 9 x 65 pixels
int launch_conv3x3_s2(const ConvArgs& a, int N, void* stream) {
    const int tiles = ((a.Ho + 3) / 4) * ((a.Wo + TW - 1) / TW);
    return launch_conv3x3_as<1, 1, 2>(a, N, tiles, stream);
}

template <int COT, bool EXACT = false>
int launch_stem_as(const ConvArgs& a, int N, void* stream) {
    if constexpr (!EXACT) {
        if (a.exact) return launch_stem_as<COT, true>(a, N, stream);
    }
    const int tiles = ((a.Ho + STEM_TH - 1) / STEM_TH) * ((a.Wo + TW - 1) / TW);
    const size_t lds = 2 * (size_t)STEM_PLANE + (size_t)STEM_CHUNKS * COT * STEP_BYTES;
    const void* fn = reinterpret_cast<const void*>(&conv7x7_s2_stem_kernel<COT, EXACT>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GPNERF_E_DEVICE;
    const dim3 grid((unsigned)tiles, (unsigned)N, (unsigned)(a.CT / COT));
    ConvArgs g = a;
    g.tile_h = STEM_TH; g.tile_w = TW;
    hipLaunchKernelGGL((conv7x7_s2_stem_kernel<COT, EXACT>), grid, dim3(WAVES * 64), lds, S_(stream), g);
    return status();
}
int launch_stem(const ConvArgs& a, int N, void* stream) { return a.CT % 2 == 0 ? launch_stem_as<2>(a, N, stream) : launch_stem_as<1>(a, N, stream); }

}  // namespace

extern "C" {

int64_t gpnerf_conv_packed_bytes(int32_t cout, int32_t cin, int32_t ks) {
    if (cout < 1 || cin < 1 || ks < 1) return 0;
    if (ks == 7 && cin <= 4) return (int64_t)STEM_CHUNKS * ((cout + 31) / 32) * STEP_BYTES;          // the stem's K order
    const int64_t chunks = cin < 8 ? (ks * ks * cin + 15) / 16 : (int64_t)ks * ks * ((cin + 15) / 16);     // narrow inputs: flat K
    return chunks * ((cout + 31) / 32) * STEP_BYTES;
}

int gpnerf_conv_pack_weight(const float* weight, int32_t cout, int32_t cin, int32_t ks, int32_t exact, void* packed, void* stream) {
    if (!weight || !packed || cout < 1 || cin < 1 || (ks != 1 && ks != 3 && ks != 7) || exact < 0 || exact > 1) return GPNERF_E_ARG;
    if (ks == 7 && cin <= 4) {
        const int CT = (cout + 31) / 32;
        const long total = (long)STEM_CHUNKS * CT * 512;
        hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S_(stream), weight, (int)cout, (int)cin, CT,
                           (int)exact, reinterpret_cast<uint16_t*>(packed));
        return status();
    }
    const int flat = cin < 8;
    const int CB = flat ? (ks * ks * cin + 15) / 16 : (cin + 15) / 16, CT = (cout + 31) / 32;
    const long total = (flat ? (long)CB : (long)ks * ks * CB) * CT * 512;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S_(stream), weight, (int)cout,
                       (int)cin, (int)ks, CB, CT, flat, (int)exact, reinterpret_cast<uint16_t*>(packed));
    return status();
}

// workgroup tiles per image of a convolution's output (= rows per image of its `tile_stats`)
int32_t gpnerf_conv_out_tiles(int32_t h, int32_t w, int32_t cin, int32_t ks, int32_t stride) {
    if (h < 1 || w < 1 || (ks != 1 && ks != 3 && ks != 7) || (stride != 1 && stride != 2)) return 0;
    const int pad = ks / 2, ho = (h + 2 * pad - ks) / stride + 1, wo = (w + 2 * pad - ks) / stride + 1;
    if (ks == 3 && stride == 1 && cin >= 8) { const int th = tile_rows(conv3x3_rows(ho, wo)); return ((ho + th - 1) / th) * ((wo + TW - 1) / TW); }
    if (ks == 3 && stride == 2 && cin >= 8) return ((ho + 3) / 4) * ((wo + TW - 1) / TW);
    if (ks == 7 && stride == 2 && cin <= 4) return ((ho + STEM_TH - 1) / STEM_TH) * ((wo + TW - 1) / TW);
    return (ho * wo + WAVES * PT * 32 - 1) / (WAVES * PT * 32);
}

int gpnerf_conv2d_norm_cat_nhwc(const float* x, int32_t cin_a, const float* x_b, int32_t cin_b, int32_t n, int32_t h, int32_t w,
                                const void* packed, const float* bias, int32_t cout, float* y, float* tile_stats, const float* gamma,
                                const float* beta, float eps, float* out_table, uint32_t* counters, uint32_t* range_flag, int32_t exact,
                                void* stream) {
    if (n == 0) return GPNERF_OK;
    if (exact < 0 || exact > 1) return GPNERF_E_ARG;
    if (!x || !x_b || !packed || !y || n < 0 || h < 2 || w < 2 || cin_a < 16 || (cin_a & 15) || cin_b < 16 || (cin_b & 15) || cout < 4 || (cout & 3))
        return GPNERF_E_ARG;
    if (out_table && (!tile_stats || !gamma || !beta || !counters)) return GPNERF_E_ARG;
    ConvArgs a;
    a.x = x; a.x2 = x_b; a.CinA = cin_a; a.packed = reinterpret_cast<const uint16_t*>(packed); a.bias = bias; a.y = y; a.stats = tile_stats;
    a.H = h; a.W = w; a.Cin = cin_a + cin_b; a.Cout = cout; a.Ho = h; a.Wo = w;
    a.CB = a.Cin / 16; a.CT = (cout + 31) / 32;
    a.in_tab = nullptr; a.in_act = 0; a.out_tab = out_table; a.gamma = gamma; a.beta = beta; a.eps = eps; a.counters = counters;
    a.flag = exact ? nullptr : range_flag; a.exact = exact;
    return launch_conv3x3(a, n, stream);
}

int gpnerf_conv2d_norm_nhwc(const float* x, int32_t n, int32_t h, int32_t w, int32_t cin, const float* in_table, int32_t in_act,
                            const void* packed, const float* bias, int32_t cout, int32_t ks, int32_t stride, float* y, float* tile_stats,
                            const float* gamma, const float* beta, float eps, float* out_table, uint32_t* counters, uint32_t* range_flag,
                            int32_t exact, void* stream) {
    if (n == 0) return GPNERF_OK;
    if (exact < 0 || exact > 1) return GPNERF_E_ARG;
    if (!x || !packed || !y || n < 0 || h < 1 || w < 1 || cin < 1 || cout < 4 || (cout & 3)) return GPNERF_E_ARG;
    if ((ks != 1 && ks != 3 && ks != 7) || (stride != 1 && stride != 2)) return GPNERF_E_ARG;
    const int pad = ks / 2;
    if (h <= pad || w <= pad) return GPNERF_E_ARG;                       // reflection needs pad < size
    const bool narrow = cin < 8;
    if (!narrow && (cin & 15)) return GPNERF_E_ARG;                       // full 16-channel chunks, 32-byte aligned loads
    if (in_table && (narrow || ks == 7)) return GPNERF_E_ARG;                       // wide 3x3 (while staging) and 1x1 (while splitting) only
    if (in_table && (in_act < 0 || in_act > 1 || cin > 1024)) return GPNERF_E_ARG;
    if (out_table && (!tile_stats || !gamma || !beta || !counters)) return GPNERF_E_ARG;
    ConvArgs a;
    a.x = x; a.x2 = nullptr; a.CinA = cin; a.packed = reinterpret_cast<const uint16_t*>(packed); a.bias = bias; a.y = y; a.stats = tile_stats;
    a.H = h; a.W = w; a.Cin = cin; a.Cout = cout;
    a.Ho = (h + 2 * pad - ks) / stride + 1; a.Wo = (w + 2 * pad - ks) / stride + 1;
    a.CB = narrow ? (ks * ks * cin + 15) / 16 : (cin + 15) / 16; a.CT = (cout + 31) / 32;
    a.in_tab = in_table; a.in_act = in_act; a.out_tab = out_table; a.gamma = gamma; a.beta = beta; a.eps = eps; a.counters = counters;
    a.flag = exact ? nullptr : range_flag; a.exact = exact;
    if (narrow) {
        if (ks == 7 && stride == 2 && cin <= 4) return launch_stem(a, n, stream);
        if (ks == 7) return GPNERF_E_ARG;                                  // (7x7 on 5 .. 7 channels: not built)
        if (ks == 3 && stride == 1) return launch_conv<3, 1, true>(a, n, stream);
        return GPNERF_E_ARG;
    }
    if (ks == 3 && stride == 1) return launch_conv3x3(a, n, stream);
    if (ks == 3 && stride == 2) return launch_conv3x3_s2(a, n, stream);
    if (ks == 1 && stride == 1) return launch_conv<1, 1, false>(a, n, stream);
    if (ks == 1 && stride == 2) return launch_conv<1, 2, false>(a, n, stream);
    return GPNERF_E_ARG;
}

int gpnerf_conv2d_nhwc(const float* x, int32_t n, int32_t h, int32_t w, int32_t cin, const void* packed, const float* bias,
                       int32_t cout, int32_t ks, int32_t stride, float* y, float* tile_stats, uint32_t* range_flag, int32_t exact, void* stream) {
    return gpnerf_conv2d_norm_nhwc(x, n, h, w, cin, nullptr, 0, packed, bias, cout, ks, stride, y, tile_stats, nullptr, nullptr, 0.f, nullptr,
                                   nullptr, range_flag, exact, stream);
}

int gpnerf_conv2d_nhwc_exact(const float* x, int32_t n, int32_t h, int32_t w, int32_t cin, const float* weight, const float* bias,
                             int32_t cout, int32_t ks, int32_t stride, float* y, void* stream) {
    if (n == 0) return GPNERF_OK;
    if (!x || !weight || !y || n < 0 || h < 1 || w < 1 || cin < 1 || cout < 1 || ks < 1 || !(ks & 1) || stride < 1) return GPNERF_E_ARG;
    const int pad = ks / 2;
    if (h <= pad || w <= pad) return GPNERF_E_ARG;                       // reflection needs pad < size
    ExactArgs a;
    a.x = x; a.w = weight; a.bias = bias; a.y = y; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout; a.KS = ks; a.stride = stride;
    a.Ho = (h + 2 * pad - ks) / stride + 1; a.Wo = (w + 2 * pad - ks) / stride + 1;
    const int tiles = (a.Ho * a.Wo + WAVES * 32 - 1) / (WAVES * 32);
    hipLaunchKernelGGL(conv2d_exact_kernel, dim3((unsigned)tiles, (unsigned)n, (unsigned)((cout + 31) / 32)), dim3(WAVES * 64), 0, S_(stream), a);
    return status();
}

int64_t gpnerf_instance_norm_nhwc_scratch_bytes(int32_t n, int64_t hw, int32_t c) {
    if (n < 1 || hw < 1 || c < 1) return 0;
    // per-chunk partial sums (double), then the [N][3][C] mean / scale / beta table
    return (int64_t)n * ((hw + PCH - 1) / PCH) * c * 2 * (int64_t)sizeof(double) + (int64_t)n * 3 * c * (int64_t)sizeof(float);
}

int gpnerf_instance_norm_act_nhwc(const float* x, const float* gamma, const float* beta,
                                  const float* residual, int32_t n, int64_t hw, int32_t c, float eps, int32_t act, float* out,
                                  void* scratch, void* stream) {
    if (n == 0 || c == 0 || hw == 0) return GPNERF_OK;
    if (!x || !gamma || !beta || !out || !scratch || n < 0 || c < 4 || (c & 3) || c > 1024 || hw < 0 || act < 0 || act > 2) return GPNERF_E_ARG;
    const int nchunks = (int)((hw + PCH - 1) / PCH);
    double* const partial = reinterpret_cast<double*>(scratch);
    float* const table = reinterpret_cast<float*>(partial + (size_t)n * nchunks * c * 2);
    const long elems = (long)hw * (c >> 2);
    const dim3 fgrid((unsigned)((c + 31) / 32), (unsigned)n);
    hipLaunchKernelGGL(nhwc_stats_kernel, dim3((unsigned)nchunks, (unsigned)n), dim3(256), 0, S_(stream), x, (long)hw, (int)c, nchunks, partial);
    hipLaunchKernelGGL(nhwc_norm_finalize_kernel<double>, fgrid, dim3(256), 0, S_(stream), (const double*)partial, gamma, beta, (long)hw,
                       (int)c, nchunks, eps, table);
    hipLaunchKernelGGL(nhwc_norm_apply_kernel, dim3((unsigned)((elems + 255) / 256), (unsigned)n), dim3(256), 0, S_(stream), x,
                       (const float*)table, residual, (const float*)nullptr, (long)hw, (int)c, (int)act, out);
    return status();
}

int gpnerf_norm_apply_nhwc(const float* x, const float* table, const float* residual, const float* res_table, int32_t n, int64_t hw,
                           int32_t c, int32_t act, float* out, void* stream) {
    if (n == 0 || c == 0 || hw == 0) return GPNERF_OK;
    if (!x || !table || !out || n < 0 || c < 4 || (c & 3) || hw < 0 || act < 0 || act > 2 || (res_table && !residual)) return GPNERF_E_ARG;
    const long elems = (long)hw * (c >> 2);
    hipLaunchKernelGGL(nhwc_norm_apply_kernel, dim3((unsigned)((elems + 255) / 256), (unsigned)n), dim3(256), 0, S_(stream), x, table,
                       residual, res_table, (long)hw, (int)c, (int)act, out);
    return status();
}

int gpnerf_upsample2x_nhwc(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, float* out, void* stream) {
    if (n == 0) return GPNERF_OK;
    if (!x || !out || n < 0 || h < 1 || w < 1 || c < 4 || (c & 3)) return GPNERF_E_ARG;
    const long total = (long)n * h * w * 4 * (c >> 2);
    hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S_(stream), x, (int)n, (int)h, (int)w,
                       (int)c, out);
    return status();
}

}  // extern "C"
