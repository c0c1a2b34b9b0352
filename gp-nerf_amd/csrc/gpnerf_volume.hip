// gpnerf_volume.hip -- per-frame sparse 3-D convolution pyramid for gfx950.
//
// The reference builds its 4 dense feature levels with the external spconv v1.2.1 CUDA library
// (libs/nerfheads/networks/SparseConvNet.py:22-111: SubMConv3d / SparseConv3d + BatchNorm1d + ReLU, then .dense()).
// spconv is not in the reference tree, so its arithmetic is restated from the published algorithm (parity unpinned):
//   submanifold conv : out[o] = sum_k W[k] . in[o - 1 + k]           over ACTIVE inputs, outputs only at input sites
//   strided conv k3 s2 p1: out[o] = sum_k W[k] . in[2*o - 1 + k]     outputs at every site some active input reaches
// Active sets are tiny (6 890 SMPL vertices, a few 10^4 sites after the strided stages), so instead of a hash-table
// rulebook each level keeps a dense int32 index grid (site -> feature row, -1 = inactive; 44 MB at the finest level,
// a 10 us memset at HBM speed) and every kernel looks its 27 neighbours up directly.  The dense levels are written
// channels-last, the layout the render kernel samples.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/gpnerf_hip.h"

namespace {

#include "gpnerf_diag.h"       // the lab's hook points, empty in the product (csrc/nodiag/)

constexpr int KV = 27;

struct Dims { int d, h, w; };

__device__ __forceinline__ long cell_of(const Dims& s, int d, int h, int w) { return ((long)d * s.h + h) * s.w + w; }

// row index of every active site; duplicates (two vertices rounded into one voxel) resolve to the highest row
__global__ void index_kernel(const int32_t* __restrict__ coords, const int* __restrict__ m_ptr, const int m_cap, const Dims s,
                             int32_t* __restrict__ grid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = m_ptr ? min(*m_ptr, m_cap) : m_cap;
    if (i >= m) return;
    const int d = coords[3 * i], h = coords[3 * i + 1], w = coords[3 * i + 2];
    if (d < 0 || d >= s.d || h < 0 || h >= s.h || w < 0 || w >= s.w) return;
    atomicMax(grid + cell_of(s, d, h, w), i);
}

// Conv + BatchNorm (inference affine) + ReLU.  One 32-lane group per output site, lane = output channel.
//   STRIDED = false: submanifold, input position o - 1 + k, same sites in and out
//   STRIDED = true : input position 2*o - 1 + k in the finer grid
template <bool STRIDED>
__global__ void __launch_bounds__(256) conv_kernel(const float* __restrict__ in, const int cin, const int32_t* __restrict__ in_grid,
                                                   const Dims in_dims, const int32_t* __restrict__ out_coords,
                                                   const int* __restrict__ m_ptr, const int m_cap, const float* __restrict__ W,
                                                   const int cout, const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ out) {
    const int site = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int co = threadIdx.x & 31;
    const int m = m_ptr ? min(*m_ptr, m_cap) : m_cap;
    if (site >= m) return;
    const int od = out_coords[3 * site], oh = out_coords[3 * site + 1], ow = out_coords[3 * site + 2];
    float acc = 0.f;
    for (int k = 0; k < KV; ++k) {
        const int kd = k / 9, kh = (k / 3) % 3, kw = k % 3;
        const int d = (STRIDED ? 2 * od : od) - 1 + kd, h = (STRIDED ? 2 * oh : oh) - 1 + kh, w = (STRIDED ? 2 * ow : ow) - 1 + kw;
        if (d < 0 || d >= in_dims.d || h < 0 || h >= in_dims.h || w < 0 || w >= in_dims.w) continue;
        const int j = in_grid[cell_of(in_dims, d, h, w)];
        if (j < 0) continue;                                   // wave-uniform per 32-lane group: no divergence inside it
        const float* x = in + (size_t)j * cin;
        const float* wk = W + (size_t)k * cin * cout + co;
        if (co < cout)
            for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], wk[(size_t)ci * cout], acc);
    }
    if (co < cout) out[(size_t)site * cout + co] = fmaxf(fmaf(acc, scale[co], shift[co]), 0.f);
}


// The same convolution on the matrix cores, for cin a multiple of 8 (gpnerf_sparse_conv3_mfma).  One wavefront = 32 output
// sites, transposed like the render kernel's MLP: out[co][site] += W_k[ci][co] . x[neighbour_k(site)][ci] on
// v_mfma_f32_32x32x2_f32 with A = packed weights (gpnerf_sparse_pack_weight: [tap][cin/8][64 lanes][4], lane = co + 32*half,
// half h owning input channels h*cin/2 ...), B = the neighbour row's channels loaded straight from the feature matrix
// (lane = site + 32*half), 27 * cin/2 MFMAs per tile.  The VALU form above walks cin*27 dependent FMAs per site.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
// One workgroup of four wavefronts per tile of 32 output sites; the 27 taps are dealt out over the wavefronts (wave w takes taps
// w, w + 4, ...: 7, 7, 7, 6 of them), each accumulates its own 32 x 32 tile, and the four partial tiles meet in LDS and are added
// in wave order (deterministic).  With one wavefront per tile (round 2) a convolution was ONE chain of 27 x cin / 2 dependent fp32
// MFMAs on a single accumulator -- 432 x 64 cycles = 11.5 us of pure matrix latency for ~200 wavefronts on a 1 024-SIMD chip,
// 29 us per launch measured, 14 launches per frame.
template <bool STRIDED>
__global__ void __launch_bounds__(256) conv_mfma_kernel(const float* __restrict__ in, const int cin, const int32_t* __restrict__ in_grid,
                                                        const Dims in_dims, const int32_t* __restrict__ out_coords,
                                                        const int* __restrict__ m_ptr, const int m_cap, const float* __restrict__ Wp,
                                                        const int cout, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ out) {
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63, s = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const int m = m_ptr ? min(*m_ptr, m_cap) : m_cap;
    const int site = (int)blockIdx.x * 32 + s;
    if (site - s >= m) return;                                  // whole tile past the end (uniform over the workgroup)
    const bool valid = site < m;
    const int od = valid ? out_coords[3 * site] : 0, oh = valid ? out_coords[3 * site + 1] : 0, ow = valid ? out_coords[3 * site + 2] : 0;
    const int half_c = cin >> 1, ng = cin >> 3;
    f32x16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int TAPS = (KV + 3) / 4;                          // taps per wavefront (the last wavefront has one fewer)
    // this wavefront's neighbour rows first (independent loads, one round trip), then its taps, weights and rows prefetched
    int nbr[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
        const int k = wave + 4 * j;
        const int kd = k / 9, kh = (k / 3) % 3, kw = k % 3;
        const int d = (STRIDED ? 2 * od : od) - 1 + kd, h = (STRIDED ? 2 * oh : oh) - 1 + kh, w = (STRIDED ? 2 * ow : ow) - 1 + kw;
        const bool inb = valid && k < KV && d >= 0 && d < in_dims.d && h >= 0 && h < in_dims.h && w >= 0 && w < in_dims.w;
        nbr[j] = inb ? in_grid[cell_of(in_dims, d, h, w)] : -1;
    }
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tap = [&](int j, f32x4v (&b)[4]) {
        const int jj = nbr[j];
        const float* x = in + (size_t)(jj < 0 ? 0 : jj) * cin + half * half_c;
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = (g < ng && jj >= 0) ? *reinterpret_cast<const f32x4v*>(x + 4 * g) : zero4;
    };
    // The MFMA chain is branch-free (absent neighbours and the fourth wavefront's missing tap contribute zeros).
    auto load_w = [&](int j, f32x4v (&a)[4]) {
        const int k = min(wave + 4 * j, KV - 1);
        const float* wk = Wp + ((size_t)k * ng * 64 + lane) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = g < ng ? *reinterpret_cast<const f32x4v*>(wk + (size_t)g * 256) : zero4;
    };
    constexpr int DEPTH = 4;
    f32x4v buf[DEPTH][4], wbuf[DEPTH][4];                       // a tap's neighbour channels and its packed weights (L2-resident)
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j) { load_tap(j, buf[j]); load_w(j, wbuf[j]); }
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
        if (j + DEPTH - 1 < TAPS) { load_tap(j + DEPTH - 1, buf[(j + DEPTH - 1) % DEPTH]); load_w(j + DEPTH - 1, wbuf[(j + DEPTH - 1) % DEPTH]); }
        const f32x4v(&cur)[4] = buf[j % DEPTH];
        const f32x4v(&wa)[4] = wbuf[j % DEPTH];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < ng) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[g][0], cur[g][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[g][1], cur[g][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[g][2], cur[g][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[g][3], cur[g][3], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (!valid) return;
    // wavefront w finishes accumulator registers 4 w .. 4 w + 3: the four partial tiles added in wave order
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * wave + i;
        const float v = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;       // 32x32 accumulator: row of register r in lane half `half`
        if (co < cout) out[(size_t)site * cout + co] = fmaxf(fmaf(v, scale[co], shift[co]), 0.f);
    }
}


// ---- the same convolution in the encoder's split-precision arithmetic (cin = 16 or 32) --------------------------------------
// v_mfma_f32_32x32x2_f32 runs on the vector ALUs at 1/16 of the f16 rate: a 32-channel tap above is 16 of them, 1 024 cycles,
// and a level of ~5 x 10^4 sites 30-39 us.  Here every fp32 operand is f16 hi + lo (gpnerf_conv.hip: hi rounded to nearest, lo the
// rest, operands scaled by exact powers of two so that lo stays a normal number; three v_mfma_f32_32x32x16_f16 per 16 channels,
// lo x lo dropped, f32 accumulation): 6 matrix instructions of 32 cycles per 32-channel tap, the conversion of the gathered rows
// (16 values per lane and tap) in their shadow.  Weights: gpnerf_sparse_pack_weight16 -- per (tap, 16-channel chunk) 2 KB of f16
// [hi | lo][lane = co + 32 half][8: channels 16 chunk + 8 half ..] followed by 2 KB of the SAME values in fp32 (scaled alike).
// The f16 range (|16 x| < 65 504) is not given by construction here (BatchNorm + ReLU outputs of trained weights, the code the
// caller hands in): a wavefront that meets a larger value in a tap runs THAT tap on the fp32 instructions with the fp32 copy
// of the weights -- same scale, same accumulator, no host round trip, results within the split's 2^-22 of each other.
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
constexpr float W16_SCALE = 4096.f, X16_SCALE = 16.f, ACC16_UNSCALE = 1.f / (4096.f * 16.f);
constexpr int STEP16_BYTES = 4096;                             // one (tap, chunk): 2 KB f16 hi | lo + 2 KB fp32
__device__ __forceinline__ unsigned pk_hi16(float a, float b) { const h2v r = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, r); }
__device__ __forceinline__ unsigned lo_pair16(unsigned w, float x0, float x1) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w), "v"(x1));
    return r;
}

template <bool STRIDED, int NC>                                // NC = cin / 16
__global__ void __launch_bounds__(256) conv_mfma16_kernel(const float* __restrict__ in, const int32_t* __restrict__ in_grid,
                                                          const Dims in_dims, const int32_t* __restrict__ out_coords,
                                                          const int* __restrict__ m_ptr, const int m_cap,
                                                          const unsigned char* __restrict__ Wp, const int cout,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ out, float* __restrict__ vol) {
    // vol (or nullptr): the level's zeroed dense volume [D][H][W][cout] of the OUTPUT sites' grid -- the rows are also written at
    // their sites' cells (SparseConvTensor.dense() for a level without duplicate sites: every level but the vertices')
    __shared__ float red[4][16][64];
    constexpr int cin = 16 * NC;
    const int lane = threadIdx.x & 63, s = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const int m = m_ptr ? min(*m_ptr, m_cap) : m_cap;
    const int site = (int)blockIdx.x * 32 + s;
    if (site - s >= m) return;                                  // whole tile past the end (uniform over the workgroup)
    const bool valid = site < m;
    // Every load below is UNCONDITIONAL (clamped index, the value dropped afterwards where it does not apply): a load under a lane
    // condition becomes a branch around it with a full wait behind it, and the taps' gathers -- meant to be DEPTH taps ahead --
    // each waited out their own round trip (a level of ~5 x 10^4 sites: 20 -> see DESIGN.md 4.2)
    const int site_c = min(site, m - 1);                       // (m >= 1 here: the tile holds a site)
    const int od = out_coords[3 * site_c], oh = out_coords[3 * site_c + 1], ow = out_coords[3 * site_c + 2];
    f32x16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int TAPS = (KV + 3) / 4;                          // taps per wavefront (the last wavefront has one fewer)
    int nbr[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
        const int k = wave + 4 * j;
        const int kd = k / 9, kh = (k / 3) % 3, kw = k % 3;
        const int d = (STRIDED ? 2 * od : od) - 1 + kd, h = (STRIDED ? 2 * oh : oh) - 1 + kh, w = (STRIDED ? 2 * ow : ow) - 1 + kw;
        const bool inb = valid && k < KV && d >= 0 && d < in_dims.d && h >= 0 && h < in_dims.h && w >= 0 && w < in_dims.w;
        const int g = in_grid[cell_of(in_dims, min(max(d, 0), in_dims.d - 1), min(max(h, 0), in_dims.h - 1), min(max(w, 0), in_dims.w - 1))];
        nbr[j] = inb ? g : -1;
    }
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    // a tap's neighbour row: this lane's 8 channels of every 16-channel chunk (absent neighbours and the fourth wavefront's missing
    // tap contribute zeros: the MFMA chain is branch-free)
    auto load_tap = [&](int j, f32x4v (&b)[NC][2]) {
        const int jj = nbr[j];
        const float* x = in + (size_t)(jj < 0 ? 0 : jj) * cin + 8 * half;      // (row 0 exists: m >= 1; its values are dropped below)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            b[c][0] = *reinterpret_cast<const f32x4v*>(x + 16 * c);
            b[c][1] = *reinterpret_cast<const f32x4v*>(x + 16 * c + 4);
        }
    };
    auto load_w = [&](int j, u32x4v (&a)[NC][2]) {
        const int k = min(wave + 4 * j, KV - 1);
        const unsigned char* wk = Wp + (size_t)k * NC * STEP16_BYTES + lane * 16;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            a[c][0] = *reinterpret_cast<const u32x4v*>(wk + c * STEP16_BYTES);
            a[c][1] = *reinterpret_cast<const u32x4v*>(wk + c * STEP16_BYTES + 1024);
        }
    };
#ifndef GPNERF_SPARSE16_DEPTH
#define GPNERF_SPARSE16_DEPTH 2   // 100 registers, 4 waves per SIMD: frame phase 0.300 -> 0.279 ms (3: 130 registers, 4: 162; tools/probes/sparse16_depth.sh)
#endif
    constexpr int DEPTH = GPNERF_SPARSE16_DEPTH;
    f32x4v buf[DEPTH][NC][2];
    u32x4v wbuf[DEPTH][NC][2];
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j) { load_tap(j, buf[j]); load_w(j, wbuf[j]); }
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
        if (j + DEPTH - 1 < TAPS) { load_tap(j + DEPTH - 1, buf[(j + DEPTH - 1) % DEPTH]); load_w(j + DEPTH - 1, wbuf[(j + DEPTH - 1) % DEPTH]); }
        f32x4v xs[NC][2];
        float big = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                xs[c][q] = nbr[j] >= 0 ? buf[j % DEPTH][c][q] * X16_SCALE : zero4;
#pragma unroll
                for (int i = 0; i < 4; ++i) big = fmaxf(big, fabsf(xs[c][q][i]));
            }
        // (a NaN compares false and takes the f16 path, where it stays a NaN)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(big >= 65504.f) != 0, 0)) {
            const int k = min(wave + 4 * j, KV - 1);
            const float* w32 = reinterpret_cast<const float*>(Wp + (size_t)k * NC * STEP16_BYTES + 2048) + lane * 8;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4v a0 = *reinterpret_cast<const f32x4v*>(w32 + c * (STEP16_BYTES / 4)),
                             a1 = *reinterpret_cast<const f32x4v*>(w32 + c * (STEP16_BYTES / 4) + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], xs[c][0][i], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], xs[c][1][i], acc, 0, 0, 0);
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            unsigned H[4], Lo[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                H[2 * q] = pk_hi16(xs[c][q][0], xs[c][q][1]); Lo[2 * q] = lo_pair16(H[2 * q], xs[c][q][0], xs[c][q][1]);
                H[2 * q + 1] = pk_hi16(xs[c][q][2], xs[c][q][3]); Lo[2 * q + 1] = lo_pair16(H[2 * q + 1], xs[c][q][2], xs[c][q][3]);
            }
            // (one wait state behind the last inline-asm conversion, carried by a statement every reader of the operand depends on:
            // gfx950 does not interlock a VALU write with an MFMA read in the next slot, and LLVM cannot see into lo_pair16 --
            // tools/micro/asm_producer_hazards.hip, gpnerf_kernels.hip settle_operand)
            asm("s_nop 0" : "+v"(Lo[0]), "+v"(Lo[1]), "+v"(Lo[2]), "+v"(Lo[3]));
            const h8v xh = __builtin_bit_cast(h8v, H), xl = __builtin_bit_cast(h8v, Lo);
            const h8v wh = __builtin_bit_cast(h8v, wbuf[j % DEPTH][c][0]), wl = __builtin_bit_cast(h8v, wbuf[j % DEPTH][c][1]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
        }
    }
    // the four channels this wavefront finishes (8 wave + 4 half ..): their BatchNorm factors are fetched before the exchange, all at
    // once and unconditionally (clamped index) -- under `co < cout` each pair was a branch with a full wait behind it
    float bsc[4], bsh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = min(i + 8 * wave + 4 * half, cout - 1);
        bsc[i] = scale[co]; bsh[i] = shift[co];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (!valid) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * wave + i;
        const float v = (((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane]) * ACC16_UNSCALE;
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < cout) {
            const float y = fmaxf(fmaf(v, bsc[i], bsh[i]), 0.f);
            out[(size_t)site * cout + co] = y;
            if (vol) {
                // (submanifold: the output grid is the input grid; strided: the coarse grid has half the input's size, rounded down)
                const Dims vd = STRIDED ? Dims{in_dims.d / 2, in_dims.h / 2, in_dims.w / 2} : in_dims;
                vol[(size_t)cell_of(vd, od, oh, ow) * cout + co] = y;
            }
        }
    }
}

// spconv's strided rulebook takes EVERY input row, so two vertices rounded into one voxel both contribute, while its
// submanifold lookups find one row per voxel.  Fold the former into the grid formulation: add the features of the other
// rows of a voxel onto the indexed row (only the vertex level can hold duplicates).
// Two launches so that the sum is ORDERED (an atomicAdd per duplicate would add a voxel's rows in arrival order, and fp32
// addition does not commute with that): count_duplicates_kernel counts, per indexed row, how many other rows share its voxel;
// merge_duplicates_kernel gives every row that has company one wavefront, which scans the row list in ascending order and
// adds the matching rows' features onto its own, lowest row index first.
// scratch: count[m], then DUP_SLOTS row slots per owner: a row that shares an owner's voxel leaves its index in the owner's
// next free slot, so the owner finds its company without scanning the row list (owners with more company than slots still scan)
constexpr int DUP_SLOTS = 8;
__global__ void count_duplicates_kernel(const int32_t* __restrict__ coords, const int32_t* __restrict__ grid, const int m,
                                        const Dims s, int32_t* __restrict__ count) {
    const int site = blockIdx.x * blockDim.x + threadIdx.x;
    if (site >= m) return;
    const int d = coords[3 * site], h = coords[3 * site + 1], w = coords[3 * site + 2];
    if (d < 0 || d >= s.d || h < 0 || h >= s.h || w < 0 || w >= s.w) return;
    const int owner = grid[cell_of(s, d, h, w)];
    if (owner != site) {
        const int slot = atomicAdd(count + owner, 1);
        if (slot < DUP_SLOTS) count[m + owner * DUP_SLOTS + slot] = site;
    }
}

__global__ void merge_duplicates_kernel(float* __restrict__ feat, const int c, const int32_t* __restrict__ coords,
                                        const int32_t* __restrict__ grid, const int m, const Dims s,
                                        const int32_t* __restrict__ count) {
    const int site = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (site >= m) return;
    const int want = count[site];
    if (want == 0) return;
    float acc = lane < c ? feat[(size_t)site * c + lane] : 0.f;
    if (want <= DUP_SLOTS) {
        // the slots were filled in arrival order: add the rows lowest index first all the same (a fixed order of additions)
        int r[DUP_SLOTS];
#pragma unroll
        for (int k = 0; k < DUP_SLOTS; ++k) r[k] = k < want ? count[m + site * DUP_SLOTS + k] : 0x7fffffff;
#pragma unroll
        for (int a = 0; a < DUP_SLOTS - 1; ++a)
#pragma unroll
            for (int b = 0; b < DUP_SLOTS - 1 - a; ++b) {
                const int lo = min(r[b], r[b + 1]), hi = max(r[b], r[b + 1]);
                r[b] = lo; r[b + 1] = hi;
            }
#pragma unroll
        for (int k = 0; k < DUP_SLOTS; ++k)
            if (k < want && lane < c) acc += feat[(size_t)r[k] * c + lane];      // rows that are added are never owners: nobody writes them
        if (lane < c) feat[(size_t)site * c + lane] = acc;
        return;
    }
    int found = 0;
    for (int base = 0; base < m && found < want; base += 64) {
        const int j = base + lane;
        bool hit = false;
        if (j < m && j != site) {
            const int d = coords[3 * j], h = coords[3 * j + 1], w = coords[3 * j + 2];
            hit = d >= 0 && d < s.d && h >= 0 && h < s.h && w >= 0 && w < s.w && grid[cell_of(s, d, h, w)] == site;
        }
        unsigned long long mask = __ballot(hit);
        found += __popcll(mask);
        while (mask) {
            const int j0 = base + __ffsll((long long)mask) - 1;
            if (lane < c) acc += feat[(size_t)j0 * c + lane];
            mask &= mask - 1;
        }
    }
    if (lane < c) feat[(size_t)site * c + lane] = acc;
}

// spconv's submanifold rulebook on rows that SHARE a voxel (oracle/producers_ref.py header item 6; recalled from spconv v1.2.1
// src/spconv/indice.cc create_submconv_indice_pair_cpu + spconv_ops.cc indiceConv): the position -> row grid holds ONE row per
// voxel (the owner: the highest row); every input row generates a pair towards the owner of each active neighbour position, and
// the centre tap is a plain `features @ W[13]` on every row.  So
//     owner j:      out[j] = f_j W_c + sum_{k != c} S(p_j - 1 + k) W_k,   S(v) = the SUM of all rows in voxel v
//     non-owner j:  out[j] = f_j W_c
// The convolution kernels compute conv(S) through the grid for every row -- exactly that wherever the row's OWN voxel is not
// shared (its neighbours' sums S are what the grid's rows hold once gpnerf_sparse_merge_duplicates has run on a copy of the
// input, and S of its own voxel is f_j).  This kernel recomputes the few rows whose own voxel IS shared
// (body-like vertices at 5 mm: ~190 of 6 890): one 32-lane group per row, lane = output channel, fp32 FMAs in tap order.
//   own: the layer's input rows as they are; merged: the copy whose owner rows hold S; count: merge_duplicates' scratch (count[j]
//   = how many other rows share owner j's voxel); W: raw [27][cin][cout]
__global__ void __launch_bounds__(256) subm_shared_rows_kernel(const float* __restrict__ own, const float* __restrict__ merged, const int cin,
                                                               const int32_t* __restrict__ grid, const Dims s, const int32_t* __restrict__ coords,
                                                               const int m, const int32_t* __restrict__ count, const float* __restrict__ W,
                                                               const int cout, const float* __restrict__ scale, const float* __restrict__ shift,
                                                               float* __restrict__ out) {
    const int j = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int co = threadIdx.x & 31;
    if (j >= m) return;
    const int d0 = coords[3 * j], h0 = coords[3 * j + 1], w0 = coords[3 * j + 2];
    if (d0 < 0 || d0 >= s.d || h0 < 0 || h0 >= s.h || w0 < 0 || w0 >= s.w) return;
    const int owner = grid[cell_of(s, d0, h0, w0)];
    if (owner == j && count[j] == 0) return;                   // the row has its voxel to itself: the convolution's value stands
    if (co >= cout) return;
    float acc = 0.f;
    {
        const float* x = own + (size_t)j * cin;
        const float* wk = W + (size_t)13 * cin * cout + co;
        for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], wk[(size_t)ci * cout], acc);
    }
    if (owner == j) {
        for (int k = 0; k < KV; ++k) {
            if (k == 13) continue;
            const int d = d0 - 1 + k / 9, h = h0 - 1 + (k / 3) % 3, w = w0 - 1 + k % 3;
            if (d < 0 || d >= s.d || h < 0 || h >= s.h || w < 0 || w >= s.w) continue;
            const int nb = grid[cell_of(s, d, h, w)];
            if (nb < 0) continue;
            const float* x = merged + (size_t)nb * cin;
            const float* wk = W + (size_t)k * cin * cout + co;
            for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], wk[(size_t)ci * cout], acc);
        }
    }
    out[(size_t)j * cout + co] = fmaxf(fmaf(acc, scale[co], shift[co]), 0.f);
}

__global__ void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, const long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// strided conv, step 1: mark every coarse site reached by an active fine site (out = (p + 1 - k) / 2 when even)
__global__ void mark_kernel(const int32_t* __restrict__ coords, const int* __restrict__ m_ptr, const int m_cap, const Dims out_dims,
                            int32_t* __restrict__ out_grid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = m_ptr ? min(*m_ptr, m_cap) : m_cap;
    if (i >= m) return;
    const int p[3] = {coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]};
    for (int k = 0; k < KV; ++k) {
        const int kk[3] = {k / 9, (k / 3) % 3, k % 3};
        int o[3];
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            const int num = p[a] + 1 - kk[a];
            ok = ok && num >= 0 && (num & 1) == 0;
            o[a] = num >> 1;
        }
        if (ok && o[0] < out_dims.d && o[1] < out_dims.h && o[2] < out_dims.w) out_grid[cell_of(out_dims, o[0], o[1], o[2])] = -2;
    }
}

// step 2: give every marked site a row (order is arbitrary; nothing downstream depends on it).  A workgroup takes
// ASSIGN_PER x 256 consecutive cells, ranks its marked cells with a wave scan + four LDS words, and claims their rows with ONE
// atomic: the marked cells are a thin shell (a wavefront holds one or two), so an atomic per site -- or per wavefront -- is
// ~10 k same-address atomics at ~10 ns each (115 us at the finest level); this is a few hundred.
constexpr int ASSIGN_PER = 16;
__global__ void __launch_bounds__(256) assign_kernel(const Dims s, int32_t* __restrict__ grid, int* __restrict__ counter, const int cap,
                                                      int32_t* __restrict__ coords) {
    __shared__ int wave_total[4];
    __shared__ int block_base;
    const long cells = (long)s.d * s.h * s.w;
    const long b0 = (long)blockIdx.x * (256 * ASSIGN_PER);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < ASSIGN_PER; ++j) {
        const long i = b0 + j * 256 + threadIdx.x;
        if (i < cells && grid[i] == -2) mask |= 1u << j;
    }
    const int cnt = __popc(mask);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        block_base = total ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    int row = block_base + incl - cnt;
    for (int w = 0; w < wave; ++w) row += wave_total[w];
    while (mask) {
        const int j = __ffs(mask) - 1;
        mask &= mask - 1;
        const long i = b0 + j * 256 + threadIdx.x;
        if (row >= cap) { grid[i] = -1; ++row; continue; }
        grid[i] = row;
        coords[3 * row + 0] = (int)(i / ((long)s.h * s.w));
        coords[3 * row + 1] = (int)((i / s.w) % s.h);
        coords[3 * row + 2] = (int)(i % s.w);
        ++row;
    }
}

// .dense() in the render kernel's channels-last layout; the volume is zero-filled by the caller side of this file
__global__ void dense_kernel(const float* __restrict__ feat, const int c, const int32_t* __restrict__ coords,
                             const int32_t* __restrict__ grid, const int* __restrict__ m_ptr, const int m_cap, const Dims s,
                             float* __restrict__ vol) {
    const int site = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int ch = threadIdx.x & 31;
    const int m = m_ptr ? min(*m_ptr, m_cap) : m_cap;
    if (site >= m || ch >= c) return;
    const long cell = cell_of(s, coords[3 * site], coords[3 * site + 1], coords[3 * site + 2]);
    if (grid[cell] != site) return;                            // duplicates: the indexed row owns the voxel
    vol[cell * c + ch] = feat[(size_t)site * c + ch];
}

hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
// Buffers are (re)set by a kernel of the library's own rather than hipMemsetAsync: a memset node captured into a HIP graph was not
// reliably ordered before the kernel node after it (gpnerf_kernels.hip, zero_words_kernel).  n_words 32-bit words of `value`.
__global__ void __launch_bounds__(256) fill_words_kernel(unsigned* __restrict__ p, const long n_words, const unsigned value) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const long n4 = n_words / 4, stride = (long)gridDim.x * blockDim.x, first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const u4 v = {value, value, value, value};
    for (long i = first; i < n4; i += stride) reinterpret_cast<u4*>(p)[i] = v;          // (hipMalloc / torch allocations: 16-byte aligned)
    for (long i = n4 * 4 + first; i < n_words; i += stride) p[i] = value;
}
bool fill_async(void* p, size_t bytes, unsigned value, void* stream) {     // bytes: a multiple of 4; p: 16-byte aligned
    const long n = (long)(bytes / 4), wgs = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)(wgs < 1 ? 1 : (wgs > 16384 ? 16384 : wgs))), dim3(256), 0, S_(stream),
                       static_cast<unsigned*>(p), n, value);
    return hipGetLastError() == hipSuccess;
}
int status() { return hipGetLastError() == hipSuccess ? GPNERF_OK : GPNERF_E_LAUNCH; }
bool bad(const int32_t* dims) { return !dims || dims[0] < 1 || dims[1] < 1 || dims[2] < 1; }


// Vertex-code attention (libs/nerfheads/trainhead.py:48-52 through networks/MultiHeadAttention.py:61-98, sum=False):
// per SMPL vertex, query = its code (length 1), keys = values = its features in the V source views.  One wavefront walks
// vertices; lane j < d_model owns output column j of every projection, its four weight rows live in registers.
//   qh = Wq q / sqrt(d_k);  kh_v = Wk f_v;  vh_v = Wv f_v;  per head: softmax_v(qh . kh_v);  out = sum_v a_v vh_v;  y = Wfc out
constexpr int ATT_MAX = 64;     // d_model, kv_dim <= 64
template <int DMAX>             // 32 when d_model, kv_dim <= 32 (the weight rows fit the register file), else 64
__global__ void __launch_bounds__(256, 1) vertex_attention_kernel(const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ wq,
                                        const float* __restrict__ wk, const float* __restrict__ wv, const float* __restrict__ wfc,
                                        const int n, const int d_model, const int kv_dim, const int n_head, const int views,
                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int j = lane < d_model ? lane : 0;
    const int d_k = d_model / n_head;
    const float inv_t = 1.f / sqrtf((float)d_k);
    float rq[DMAX], rk[DMAX], rv[DMAX], rf[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
        rq[i] = i < d_model ? wq[j * d_model + i] : 0.f;
        rf[i] = i < d_model ? wfc[j * d_model + i] : 0.f;
        rk[i] = i < kv_dim ? wk[j * kv_dim + i] : 0.f;
        rv[i] = i < kv_dim ? wv[j * kv_dim + i] : 0.f;
    }
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long v = wave; v < n; v += n_waves) {
        const float qx = lane < d_model ? q[v * d_model + lane] : 0.f;
        float qh = 0.f;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) qh = fmaf(rq[i], __shfl(qx, i), qh);
        qh *= inv_t;
        float score[4], val[4];                 // views <= 4
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            score[s] = -INFINITY; val[s] = 0.f;
            if (s < views) {
                const float fx = lane < kv_dim ? kv[(v * views + s) * kv_dim + lane] : 0.f;
                float kh = 0.f, vh = 0.f;
#pragma unroll
                for (int i = 0; i < DMAX; ++i) {
                    const float f = __shfl(fx, i);
                    kh = fmaf(rk[i], f, kh);
                    vh = fmaf(rv[i], f, vh);
                }
                float p = qh * kh;              // sum over the d_k lanes of this head (d_k is a power of two)
                for (int o = 1; o < d_k; o <<= 1) p += __shfl_xor(p, o);
                score[s] = p; val[s] = vh;
            }
        }
        float m = fmaxf(fmaxf(score[0], score[1]), fmaxf(score[2], score[3]));
        float den = 0.f, acc = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float e = s < views ? __expf(score[s] - m) : 0.f;
            den += e; acc = fmaf(e, val[s], acc);
        }
        const float o = lane < d_model ? acc / den : 0.f;
        float y = 0.f;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) y = fmaf(rf[i], __shfl(o, i), y);
        if (lane < d_model) out[v * d_model + lane] = y;
    }
}

// The same attention on the matrix cores for d_model = kv_dim = 32 (the reference's sizes): one wavefront = 32 vertices, transposed
// like everything else here -- out[channel][vertex] += W[channel][k] . x[k][vertex] on v_mfma_f32_32x32x2_f32, 16 per projection, eight
// projections per tile (q, k and v of up to four views, fc).  In the accumulator layout a lane holds 16 channels of its vertex, four
// runs of 4 (register group g = channels 8 g + 4 half ..): the per-head dot products are sums over register groups (+ one exchange
// between the lane halves for heads of 8 channels and more), the softmax over the views is per lane, and the attention output goes
// through LDS once to become the B operand of the last projection.  The shuffle form above spends ~1 us per vertex and wavefront.
template <int VIEWS>
__global__ void __launch_bounds__(64) vertex_attention_mfma_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                                   const float* __restrict__ wq, const float* __restrict__ wk,
                                                                   const float* __restrict__ wv, const float* __restrict__ wfc, const int n,
                                                                   const int n_head, float* __restrict__ out) {
    __shared__ float stage[32][33];
    const int lane = threadIdx.x, nn = lane & 31, half = lane >> 5;
    const int v = (int)blockIdx.x * 32 + nn, vc = min(v, n - 1);
    const int d_k = 32 / n_head;
    const float inv_t = 1.f / sqrtf((float)d_k);
    // operand of k-step s from a row of 32 values held as 8 x float4: element 2 s + half
    auto pick = [&](const f32x4v (&row)[8], int s) { const f32x4v t = row[s >> 1]; return half ? t[2 * (s & 1) + 1] : t[2 * (s & 1)]; };
    auto load_row = [&](const float* p, f32x4v (&row)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) row[i] = reinterpret_cast<const f32x4v*>(p)[i];
    };
    // every row the tile needs -- four weight rows, the vertex's code, its features in the views -- is requested before the first
    // MFMA (a 64-thread workgroup has the whole register file): loaded per projection, each of the eight waited out its own round trip
    f32x4v wrq[8], wrk[8], wrv[8], wrf[8], xq[8], xf[VIEWS][8];
    load_row(wq + nn * 32, wrq); load_row(wk + nn * 32, wrk); load_row(wv + nn * 32, wrv); load_row(wfc + nn * 32, wrf);
    load_row(q + (size_t)vc * 32, xq);
#pragma unroll
    for (int s = 0; s < VIEWS; ++s) load_row(kv + ((size_t)vc * VIEWS + s) * 32, xf[s]);
    // D = W . X for one 32 x 32 weight matrix (row m = this lane's output channel as A operand) and the tile's 32 input rows
    auto project = [&](const f32x4v (&wr)[8], const f32x4v (&x)[8]) {
        f32x16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pick(wr, s), pick(x, s), acc, 0, 0, 0);
        return acc;
    };
    f32x16v qh = project(wrq, xq);
#pragma unroll
    for (int r = 0; r < 16; ++r) qh[r] *= inv_t;
    float score[VIEWS][4];
    f32x16v val[VIEWS];
#pragma unroll
    for (int s = 0; s < VIEWS; ++s) {
        const f32x16v kh = project(wrk, xf[s]);
        val[s] = project(wrv, xf[s]);
        float sg[4];                                    // register group g: channels 8 g + 4 half .. + 3 of this lane's vertex
#pragma unroll
        for (int g = 0; g < 4; ++g) sg[g] = ((qh[4 * g] * kh[4 * g] + qh[4 * g + 1] * kh[4 * g + 1]) + qh[4 * g + 2] * kh[4 * g + 2]) + qh[4 * g + 3] * kh[4 * g + 3];
        if (d_k >= 8) {
#pragma unroll
            for (int g = 0; g < 4; ++g) sg[g] += __shfl_xor(sg[g], 32);
        }
        if (d_k == 16) { const float a = sg[0] + sg[1], b = sg[2] + sg[3]; sg[0] = sg[1] = a; sg[2] = sg[3] = b; }
        if (d_k == 32) { const float a = (sg[0] + sg[1]) + (sg[2] + sg[3]); sg[0] = sg[1] = sg[2] = sg[3] = a; }
#pragma unroll
        for (int g = 0; g < 4; ++g) score[s][g] = sg[g];
    }
    // softmax over the views per head, and the heads' weighted values: this lane's 16 channels of the attention output
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float m = score[0][g];
#pragma unroll
        for (int s = 1; s < VIEWS; ++s) m = fmaxf(m, score[s][g]);
        float den = 0.f, e[VIEWS];
#pragma unroll
        for (int s = 0; s < VIEWS; ++s) { e[s] = __expf(score[s][g] - m); den += e[s]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < VIEWS; ++s) acc = fmaf(e[s], val[s][4 * g + i], acc);
            stage[nn][8 * g + 4 * half + i] = acc / den;
        }
    }
    __syncthreads();
    f32x4v xo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xo[i] = f32x4v{stage[nn][4 * i], stage[nn][4 * i + 1], stage[nn][4 * i + 2], stage[nn][4 * i + 3]};
    const f32x16v y = project(wrf, xo);
    if (v < n) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4v*>(out + (size_t)v * 32 + 8 * g + 4 * half) = f32x4v{y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
    }
}
}  // namespace

extern "C" {

int gpnerf_sparse_index(const int32_t* coords, const int32_t* m_dev, int32_t m_cap, const int32_t* dims, int32_t* grid,
                        void* stream) {
    if (!coords || !grid || bad(dims) || m_cap < 0) return GPNERF_E_ARG;
    const Dims s{dims[0], dims[1], dims[2]};
    if (!fill_async(grid, sizeof(int32_t) * (size_t)s.d * s.h * s.w, 0xFFFFFFFFu, stream)) return GPNERF_E_LAUNCH;
    if (m_cap == 0) return GPNERF_OK;
    hipLaunchKernelGGL(index_kernel, dim3((m_cap + 255) / 256), dim3(256), 0, S_(stream), coords, (const int*)m_dev, (int)m_cap, s, grid);
    return status();
}

int gpnerf_sparse_conv3(int32_t strided, const float* in, int32_t cin, const int32_t* in_grid, const int32_t* in_dims,
                        const int32_t* out_coords, const int32_t* m_dev, int32_t m_cap, const float* weight, int32_t cout,
                        const float* bn_scale, const float* bn_shift, float* out, void* stream) {
    if (!in || !in_grid || bad(in_dims) || !out_coords || !weight || !bn_scale || !bn_shift || !out || cin < 1 || cout < 1 ||
        cout > 32 || m_cap < 0)
        return GPNERF_E_ARG;
    if (m_cap == 0) return GPNERF_OK;
    const Dims s{in_dims[0], in_dims[1], in_dims[2]};
    const dim3 grid((m_cap + 7) / 8), block(256);
    if (strided)
        hipLaunchKernelGGL(conv_kernel<true>, grid, block, 0, S_(stream), in, (int)cin, in_grid, s, out_coords, (const int*)m_dev,
                           (int)m_cap, weight, (int)cout, bn_scale, bn_shift, out);
    else
        hipLaunchKernelGGL(conv_kernel<false>, grid, block, 0, S_(stream), in, (int)cin, in_grid, s, out_coords, (const int*)m_dev,
                           (int)m_cap, weight, (int)cout, bn_scale, bn_shift, out);
    return status();
}

int64_t gpnerf_sparse_packed_weight_floats(int32_t cin) { return cin >= 8 && cin <= 32 && cin % 8 == 0 ? (int64_t)KV * (cin / 8) * 256 : 0; }

int gpnerf_sparse_pack_weight(const float* weight, int32_t cin, int32_t cout, float* packed) {
    if (!weight || !packed || cin < 8 || cin > 32 || cin % 8 || cout < 1 || cout > 32) return GPNERF_E_ARG;
    const int ng = cin / 8, half_c = cin / 2;
    for (int k = 0; k < KV; ++k)
        for (int g = 0; g < ng; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    const int co = lane & 31, ci = (lane >> 5) * half_c + 4 * g + i;
                    packed[(((size_t)k * ng + g) * 64 + lane) * 4 + i] = co < cout ? weight[((size_t)k * cin + ci) * cout + co] : 0.f;
                }
    return GPNERF_OK;
}


int64_t gpnerf_sparse_packed_weight16_bytes(int32_t cin) { return (cin == 16 || cin == 32) ? (int64_t)KV * (cin / 16) * STEP16_BYTES : 0; }

int gpnerf_sparse_pack_weight16(const float* weight, int32_t cin, int32_t cout, void* packed) {
    if (!weight || !packed || (cin != 16 && cin != 32) || cout < 1 || cout > 32) return GPNERF_E_ARG;
    const int nc = cin / 16;
    unsigned char* const p = static_cast<unsigned char*>(packed);
    for (int k = 0; k < KV; ++k)
        for (int c = 0; c < nc; ++c) {
            unsigned char* blk = p + ((size_t)k * nc + c) * STEP16_BYTES;
            _Float16* hi = reinterpret_cast<_Float16*>(blk);
            _Float16* lo = reinterpret_cast<_Float16*>(blk + 1024);
            float* f32 = reinterpret_cast<float*>(blk + 2048);
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int co = lane & 31, ci = 16 * c + 8 * (lane >> 5) + i;
                    const float w = co < cout ? weight[((size_t)k * cin + ci) * cout + co] : 0.f;
                    if (!(fabsf(w) < 15.99f)) return GPNERF_E_ARG;          // the packed range (also refuses NaN)
                    const float ws = w * W16_SCALE;
                    const _Float16 h = (_Float16)ws;
                    hi[lane * 8 + i] = h;
                    lo[lane * 8 + i] = (_Float16)(ws - (float)h);
                    f32[lane * 8 + i] = ws;
                }
        }
    return GPNERF_OK;
}

static int sparse_conv16(int32_t strided, const float* in, int32_t cin, const int32_t* in_grid, const int32_t* in_dims,
                         const int32_t* out_coords, const int32_t* m_dev, int32_t m_cap, const void* packed_weight16, int32_t cout,
                         const float* bn_scale, const float* bn_shift, float* out, float* vol, void* stream);

int gpnerf_sparse_conv3_mfma16(int32_t strided, const float* in, int32_t cin, const int32_t* in_grid, const int32_t* in_dims,
                               const int32_t* out_coords, const int32_t* m_dev, int32_t m_cap, const void* packed_weight16, int32_t cout,
                               const float* bn_scale, const float* bn_shift, float* out, void* stream) {
    return sparse_conv16(strided, in, cin, in_grid, in_dims, out_coords, m_dev, m_cap, packed_weight16, cout, bn_scale, bn_shift, out, nullptr, stream);
}

static int sparse_conv16(int32_t strided, const float* in, int32_t cin, const int32_t* in_grid, const int32_t* in_dims,
                         const int32_t* out_coords, const int32_t* m_dev, int32_t m_cap, const void* packed_weight16, int32_t cout,
                         const float* bn_scale, const float* bn_shift, float* out, float* vol, void* stream) {
    if (!in || !in_grid || bad(in_dims) || !out_coords || !packed_weight16 || !bn_scale || !bn_shift || !out) return GPNERF_E_ARG;
    if ((cin != 16 && cin != 32) || cout < 1 || cout > 32 || m_cap < 0) return GPNERF_E_ARG;
    if (m_cap == 0) return GPNERF_OK;
    const Dims s{in_dims[0], in_dims[1], in_dims[2]};
    const dim3 grid((unsigned)((m_cap + 31) / 32)), block(256);         // one workgroup (4 wavefronts, taps dealt out) per 32 sites
    const unsigned char* wp = static_cast<const unsigned char*>(packed_weight16);
#define GPNERF_LAUNCH16(ST, NC_)                                                                                                  \
    hipLaunchKernelGGL((conv_mfma16_kernel<ST, NC_>), grid, block, 0, S_(stream), in, in_grid, s, out_coords, (const int*)m_dev, (int)m_cap, \
                       wp, (int)cout, bn_scale, bn_shift, out, vol)
    if (strided) { if (cin == 32) GPNERF_LAUNCH16(true, 2); else GPNERF_LAUNCH16(true, 1); }
    else { if (cin == 32) GPNERF_LAUNCH16(false, 2); else GPNERF_LAUNCH16(false, 1); }
#undef GPNERF_LAUNCH16
    return status();
}

int gpnerf_sparse_conv3_mfma(int32_t strided, const float* in, int32_t cin, const int32_t* in_grid, const int32_t* in_dims,
                             const int32_t* out_coords, const int32_t* m_dev, int32_t m_cap, const float* packed_weight, int32_t cout,
                             const float* bn_scale, const float* bn_shift, float* out, void* stream) {
    if (!in || !in_grid || bad(in_dims) || !out_coords || !packed_weight || !bn_scale || !bn_shift || !out) return GPNERF_E_ARG;
    if (cin < 8 || cin > 32 || cin % 8 || cout < 1 || cout > 32 || m_cap < 0) return GPNERF_E_ARG;
    if (m_cap == 0) return GPNERF_OK;
    const Dims s{in_dims[0], in_dims[1], in_dims[2]};
    const dim3 grid((unsigned)((m_cap + 31) / 32)), block(256);         // one workgroup (4 wavefronts, taps dealt out) per 32 sites
    if (strided)
        hipLaunchKernelGGL(conv_mfma_kernel<true>, grid, block, 0, S_(stream), in, (int)cin, in_grid, s, out_coords, (const int*)m_dev,
                           (int)m_cap, packed_weight, (int)cout, bn_scale, bn_shift, out);
    else
        hipLaunchKernelGGL(conv_mfma_kernel<false>, grid, block, 0, S_(stream), in, (int)cin, in_grid, s, out_coords, (const int*)m_dev,
                           (int)m_cap, packed_weight, (int)cout, bn_scale, bn_shift, out);
    return status();
}

int gpnerf_sparse_down_sites(const int32_t* in_coords, const int32_t* m_in_dev, int32_t m_in_cap, const int32_t* out_dims,
                             int32_t* out_grid, int32_t* out_coords, int32_t* m_out_dev, int32_t m_out_cap, void* stream) {
    if (!in_coords || bad(out_dims) || !out_grid || !out_coords || !m_out_dev || m_in_cap < 0 || m_out_cap < 0) return GPNERF_E_ARG;
    const Dims s{out_dims[0], out_dims[1], out_dims[2]};
    const long cells = (long)s.d * s.h * s.w;
    if (!fill_async(out_grid, sizeof(int32_t) * (size_t)cells, 0xFFFFFFFFu, stream) || !fill_async(m_out_dev, sizeof(int32_t), 0u, stream))
        return GPNERF_E_LAUNCH;
    if (m_in_cap == 0) return GPNERF_OK;
    hipLaunchKernelGGL(mark_kernel, dim3((m_in_cap + 255) / 256), dim3(256), 0, S_(stream), in_coords, (const int*)m_in_dev,
                       (int)m_in_cap, s, out_grid);
    hipLaunchKernelGGL(assign_kernel, dim3((unsigned)((cells + 256 * ASSIGN_PER - 1) / (256 * ASSIGN_PER))), dim3(256), 0, S_(stream), s, out_grid, (int*)m_out_dev,
                       (int)m_out_cap, out_coords);
    return status();
}

int gpnerf_sparse_merge_duplicates(float* feat, int32_t channels, const int32_t* coords, const int32_t* grid, int32_t m,
                                   const int32_t* dims, int32_t* scratch, void* stream) {
    if (!feat || !coords || !grid || !scratch || bad(dims) || channels < 1 || channels > 32 || m < 0) return GPNERF_E_ARG;
    if (m == 0) return GPNERF_OK;
    const Dims s{dims[0], dims[1], dims[2]};
    if (!fill_async(scratch, sizeof(int32_t) * (size_t)m, 0u, stream)) return GPNERF_E_LAUNCH;     // the counts
    hipLaunchKernelGGL(count_duplicates_kernel, dim3((m + 255) / 256), dim3(256), 0, S_(stream), coords, grid, (int)m, s, scratch);
    hipLaunchKernelGGL(merge_duplicates_kernel, dim3((m + 3) / 4), dim3(256), 0, S_(stream), feat, (int)channels, coords, grid,
                       (int)m, s, (const int32_t*)scratch);
    return status();
}

int gpnerf_zero_volume(float* vol_ndhwc, int32_t channels, const int32_t* dims, void* stream) {
    if (!vol_ndhwc || bad(dims) || channels < 1) return GPNERF_E_ARG;
    return fill_async(vol_ndhwc, sizeof(float) * (size_t)dims[0] * dims[1] * dims[2] * channels, 0u, stream) ? GPNERF_OK : GPNERF_E_LAUNCH;
}

int gpnerf_sparse_to_dense(const float* feat, int32_t channels, const int32_t* coords, const int32_t* grid, const int32_t* m_dev,
                           int32_t m_cap, const int32_t* dims, float* vol_ndhwc, void* stream) {
    return gpnerf_sparse_scatter_dense(feat, channels, coords, grid, m_dev, m_cap, dims, vol_ndhwc, 0, stream);
}

int gpnerf_sparse_scatter_dense(const float* feat, int32_t channels, const int32_t* coords, const int32_t* grid, const int32_t* m_dev,
                                int32_t m_cap, const int32_t* dims, float* vol_ndhwc, int32_t prezeroed, void* stream) {
    if (!feat || !coords || !grid || bad(dims) || !vol_ndhwc || channels < 1 || channels > 32 || m_cap < 0) return GPNERF_E_ARG;
    const Dims s{dims[0], dims[1], dims[2]};
    if (!prezeroed && !fill_async(vol_ndhwc, sizeof(float) * (size_t)s.d * s.h * s.w * channels, 0u, stream)) return GPNERF_E_LAUNCH;
    if (m_cap == 0) return GPNERF_OK;
    hipLaunchKernelGGL(dense_kernel, dim3((m_cap + 7) / 8), dim3(256), 0, S_(stream), feat, (int)channels, coords, grid,
                       (const int*)m_dev, (int)m_cap, s, vol_ndhwc);
    return status();
}

int gpnerf_sparse_pyramid_plan(const GpnerfPyramid* p, void* stream) {
    if (!p || p->n_levels < 1 || p->n_levels > GPNERF_PYRAMID_MAX_LEVELS || p->m0 < 0 || !p->coords0 || !p->grid0) return GPNERF_E_ARG;
    int rc = gpnerf_sparse_index(p->coords0, nullptr, p->m0, p->dims0, p->grid0, stream);
    if (rc != GPNERF_OK) return rc;
    const int32_t* coords = p->coords0;
    const int32_t* m_dev = nullptr;
    int32_t m_cap = p->m0;
    for (int i = 0; i < p->n_levels; ++i) {
        if (!p->grid[i] || !p->coords[i] || !p->m[i] || !p->vol[i]) return GPNERF_E_ARG;
        rc = gpnerf_sparse_down_sites(coords, m_dev, m_cap, p->dims[i], p->grid[i], p->coords[i], p->m[i], p->cap[i], stream);
        if (rc != GPNERF_OK) return rc;
        rc = gpnerf_zero_volume(p->vol[i], p->ch[i], p->dims[i], stream);
        if (rc != GPNERF_OK) return rc;
        coords = p->coords[i]; m_dev = p->m[i]; m_cap = p->cap[i];
    }
    return GPNERF_OK;
}

int gpnerf_sparse_pyramid_run(const GpnerfPyramid* p, const float* code, int32_t code_ch, const GpnerfSparseConv* convs, int32_t n_convs,
                              void* stream) {
    if (!p || !code || !convs || p->n_levels < 1 || p->n_levels > GPNERF_PYRAMID_MAX_LEVELS || n_convs != 2 + 3 * p->n_levels || !p->feat_a ||
        !p->feat_b || !p->dup_scratch)
        return GPNERF_E_ARG;
    auto conv = [&](const GpnerfSparseConv& c, const float* in, const int32_t* in_grid, const int32_t* in_dims, const int32_t* coords,
                    const int32_t* m_dev, int32_t m_cap, float* out) -> int {
        if (c.form == 2)
            return gpnerf_sparse_conv3_mfma16(c.strided, in, c.cin, in_grid, in_dims, coords, m_dev, m_cap, c.weight, c.cout, c.bn_scale, c.bn_shift,
                                              out, stream);
        if (c.form == 1)
            return gpnerf_sparse_conv3_mfma(c.strided, in, c.cin, in_grid, in_dims, coords, m_dev, m_cap, static_cast<const float*>(c.weight), c.cout,
                                            c.bn_scale, c.bn_shift, out, stream);
        return gpnerf_sparse_conv3(c.strided, in, c.cin, in_grid, in_dims, coords, m_dev, m_cap, static_cast<const float*>(c.weight), c.cout,
                                   c.bn_scale, c.bn_shift, out, stream);
    };
    if (convs[0].cin != code_ch || convs[0].strided || convs[1].strided) return GPNERF_E_ARG;
    if (!p->feat_c || !convs[0].weight_raw || !convs[1].weight_raw) return GPNERF_E_ARG;
    // The vertex level: two submanifold convolutions on rows that may SHARE voxels, in spconv's rulebook semantics (see
    // subm_shared_rows_kernel): each runs on a copy of its input whose owner rows hold their voxel's sum, then the rows whose
    // own voxel is shared are recomputed from their own features.
    const Dims d0 = {p->dims0[0], p->dims0[1], p->dims0[2]};
    auto subm_vertex = [&](const GpnerfSparseConv& c, const float* in, float* merged, float* out) -> int {
        const long n = (long)p->m0 * c.cin;
        if (n > 0) hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), in, merged, n);
        int r = gpnerf_sparse_merge_duplicates(merged, c.cin, p->coords0, p->grid0, p->m0, p->dims0, p->dup_scratch, stream);
        if (r != GPNERF_OK) return r;
        r = conv(c, merged, p->grid0, p->dims0, p->coords0, nullptr, p->m0, out);
        if (r != GPNERF_OK) return r;
        if (p->m0 > 0)
            hipLaunchKernelGGL(subm_shared_rows_kernel, dim3((unsigned)((p->m0 + 7) / 8)), dim3(256), 0, S_(stream), in, (const float*)merged, (int)c.cin,
                               (const int32_t*)p->grid0, d0, p->coords0, (int)p->m0, (const int32_t*)p->dup_scratch, c.weight_raw, (int)c.cout,
                               c.bn_scale, c.bn_shift, out);
        return status();
    };
    float* cur = p->feat_a;
    float* other = p->feat_b;
    int rc = subm_vertex(convs[0], code, p->feat_c, cur);
    if (rc != GPNERF_OK) return rc;
    rc = subm_vertex(convs[1], cur, p->feat_c, other);
    if (rc != GPNERF_OK) return rc;
    { float* t = cur; cur = other; other = t; }
    // the strided convolution takes EVERY input row: fold a voxel's rows into its owner, which is what the grid finds
    rc = gpnerf_sparse_merge_duplicates(cur, convs[1].cout, p->coords0, p->grid0, p->m0, p->dims0, p->dup_scratch, stream);
    if (rc != GPNERF_OK) return rc;
    const int32_t* grid = p->grid0;
    const int32_t* dims = p->dims0;
    for (int i = 0; i < p->n_levels; ++i) {
        const GpnerfSparseConv* c = convs + 2 + 3 * i;
        if (!c[0].strided || c[1].strided || c[2].strided || c[2].cout != p->ch[i]) return GPNERF_E_ARG;
        rc = conv(c[0], cur, grid, dims, p->coords[i], p->m[i], p->cap[i], other);               // reads the finer level at the coarser sites
        if (rc != GPNERF_OK) return rc;
        grid = p->grid[i]; dims = p->dims[i];
        rc = conv(c[1], other, grid, dims, p->coords[i], p->m[i], p->cap[i], cur);
        if (rc != GPNERF_OK) return rc;
        if (c[2].form == 2) {
            // the level's last convolution also writes its rows at their sites' cells of the (zeroed) dense volume: no scatter launch
            rc = sparse_conv16(0, cur, c[2].cin, grid, dims, p->coords[i], p->m[i], p->cap[i], c[2].weight, c[2].cout, c[2].bn_scale, c[2].bn_shift,
                               other, p->vol[i], stream);
            if (rc != GPNERF_OK) return rc;
            { float* t = cur; cur = other; other = t; }
        } else {
            rc = conv(c[2], cur, grid, dims, p->coords[i], p->m[i], p->cap[i], other);
            if (rc != GPNERF_OK) return rc;
            { float* t = cur; cur = other; other = t; }
            rc = gpnerf_sparse_scatter_dense(cur, c[2].cout, p->coords[i], grid, p->m[i], p->cap[i], dims, p->vol[i], 1, stream);
            if (rc != GPNERF_OK) return rc;
        }
    }
    return GPNERF_OK;
}

int gpnerf_vertex_attention(const float* q, const float* kv, const float* wq, const float* wk, const float* wv, const float* wfc,
                            int32_t n, int32_t d_model, int32_t kv_dim, int32_t n_head, int32_t views, float* out, void* stream) {
    if (n == 0) return GPNERF_OK;
    if (!q || !kv || !wq || !wk || !wv || !wfc || !out || n < 0) return GPNERF_E_ARG;
    if (d_model < 1 || d_model > ATT_MAX || kv_dim < 1 || kv_dim > ATT_MAX || n_head < 1 || d_model % n_head || views < 1 || views > 4)
        return GPNERF_E_ARG;
    const int d_k = d_model / n_head;
    if (d_k & (d_k - 1)) return GPNERF_E_ARG;          // the per-head reduction is a butterfly
    // two workgroups per CU: a wavefront reads its 4 x 32 weight rows once and walks ~3 vertices (6 890 vertices: 128 / 256 / 512 /
    // 768 / 1 024 / 1 723 workgroups: 52 / 32 / 30 / 37 / 40 / 57 us, tools/probes/attention_time.py)
    // the reference's shape: the matrix-core form (GPNERF_ATT_SHUFFLE=1 under GPNERF_DEBUG=1 keeps the shuffle form for comparison)
    if (d_model == 32 && kv_dim == 32 && d_k >= 4) {
        static int f_shuffle = -1;
        if (f_shuffle < 0) f_shuffle = dbg_int("GPNERF_ATT_SHUFFLE", 0, 0, 1);       // experiment knob (gpnerf_diag.h: 0 in the product)
        if (!f_shuffle) {
            const dim3 grid((unsigned)((n + 31) / 32)), block(64);
            hipStream_t st = reinterpret_cast<hipStream_t>(stream);
            if (views == 1) hipLaunchKernelGGL(vertex_attention_mfma_kernel<1>, grid, block, 0, st, q, kv, wq, wk, wv, wfc, (int)n, (int)n_head, out);
            else if (views == 2) hipLaunchKernelGGL(vertex_attention_mfma_kernel<2>, grid, block, 0, st, q, kv, wq, wk, wv, wfc, (int)n, (int)n_head, out);
            else if (views == 3) hipLaunchKernelGGL(vertex_attention_mfma_kernel<3>, grid, block, 0, st, q, kv, wq, wk, wv, wfc, (int)n, (int)n_head, out);
            else hipLaunchKernelGGL(vertex_attention_mfma_kernel<4>, grid, block, 0, st, q, kv, wq, wk, wv, wfc, (int)n, (int)n_head, out);
            return hipGetLastError() == hipSuccess ? GPNERF_OK : GPNERF_E_LAUNCH;
        }
    }
    int blocks = n < 2048 ? (n + 3) / 4 : 512;
    blocks = dbg_int("GPNERF_ATT_BLOCKS", blocks, 1, 4096);                         // experiment knob (gpnerf_diag.h)
    if (d_model <= 32 && kv_dim <= 32)
        hipLaunchKernelGGL(vertex_attention_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q,
                           kv, wq, wk, wv, wfc, (int)n, (int)d_model, (int)kv_dim, (int)n_head, (int)views, out);
    else
        hipLaunchKernelGGL(vertex_attention_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q,
                           kv, wq, wk, wv, wfc, (int)n, (int)d_model, (int)kv_dim, (int)n_head, (int)views, out);
    return hipGetLastError() == hipSuccess ? GPNERF_OK : GPNERF_E_LAUNCH;
}

}  // extern "C"
