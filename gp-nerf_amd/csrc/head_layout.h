// head_layout.h -- the LDS image of the per-ray MLP ("head blob"), shared by the host
// packer (gpnerf_pack_head) and the device kernels.
//
// The MLP runs transposed on v_mfma_f32_32x32x2_f32:  H_out[feat][ray] = W[feat][in] * H_in[in][ray].
//   A operand = weights   (lane l supplies W[row = l&31][k = k_of(step, l>>5)])
//   B operand = activations, one VGPR per k-step (lane l supplies H_in[k_of(step, l>>5)][ray = l&31])
//   C/D       = 32 features x 32 rays, feature f held by (reg r, half h) with f = (r&3) + 8*(r>>2) + 4*h
// Because the accumulator already has "ray on the lane, features in registers", the output tile of
// one layer IS the B operand of the next: k-step t of a tile-sourced input consumes register t of both
// halves, i.e. features ft(t,0) and ft(t,1).  The weight columns are permuted at pack time to match,
// so activations never leave the register file.
#pragma once

namespace gpl {

constexpr int GEO = 0, D1 = 1, D2 = 2, D3 = 3, BS = 4, BV = 5, B2 = 6, V1 = 7, V2 = 8, R1 = 9, R2 = 10, NLAYER = 11;
// k-steps (K/2, K padded to even) and 32-row output tiles of each MFMA layer
//                              GEO  D1  D2  D3  BS  BV  B2  V1  V2  R1  R2
constexpr int NT[NLAYER] = {64, 68, 32, 16, 36, 18, 32, 16, 16, 48, 16};
constexpr int MT[NLAYER] = {2, 2, 1, 1, 2, 2, 1, 1, 1, 1, 1};

constexpr int w_off(int l) {
    int o = 0;
    for (int j = 0; j < l; ++j) o += NT[j] * MT[j] * 64;
    return o;
}
constexpr int W_TOTAL = w_off(NLAYER);
constexpr int b_off(int l) {
    int o = W_TOTAL;
    for (int j = 0; j < l; ++j) o += MT[j] * 32;
    return o;
}
constexpr int B_END = b_off(NLAYER);
// VALU tails: density 16->1 and colour 16->3, weights as [out][half][8], then biases
constexpr int D4_W = B_END;         // 16 floats
constexpr int D4_B = D4_W + 16;     // 1 (+3 pad)
constexpr int R3_W = D4_B + 4;      // 48 floats
constexpr int R3_B = R3_W + 48;     // 3 (+1 pad)
constexpr int BLOB_FLOATS = R3_B + 4;
static_assert(BLOB_FLOATS % 4 == 0, "blob is copied as float4");
static_assert(BLOB_FLOATS * 4 <= 160 * 1024, "head image must fit the 160 KiB LDS of one CU");

// feature index held by accumulator register r of lane-half h (32x32 C/D layout)
constexpr int ft(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// index into the reference's 35-vector [r,g,b, feat0..31] of slot t (0..17) of half h; -1 = zero pad
constexpr int idx35(int t, int h) {
    return t < 16 ? 3 + 16 * h + t : (t == 16 ? (h == 0 ? 0 : 1) : (h == 0 ? 2 : -1));
}

// PyTorch weight column consumed by k-step t, lane-half h of layer l (-1: zero)
constexpr int col_of(int l, int t, int h) {
    switch (l) {
        case GEO: return 32 * (t >> 4) + 16 * h + (t & 15);                       // [level][channel]
        case D1:  return t < 32 ? 32 * (t >> 4) + ft(t & 15, h)
                       : (t < 50 ? (idx35(t - 32, h) < 0 ? -1 : 64 + idx35(t - 32, h))
                                 : (idx35(t - 50, h) < 0 ? -1 : 99 + idx35(t - 50, h)));
        case BS:  return t < 18 ? (idx35(t, h) < 0 ? -1 : idx35(t, h))
                                : (idx35(t - 18, h) < 0 ? -1 : 35 + idx35(t - 18, h));
        case BV:  return idx35(t, h) < 0 ? -1 : 70 + idx35(t, h);
        case D2: case B2: case R1: return 32 * (t >> 4) + ft(t & 15, h);
        default:  return ft(t, h);                                                  // D3 V1 V2 R2
    }
}

}  // namespace gpl

// ------------------------------------------------------------------------------------------------------------------
// Reference-order image (gpnerf_pack_head_ref; the fp32 form without folded volumes, GPNERF_FLAG_REF_ORDER): the same tiles,
// offsets and sizes as above, but every layer accumulates in the order the reference's sgemm does -- k ASCENDING from zero,
// bias added last -- in the unscaled domain, so that on trained parameters the kernel's rounding stays correlated with the
// reference's through the whole chain (oracle/kernel_order.inc measures what each deviation costs).
//   k-step t of a layer consumes k = 2t (lane half 0) and k = 2t + 1 (half 1): v_mfma_f32_32x32x2_f32 is fma(a1, b1, fma(a0, b0, c));
//   accumulator register r of half h therefore has to hold OUTPUT feature 2r + h of its tile (tile row ft(r, h) carries the
//   weights of feature 32 m + 2 r + h), and the output tile of one layer is again the B operand of the next, in order;
//   raw inputs are re-interleaved in registers (v_permlane32_swap: channels c, 16 + c of the two halves -> 2i, 2i + 1);
//   the 35-vectors [r, g, b, f0 .. f31] take 18 k-steps as (r, g) (b, 0) (f0, f1) ... (f30, f31): the zero of step 1 adds
//   fma(0, 0, s) = s, which keeps the chain's bits and re-aligns the parity of everything behind it.
// ------------------------------------------------------------------------------------------------------------------
namespace gpr {

using gpl::GEO; using gpl::D1; using gpl::D2; using gpl::D3; using gpl::BS; using gpl::BV; using gpl::B2;
using gpl::V1; using gpl::V2; using gpl::R1; using gpl::R2;

// output feature (within its 32-row tile) carried by tile row i: i = ft(r, h)  ->  2 r + h
constexpr int feat_of_row(int i) { return 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); }
// index into the reference's 35-vector of slot t (0..17) of half h; -1 = the zero pad of step 1
constexpr int ref35(int t, int h) { return t == 0 ? h : (t == 1 ? (h == 0 ? 2 : -1) : 3 + 2 * (t - 2) + h); }
// PyTorch weight column consumed by k-step t, lane-half h of layer l (-1: zero)
constexpr int col_of(int l, int t, int h) {
    switch (l) {
        case D1: return t < 32 ? 2 * t + h : (t < 50 ? (ref35(t - 32, h) < 0 ? -1 : 64 + ref35(t - 32, h))
                                                     : (ref35(t - 50, h) < 0 ? -1 : 99 + ref35(t - 50, h)));
        case BS: return t < 18 ? ref35(t, h) : (ref35(t - 18, h) < 0 ? -1 : 35 + ref35(t - 18, h));
        case BV: return ref35(t, h) < 0 ? -1 : 70 + ref35(t, h);
        default: return 2 * t + h;                  // GEO (raw, re-interleaved) and every tile-sourced layer
    }
}

}  // namespace gpr

// ------------------------------------------------------------------------------------------------------------------
// Split-precision image (GPNERF_FLAG_SPLIT_F16): the same layers on v_mfma_f32_32x32x16_f16 with every fp32 operand
// written as hi + lo in f16 (hi = f16(x) toward zero, lo = f16(x - hi)) and three MFMAs per k-step:
//   W.h ~= Whi.hhi + Whi.hlo + Wlo.hhi        (the dropped lo.lo term is ~2^-22 relative; f32 accumulation)
// which keeps ~fp32 accuracy (measured 1e-6 on rgb) at 3/16 of the fp32-MFMA cost.
//   A operand: lane l holds W[row l&31][k = 8*(l>>5) + j], j = 0..7      (8 halfs = one ds_read_b128)
//   B operand: lane l holds H[k = 8*(l>>5) + j][ray l&31]
//   C/D      : as the fp32 form, feature ft(r,h) in accumulator register r of half h
// A 32-feature accumulator tile feeds the next layer as two 16-deep k-steps: step u takes registers 8u..8u+7.
// ------------------------------------------------------------------------------------------------------------------
namespace gph {

using gpl::GEO; using gpl::D1; using gpl::D2; using gpl::D3; using gpl::BS; using gpl::BV; using gpl::B2;
using gpl::V1; using gpl::V2; using gpl::R1; using gpl::R2; using gpl::NLAYER; using gpl::MT; using gpl::ft;

//                              GEO  D1  D2  D3  BS  BV  B2  V1  V2  R1  R2      k-steps of 16
constexpr int NS[NLAYER] = {8, 10, 4, 2, 6, 3, 4, 2, 2, 6, 2};
constexpr int STEP_WORDS = 512;          // one k-step of one tile: 64 lanes x 8 halfs hi (256 words) + the same for lo

constexpr int w_off(int l) {             // in 32-bit words
    int o = 0;
    for (int j = 0; j < l; ++j) o += NS[j] * MT[j] * STEP_WORDS;
    return o;
}
constexpr int W_TOTAL = w_off(NLAYER);
constexpr int b_off(int l) {             // fp32 biases, [tile][half][16] as in the fp32 image
    int o = W_TOTAL;
    for (int j = 0; j < l; ++j) o += MT[j] * 32;
    return o;
}
constexpr int B_END = b_off(NLAYER);
constexpr int D4_W = B_END, D4_B = D4_W + 16, R3_W = D4_B + 4, R3_B = R3_W + 48;
constexpr int BLOB_WORDS = R3_B + 4;
static_assert(BLOB_WORDS % 4 == 0, "image is copied as 16-byte pieces");
static_assert(BLOB_WORDS * 4 <= 160 * 1024, "split image must fit the 160 KiB LDS of one CU");

// slot of the 35-vector [r,g,b, feat0..31] held by half h at (k-step u of 3, element j); -1 = zero pad
constexpr int x35(int u, int j, int h) {
    return u < 2 ? 3 + 16 * h + 8 * u + j : (j == 0 ? (h == 0 ? 0 : 1) : (j == 1 ? (h == 0 ? 2 : -1) : -1));
}
constexpr int off_or_pad(int base, int idx) { return idx < 0 ? -1 : base + idx; }
constexpr int tile_col(int s, int j, int h) { return 32 * (s >> 1) + ft(8 * (s & 1) + j, h); }

// PyTorch weight column consumed at (layer l, k-step s, element j, half h); -1: zero
constexpr int col_of(int l, int s, int j, int h) {
    switch (l) {
        case GEO: return 32 * (s >> 1) + 16 * h + 8 * (s & 1) + j;
        case D1:  return s < 4 ? tile_col(s, j, h) : (s < 7 ? off_or_pad(64, x35(s - 4, j, h)) : off_or_pad(99, x35(s - 7, j, h)));
        case BS:  return s < 3 ? off_or_pad(0, x35(s, j, h)) : off_or_pad(35, x35(s - 3, j, h));
        case BV:  return off_or_pad(70, x35(s, j, h));
        default:  return tile_col(s, j, h);          // D2 B2 R1 (several tiles) and D3 V1 V2 R2 (one tile)
    }
}

}  // namespace gph
