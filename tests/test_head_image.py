"""CPU: gpnerf_pack_head's LDS image, read back the way the kernel reads it and pushed through an
emulation of v_mfma_f32_32x32x2_f32's operand/accumulator maps, reproduces NeRFHead.forward (oracle).

This pins the weight permutation (head_layout.h) without a GPU: any slip in col_of / ft / the
[group][lane][4] image order changes the result.
"""
import ctypes as C
import importlib

import numpy as np

NAMES = ["GEO", "D1", "D2", "D3", "BS", "BV", "B2", "V1", "V2", "R1", "R2"]


def ft(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


class Wave:
    """32 samples on 64 lanes: lane = sample + 32*half."""

    def __init__(self, blob, table):
        self.blob = blob
        self.lay = {n: tuple(table[4 * i:4 * i + 4]) for i, n in enumerate(NAMES)}
        self.tail = tuple(table[44:48])

    def bias_tile(self, name, m):
        nt, mt, wo, bo = self.lay[name]
        acc = np.zeros((16, 64), np.float32)            # [reg][lane]
        for h in range(2):
            acc[:, 32 * h:32 * h + 32] = self.blob[bo + m * 32 + h * 16: bo + m * 32 + h * 16 + 16][:, None]
        return acc

    def mfma_tile(self, name, m, b, acc):
        """b: [NT][64 lanes] B-operand registers; acc [16][64] accumulator registers."""
        nt, mt, wo, bo = self.lay[name]
        w = self.blob[wo + m * nt * 64:]
        ng = nt // 4
        D = np.zeros((32, 32), np.float64)              # [row = feature][col = sample]
        for t in range(nt):
            g = t // 4
            if g < ng:
                a = np.array([w[(g * 64 + lane) * 4 + (t & 3)] for lane in range(64)])
            else:
                a = np.array([w[ng * 256 + lane * 2 + (t - 4 * ng)] for lane in range(64)])
            for h in range(2):                          # A[i][k=h] = a[i + 32h], B[k=h][j] = b[j + 32h]
                D += np.outer(a[32 * h:32 * h + 32].astype(np.float64), b[t][32 * h:32 * h + 32].astype(np.float64))
        out = acc.copy()
        for r in range(16):
            for h in range(2):
                out[r, 32 * h:32 * h + 32] += D[ft(r, h), :].astype(np.float32)
        return out


LOG2E = np.float32(1.4426950408889634)


def elu(x):
    """The kernels' scaled-domain ELU (elus() in gpnerf_kernels.hip): accumulators hold log2(e) * (W h + b), activations
    log2(e) * ELU(.) -- the packers scale biases, raw-input weight columns and the VALU tails to match."""
    x = x.astype(np.float32)
    return np.where(x > 0, x, (np.exp2(np.minimum(x, 0)) - np.float32(1)) * LOG2E).astype(np.float32)


def emulate(blob, table, vol_feat, rgb_feat, mask):
    """vol_feat [32,128], rgb_feat [32,3,35], mask [32,3] -> raw [32,4] following mlp_eval's register dataflow."""
    w = Wave(blob, table)
    lane_s = np.arange(64) % 32
    lane_h = np.arange(64) // 32
    fv = np.zeros((64, 64), np.float32)
    for t in range(64):
        fv[t] = vol_feat[lane_s, 32 * (t >> 4) + 16 * lane_h + (t & 15)]
    x = np.zeros((3, 18, 64), np.float32)
    for v in range(3):
        for t in range(16):
            x[v, t] = rgb_feat[lane_s, v, 3 + 16 * lane_h + t]
        x[v, 16] = np.where(lane_h == 1, rgb_feat[lane_s, v, 1], rgb_feat[lane_s, v, 0])
        x[v, 17] = np.where(lane_h == 1, 0.0, rgb_feat[lane_s, v, 2])
    nvalid = mask.sum(1)[lane_s]
    g0 = w.mfma_tile("GEO", 0, fv, w.bias_tile("GEO", 0))
    g1 = w.mfma_tile("GEO", 1, fv, w.bias_tile("GEO", 1))
    d1in = np.concatenate([elu(g0), elu(g1)], 0)
    m = (x[0] + x[1] + x[2]) * np.float32(1 / 3)
    var = ((x[0] - m) ** 2 + (x[1] - m) ** 2 + (x[2] - m) ** 2) * np.float32(1 / 3)
    mv = np.concatenate([m, var], 0)
    d1in = np.concatenate([d1in, mv], 0)
    a0 = w.mfma_tile("D1", 0, d1in, w.bias_tile("D1", 0))
    a1 = w.mfma_tile("D1", 1, d1in, w.bias_tile("D1", 1))
    h1 = np.concatenate([elu(a0), elu(a1)], 0)
    a2 = w.mfma_tile("D2", 0, h1, w.bias_tile("D2", 0))
    a3 = w.mfma_tile("D3", 0, elu(a2), w.bias_tile("D3", 0))
    d4w, d4b, r3w, r3b = w.tail
    e3 = elu(a3)
    part = np.zeros(64, np.float32)
    for r in range(8):
        part += blob[d4w + lane_h * 8 + r] * e3[r]
    s = part + part[(np.arange(64) + 32) % 64] + blob[d4b]
    sigma = np.where(nvalid < 1, 0.0, np.maximum(s, 0))
    s0 = w.mfma_tile("BS", 0, mv, w.bias_tile("BS", 0))
    s1 = w.mfma_tile("BS", 1, mv, w.bias_tile("BS", 1))
    y = []
    for v in range(3):
        b0 = w.mfma_tile("BV", 0, x[v], s0)
        b1 = w.mfma_tile("BV", 1, x[v], s1)
        hh = np.concatenate([elu(b0), elu(b1)], 0)
        xb = elu(w.mfma_tile("B2", 0, hh, w.bias_tile("B2", 0)))
        t1 = elu(w.mfma_tile("V1", 0, xb, w.bias_tile("V1", 0)))      # the packed weights carry the 1 / num_views
        t2 = elu(w.mfma_tile("V2", 0, t1, w.bias_tile("V2", 0)))
        y.append(xb + t2)
    y = np.concatenate(y, 0)
    c1 = elu(w.mfma_tile("R1", 0, y, w.bias_tile("R1", 0)))
    c2 = elu(w.mfma_tile("R2", 0, c1, w.bias_tile("R2", 0)))
    rgb = []
    for o in range(3):
        part = np.zeros(64, np.float32)
        for r in range(8):
            part += blob[r3w + o * 16 + lane_h * 8 + r] * c2[r]
        s = part + part[(np.arange(64) + 32) % 64] + blob[r3b + o]
        rgb.append(1 / (1 + np.exp(-s)))
    raw = np.stack(rgb + [sigma], 1)
    assert np.array_equal(raw[:32], raw[32:]), "both lane halves must hold the same result"
    return raw[:32].astype(np.float32)


def test_head_image_reproduces_the_reference_head(pkg, oracle, syn):
    L = pkg._lib
    lib = L.lib()
    head = syn.make_head_weights(seed=3, bias_std=0.2)
    params = L.GpnerfHeadParams()
    keep = []
    for short, name in L.HEAD_FIELDS:
        for suf, fld in (("weight", "_w"), ("bias", "_b")):
            a = np.ascontiguousarray(head[f"{name}.{suf}"], np.float32)
            keep.append(a)
            setattr(params, short + fld, a.ctypes.data_as(L.FP))
    blob = np.zeros(lib.gpnerf_head_blob_floats(), np.float32)
    assert lib.gpnerf_pack_head(C.byref(params), blob.ctypes.data_as(L.FP)) == 0
    table = (C.c_int32 * 48)()
    assert lib.gpnerf_head_layout(table) == 0
    g = np.random.Generator(np.random.PCG64(12))
    vol = g.standard_normal((32, 128), dtype=np.float32)
    feat = g.standard_normal((32, 3, 35), dtype=np.float32)
    feat[..., :3] = g.random((32, 3, 3), dtype=np.float32)
    mask = (g.random((32, 3)) > 0.4).astype(np.float32)
    mask[:3] = 0
    got = emulate(blob, list(table), vol, feat, mask)
    ref = oracle.head_forward(head, vol, feat, mask)
    err = np.abs(got - ref).max()
    assert err < 2e-5, err
    assert (ref[:3, 3] == 0).all() and (got[:3, 3] == 0).all()       # masked_fill(num_valid_obs < 1, 0)


def elu_ref(x):
    """elur_n's activation (gpnerf_kernels.hip): unscaled, x > 0 ? x : 2^(x log2 e) - 1"""
    x = x.astype(np.float32)
    return np.where(x > 0, x, np.exp2(np.minimum(x, 0) * LOG2E) - np.float32(1)).astype(np.float32)


def div3(x):
    """div3() of the kernel: q = x c, q + fma(-3, q, x) c -- must equal IEEE x / 3"""
    x = x.astype(np.float32)
    c = np.float32(1 / 3)
    q = (x * c).astype(np.float32)
    r = (x.astype(np.float64) - 3.0 * q.astype(np.float64)).astype(np.float32)          # fma(-3, q, x) is exact in double
    return (q.astype(np.float64) + r.astype(np.float64) * np.float64(c)).astype(np.float32)


def interleave16(regs):
    """interleave16(): 16 registers [16][64], register c = channel c (lanes 0..31) / 16 + c (lanes 32..63) ->
    out[i] = channels (2i, 2i + 1), out[8 + i] = (16 + 2i, 17 + 2i); v_permlane32_swap semantics as measured
    (tools/micro/permlane32_swap.hip): (a, b) -> ([a.lo, b.lo], [a.hi, b.hi])"""
    out = np.zeros_like(regs)
    for i in range(8):
        a, b = regs[2 * i], regs[2 * i + 1]
        out[i] = np.concatenate([a[:32], b[:32]])
        out[8 + i] = np.concatenate([a[32:], b[32:]])
    return out


def emulate_ref(blob, table, vol_feat, rgb_feat, mask):
    """The reference-order form's register dataflow (mlp_eval_ref / geo_eval_ref / render_tile's FORM_F32 branch): chains from
    zero, bias after the chain, features re-interleaved to k order.  Any slip in gpr::col_of / feat_of_row / the slot
    arrangement changes the result."""
    w = Wave(blob, table)
    lane_s = np.arange(64) % 32
    lane_h = np.arange(64) // 32
    zero = np.zeros((16, 64), np.float32)
    fv = np.zeros((64, 64), np.float32)
    for l in range(4):
        g = np.zeros((16, 64), np.float32)
        for c in range(16):
            g[c] = vol_feat[lane_s, 32 * l + 16 * lane_h + c]              # what gather_volume leaves in register c
        fv[16 * l:16 * l + 16] = interleave16(g)
    x = np.zeros((3, 18, 64), np.float32)
    for v in range(3):
        g = np.zeros((16, 64), np.float32)
        for c in range(16):
            g[c] = rgb_feat[lane_s, v, 3 + 16 * lane_h + c]
        x[v, 2:18] = interleave16(g)
        x[v, 0] = np.where(lane_h == 1, rgb_feat[lane_s, v, 1], rgb_feat[lane_s, v, 0])
        x[v, 1] = np.where(lane_h == 1, 0.0, rgb_feat[lane_s, v, 2])
    nvalid = mask.sum(1)[lane_s]
    act = lambda name, m, acc: elu_ref(acc + w.bias_tile(name, m))
    sf = np.concatenate([act("GEO", 0, w.mfma_tile("GEO", 0, fv, zero)), act("GEO", 1, w.mfma_tile("GEO", 1, fv, zero))], 0)
    m = div3((x[0] + x[1]) + x[2])
    var = div3(((x[0] - m) ** 2 + (x[1] - m) ** 2) + (x[2] - m) ** 2)
    mv = np.concatenate([m, var], 0)
    d1in = np.concatenate([sf, mv], 0)
    h1 = np.concatenate([act("D1", 0, w.mfma_tile("D1", 0, d1in, zero)), act("D1", 1, w.mfma_tile("D1", 1, d1in, zero))], 0)
    h2 = act("D2", 0, w.mfma_tile("D2", 0, h1, zero))
    h3 = act("D3", 0, w.mfma_tile("D3", 0, h2, zero))
    d4w, d4b, r3w, r3b = w.tail
    part = np.zeros(64, np.float32)
    for r in range(8):
        part += blob[d4w + lane_h * 8 + r] * h3[r]
    s = part + part[(np.arange(64) + 32) % 64] + blob[d4b]
    sigma = np.where(nvalid < 1, 0.0, np.maximum(s, 0))
    s0 = w.mfma_tile("BS", 0, mv, zero)
    s1 = w.mfma_tile("BS", 1, mv, zero)
    y = []
    for v in range(3):
        hh = np.concatenate([act("BS", 0, w.mfma_tile("BV", 0, x[v], s0)), act("BS", 1, w.mfma_tile("BV", 1, x[v], s1))], 0)
        xb = act("B2", 0, w.mfma_tile("B2", 0, hh, zero))
        t1 = act("V1", 0, w.mfma_tile("V1", 0, div3(xb), zero))
        t2 = act("V2", 0, w.mfma_tile("V2", 0, t1, zero))
        y.append(xb + t2)
    y = np.concatenate(y, 0)
    c1 = act("R1", 0, w.mfma_tile("R1", 0, y, zero))
    c2 = act("R2", 0, w.mfma_tile("R2", 0, c1, zero))
    rgb = []
    for o in range(3):
        part = np.zeros(64, np.float32)
        for r in range(8):
            part += blob[r3w + o * 16 + lane_h * 8 + r] * c2[r]
        s = part + part[(np.arange(64) + 32) % 64] + blob[r3b + o]
        rgb.append(1 / (1 + np.exp(-s)))
    raw = np.stack(rgb + [sigma], 1)
    assert np.array_equal(raw[:32], raw[32:]), "both lane halves must hold the same result"
    return raw[:32].astype(np.float32)


def test_reference_order_image_reproduces_the_reference_head(pkg, oracle, syn):
    """gpnerf_pack_head_ref's image through the reference-order form's dataflow = NeRFHead.forward"""
    L = pkg._lib
    lib = L.lib()
    head = syn.make_head_weights(seed=5, bias_std=0.3)
    params = L.GpnerfHeadParams()
    keep = []
    for short, name in L.HEAD_FIELDS:
        for suf, fld in (("weight", "_w"), ("bias", "_b")):
            a = np.ascontiguousarray(head[f"{name}.{suf}"], np.float32)
            keep.append(a)
            setattr(params, short + fld, a.ctypes.data_as(L.FP))
    blob = np.zeros(lib.gpnerf_head_blob_floats(), np.float32)
    assert lib.gpnerf_pack_head_ref(C.byref(params), blob.ctypes.data_as(L.FP)) == 0
    table = (C.c_int32 * 48)()
    assert lib.gpnerf_head_layout(table) == 0
    g = np.random.Generator(np.random.PCG64(13))
    vol = g.standard_normal((32, 128), dtype=np.float32)
    feat = g.standard_normal((32, 3, 35), dtype=np.float32)
    feat[..., :3] = g.random((32, 3, 3), dtype=np.float32)
    mask = (g.random((32, 3)) > 0.4).astype(np.float32)
    mask[:3] = 0
    got = emulate_ref(blob, list(table), vol, feat, mask)
    ref = oracle.head_forward(head, vol, feat, mask)
    err = np.abs(got - ref).max()
    assert err < 2e-5, err
    assert (ref[:3, 3] == 0).all() and (got[:3, 3] == 0).all()
    # nothing in the image is pre-scaled: every weight of the state_dict appears in it bit for bit
    for short, name in L.HEAD_FIELDS:
        wts = np.ascontiguousarray(head[f"{name}.weight"], np.float32).ravel()
        assert np.isin(wts, blob).all(), name


def test_div3_is_ieee_division():
    """q = x c; q + fma(-3, q, x) c is the correctly rounded x / 3 (what torch.mean computes) on random and edge values"""
    g = np.random.Generator(np.random.PCG64(7))
    x = np.concatenate([g.standard_normal(2_000_000).astype(np.float32) * np.float32(10.0) ** g.integers(-20, 20, 2_000_000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, 3.0, 1e-30, 3e38, -3e38, np.float32(2 ** -120)], np.float32)])
    assert np.array_equal(div3(x), (x / np.float32(3)).astype(np.float32))


def test_pack_rejects_missing_tensors(pkg):
    L = pkg._lib
    blob = np.zeros(L.lib().gpnerf_head_blob_floats(), np.float32)
    assert L.lib().gpnerf_pack_head(C.byref(L.GpnerfHeadParams()), blob.ctypes.data_as(L.FP)) == -1


# ---- split-precision image (GPNERF_FLAG_SPLIT_F16) ------------------------------------------------------------------
NS16 = [8, 10, 4, 2, 6, 3, 4, 2, 2, 6, 2]
MT16 = [2, 2, 1, 1, 2, 2, 1, 1, 1, 1, 1]


def split16(v):
    """hi = f16 toward zero, lo = f16(v - hi), as make_frag does on the device."""
    v = np.asarray(v, np.float32)
    hi = v.astype(np.float16)
    away = (np.abs(hi.astype(np.float32)) > np.abs(v))
    hi = np.where(away, (hi.view(np.uint16) - 1).view(np.float16), hi)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


class Wave16:
    """v_mfma_f32_32x32x16_f16 operand maps: A/B lane l holds k = 8*(l>>5)+j; C/D feature ft(r,h) in register r."""

    def __init__(self, blob):
        self.words = blob
        self.halfs = blob.view(np.float16)
        off, self.w_off = 0, {}
        for n, ns, mt in zip(NAMES, NS16, MT16):
            self.w_off[n] = off
            off += ns * mt * 512
        self.b_off = {}
        for n, mt in zip(NAMES, MT16):
            self.b_off[n] = off
            off += mt * 32
        self.d4w, self.d4b, self.r3w, self.r3b = off, off + 16, off + 20, off + 68

    def bias_tile(self, name, m):
        acc = np.zeros((16, 64), np.float32)
        bo = self.b_off[name]
        for h in range(2):
            acc[:, 32 * h:32 * h + 32] = self.words[bo + m * 32 + h * 16: bo + m * 32 + h * 16 + 16][:, None]
        return acc

    def steps(self, name, m, s0, frags, acc):
        """frags: list of [8][64] fp32 slot arrays (one per k-step)."""
        ns = NS16[NAMES.index(name)]
        D = np.zeros((32, 32), np.float64)
        for i, fr in enumerate(frags):
            base = (self.w_off[name] + (m * ns + s0 + i) * 512) * 2
            ah = self.halfs[base: base + 512].astype(np.float32).reshape(64, 8)
            al = self.halfs[base + 512: base + 1024].astype(np.float32).reshape(64, 8)
            bh, bl = split16(fr)                                   # [8][64]
            for h in range(2):
                A_h, A_l = ah[32 * h:32 * h + 32], al[32 * h:32 * h + 32]          # [row][j]
                B_h, B_l = bh[:, 32 * h:32 * h + 32], bl[:, 32 * h:32 * h + 32]    # [j][col]
                D += A_l.astype(np.float64) @ B_h + A_h.astype(np.float64) @ B_l + A_h.astype(np.float64) @ B_h
        out = acc.copy()
        for r in range(16):
            for h in range(2):
                out[r, 32 * h:32 * h + 32] += D[ft(r, h), :].astype(np.float32)
        return out


def emulate_split(blob, vol_feat, rgb_feat, mask):
    w = Wave16(blob)
    lane_s, lane_h = np.arange(64) % 32, np.arange(64) // 32
    fv = np.stack([vol_feat[lane_s, 32 * (t >> 4) + 16 * lane_h + (t & 15)] for t in range(64)], 0)
    x = np.zeros((3, 18, 64), np.float32)
    for v in range(3):
        for t in range(16):
            x[v, t] = rgb_feat[lane_s, v, 3 + 16 * lane_h + t]
        x[v, 16] = np.where(lane_h == 1, rgb_feat[lane_s, v, 1], rgb_feat[lane_s, v, 0])
        x[v, 17] = np.where(lane_h == 1, 0.0, rgb_feat[lane_s, v, 2])
    nvalid = mask.sum(1)[lane_s]
    z6 = np.zeros((6, 64), np.float32)
    tile2 = lambda a: [a[0:8], a[8:16]]
    x3 = lambda a: [a[0:8], a[8:16], np.concatenate([a[16:18], z6], 0)]
    fvf = [fv[8 * s: 8 * s + 8] for s in range(8)]
    g0 = w.steps("GEO", 0, 0, fvf, w.bias_tile("GEO", 0))
    g1 = w.steps("GEO", 1, 0, fvf, w.bias_tile("GEO", 1))
    sff = tile2(elu(g0)) + tile2(elu(g1))
    m = (x[0] + x[1] + x[2]) * np.float32(1 / 3)
    var = ((x[0] - m) ** 2 + (x[1] - m) ** 2 + (x[2] - m) ** 2) * np.float32(1 / 3)
    mvf = x3(m) + x3(var)
    a0 = w.steps("D1", 0, 4, mvf, w.steps("D1", 0, 0, sff, w.bias_tile("D1", 0)))
    a1 = w.steps("D1", 1, 4, mvf, w.steps("D1", 1, 0, sff, w.bias_tile("D1", 1)))
    a2 = w.steps("D2", 0, 0, tile2(elu(a0)) + tile2(elu(a1)), w.bias_tile("D2", 0))
    a3 = w.steps("D3", 0, 0, tile2(elu(a2)), w.bias_tile("D3", 0))
    e3 = elu(a3)
    part = sum(blob[w.d4w + lane_h * 8 + r] * e3[r] for r in range(8)).astype(np.float32)
    s = part + part[(np.arange(64) + 32) % 64] + blob[w.d4b]
    sigma = np.where(nvalid < 1, 0.0, np.maximum(s, 0))
    s0 = w.steps("BS", 0, 0, mvf, w.bias_tile("BS", 0))
    s1 = w.steps("BS", 1, 0, mvf, w.bias_tile("BS", 1))
    yf = []
    for v in range(3):
        b0 = w.steps("BV", 0, 0, x3(x[v]), s0)
        b1 = w.steps("BV", 1, 0, x3(x[v]), s1)
        xb = elu(w.steps("B2", 0, 0, tile2(elu(b0)) + tile2(elu(b1)), w.bias_tile("B2", 0)))
        t1 = elu(w.steps("V1", 0, 0, tile2(xb), w.bias_tile("V1", 0)))
        t2 = elu(w.steps("V2", 0, 0, tile2(t1), w.bias_tile("V2", 0)))
        yf += tile2(xb + t2)
    c1 = elu(w.steps("R1", 0, 0, yf, w.bias_tile("R1", 0)))
    c2 = elu(w.steps("R2", 0, 0, tile2(c1), w.bias_tile("R2", 0)))
    rgb = []
    for o in range(3):
        part = sum(blob[w.r3w + o * 16 + lane_h * 8 + r] * c2[r] for r in range(8)).astype(np.float32)
        s = part + part[(np.arange(64) + 32) % 64] + blob[w.r3b + o]
        rgb.append(1 / (1 + np.exp(-s)))
    raw = np.stack(rgb + [sigma], 1)
    assert np.array_equal(raw[:32], raw[32:])
    return raw[:32].astype(np.float32)


def test_split_head_image_reproduces_the_reference_head(pkg, oracle, syn):
    L = pkg._lib
    lib = L.lib()
    head = syn.make_head_weights(seed=4, bias_std=0.2)
    params = L.GpnerfHeadParams()
    keep = []
    for short, name in L.HEAD_FIELDS:
        for suf, fld in (("weight", "_w"), ("bias", "_b")):
            a = np.ascontiguousarray(head[f"{name}.{suf}"], np.float32)
            keep.append(a)
            setattr(params, short + fld, a.ctypes.data_as(L.FP))
    n = lib.gpnerf_head_blob_split_floats()
    assert n * 4 <= 160 * 1024
    blob = np.zeros(n, np.float32)
    assert lib.gpnerf_pack_head_split(C.byref(params), blob.ctypes.data_as(L.FP)) == 0
    g = np.random.Generator(np.random.PCG64(13))
    vol = g.standard_normal((32, 128), dtype=np.float32) * 2
    feat = g.standard_normal((32, 3, 35), dtype=np.float32)
    feat[..., :3] = g.random((32, 3, 3), dtype=np.float32)
    mask = (g.random((32, 3)) > 0.4).astype(np.float32)
    mask[:3] = 0
    got = emulate_split(blob, vol, feat, mask)
    ref = oracle.head_forward(head, vol, feat, mask)
    err = np.abs(got - ref).max()
    assert err < 3e-5, err          # hi/lo split keeps ~22 significant bits: fp32-class accuracy
