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


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(np.float32)


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
        t1 = elu(w.mfma_tile("V1", 0, xb * np.float32(1 / 3), w.bias_tile("V1", 0)))
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


def test_pack_rejects_missing_tensors(pkg):
    L = pkg._lib
    blob = np.zeros(L.lib().gpnerf_head_blob_floats(), np.float32)
    assert L.lib().gpnerf_pack_head(C.byref(L.GpnerfHeadParams()), blob.ctypes.data_as(L.FP)) == -1
