"""gpnerf_sparse_conv3_mfma16 (split-precision sparse convolution, gpnerf_volume.hip) against the fp32 matrix form and against a
float64 restatement of the rulebook on the same sites; the fp32 fall-back of a tap that meets a value beyond the f16 range."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cin, cout, m, dims, seed, strided):
    L = importlib.import_module("gp-nerf_amd._lib")
    lib = L.lib()
    g = np.random.default_rng(seed)
    cells = g.choice(dims[0] * dims[1] * dims[2], size=m, replace=False)
    coords = np.stack(np.unravel_index(cells, dims), 1).astype(np.int32)
    feat = g.standard_normal((m, cin)).astype(np.float32)
    w = (g.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)
    scale = g.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = (g.standard_normal(cout) * 0.2).astype(np.float32)
    if strided:
        odims = tuple(n // 2 for n in dims)
        oc = np.unique(coords // 2, axis=0).astype(np.int32)          # a subset of the reachable coarse sites is enough here
    else:
        odims, oc = dims, coords
    return L, lib, coords, feat, w, scale, shift, oc, odims


def _reference(coords, feat, w, scale, shift, oc, dims, strided):
    grid = -np.ones(dims, np.int64)
    grid[coords[:, 0], coords[:, 1], coords[:, 2]] = np.arange(len(coords))
    out = np.zeros((len(oc), w.shape[2]), np.float64)
    for k in range(27):
        kd, kh, kw = k // 9, (k // 3) % 3, k % 3
        p = (2 * oc if strided else oc).astype(np.int64) - 1 + np.array([kd, kh, kw])
        ok = np.all((p >= 0) & (p < np.array(dims)), 1)
        j = np.full(len(oc), -1)
        j[ok] = grid[p[ok, 0], p[ok, 1], p[ok, 2]]
        hit = j >= 0
        out[hit] += feat[j[hit]].astype(np.float64) @ w[k].astype(np.float64)
    return np.maximum(out * scale + shift, 0.0)


def _run(lib, L, fn16, coords, feat, w, scale, shift, oc, dims, strided):
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    I3 = C.c_int32 * 3
    cin, cout = w.shape[1], w.shape[2]
    cd, fd, ocd, sc, sh = t(coords), t(feat), t(oc), t(scale), t(shift)
    grid = torch.empty(dims, device=dev, dtype=torch.int32)
    L.check(lib.gpnerf_sparse_index(cd.data_ptr(), None, len(coords), I3(*dims), grid.data_ptr(), None), "index")
    out = torch.empty((len(oc), cout), device=dev)
    if fn16:
        packed = np.zeros(int(lib.gpnerf_sparse_packed_weight16_bytes(cin)), np.uint8)
        L.check(lib.gpnerf_sparse_pack_weight16(w.ctypes.data_as(L.FP), cin, cout, packed.ctypes.data_as(C.c_void_p)), "pack16")
        wp = t(packed)
        L.check(lib.gpnerf_sparse_conv3_mfma16(int(strided), fd.data_ptr(), cin, grid.data_ptr(), I3(*dims), ocd.data_ptr(), None, len(oc),
                                               wp.data_ptr(), cout, sc.data_ptr(), sh.data_ptr(), out.data_ptr(), None), "conv16")
    else:
        packed = np.zeros(int(lib.gpnerf_sparse_packed_weight_floats(cin)), np.float32)
        L.check(lib.gpnerf_sparse_pack_weight(w.ctypes.data_as(L.FP), cin, cout, packed.ctypes.data_as(L.FP)), "pack")
        wp = t(packed)
        L.check(lib.gpnerf_sparse_conv3_mfma(int(strided), fd.data_ptr(), cin, grid.data_ptr(), I3(*dims), ocd.data_ptr(), None, len(oc),
                                             wp.data_ptr(), cout, sc.data_ptr(), sh.data_ptr(), out.data_ptr(), None), "conv")
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("cin,cout,strided", [(32, 32, False), (32, 32, True), (16, 32, False), (16, 16, True), (32, 24, False)])
def test_split_precision_sparse_conv_matches_fp32_form_and_float64(cin, cout, strided):
    dims = (16, 24, 16)
    L, lib, coords, feat, w, scale, shift, oc, odims = _setup(cin, cout, 1500, dims, cin + cout + strided, strided)
    ref = _reference(coords, feat, w, scale, shift, oc, dims, strided)
    a = _run(lib, L, True, coords, feat, w, scale, shift, oc, dims, strided)
    b = _run(lib, L, False, coords, feat, w, scale, shift, oc, dims, strided)
    again = _run(lib, L, True, coords, feat, w, scale, shift, oc, dims, strided)
    assert np.array_equal(a, again)
    top = max(1.0, float(np.abs(ref).max()))
    # the split form is as close to the exact result as the fp32 instructions are (both ~1e-6 of the output range)
    assert np.abs(a - ref).max() < 4e-6 * top, np.abs(a - ref).max()
    assert np.abs(b - ref).max() < 4e-6 * top
    assert (a > 0).mean() > 0.2


def test_values_beyond_the_f16_range_take_the_fp32_instructions():
    dims = (16, 16, 16)
    L, lib, coords, feat, w, scale, shift, oc, odims = _setup(32, 32, 900, dims, 5, False)
    feat[::7, 3] = 5000.0                    # 16 x 5000 > 65 504: hi would be inf
    feat[5::11, 20] = -70000.0
    ref = _reference(coords, feat, w, scale, shift, oc, dims, False)
    a = _run(lib, L, True, coords, feat, w, scale, shift, oc, dims, False)
    b = _run(lib, L, False, coords, feat, w, scale, shift, oc, dims, False)
    assert np.isfinite(a).all()
    top = float(np.abs(ref).max())
    assert np.abs(a - ref).max() < 4e-6 * top and np.abs(b - ref).max() < 4e-6 * top


def test_pack_refuses_weights_beyond_the_packed_range():
    L = importlib.import_module("gp-nerf_amd._lib")
    lib = L.lib()
    w = np.zeros((27, 16, 16), np.float32)
    packed = np.zeros(int(lib.gpnerf_sparse_packed_weight16_bytes(16)), np.uint8)
    assert lib.gpnerf_sparse_packed_weight16_bytes(24) == 0 and len(packed) == 27 * 4096
    assert lib.gpnerf_sparse_pack_weight16(w.ctypes.data_as(L.FP), 16, 16, packed.ctypes.data_as(C.c_void_p)) == 0
    w[3, 2, 1] = 16.5
    assert lib.gpnerf_sparse_pack_weight16(w.ctypes.data_as(L.FP), 16, 16, packed.ctypes.data_as(C.c_void_p)) == -1
    w[3, 2, 1] = np.nan
    assert lib.gpnerf_sparse_pack_weight16(w.ctypes.data_as(L.FP), 16, 16, packed.ctypes.data_as(C.c_void_p)) == -1
