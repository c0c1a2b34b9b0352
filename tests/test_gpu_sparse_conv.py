"""gpnerf_sparse_conv3_mfma16 (split-precision sparse convolution, gpnerf_volume.hip) against the fp32 matrix form and against a
float64 restatement of the rulebook on the same sites; the fp32 fall-back of a tap that meets a value beyond the f16 range."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

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


def _body_like_coords(seed, dims, n_vertices=3000):
    """A person-shaped vertex set on a small grid: a torso box, a head sphere and four capsule limbs sampled on their SURFACES
    (SMPL vertices are a surface mesh), quantised to voxels of a (D, H, W) grid and de-duplicated -- every active voxel holds ONE
    row, so no rule about rows that share a voxel enters."""
    g = np.random.default_rng(seed)
    D, H, W = dims

    def capsule(a, b, r, n):
        t = g.random(n)[:, None]
        axis = (b - a) / np.linalg.norm(b - a)
        v = g.normal(size=(n, 3))
        v -= (v @ axis)[:, None] * axis
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        return a + t * (b - a) + r * v

    c = np.array([D, H, W], np.float64) / 2
    sway = g.normal(size=(6, 3)) * np.array([D, H, W]) * 0.04
    parts = [capsule(c + [0, -0.10 * H, 0] + sway[0], c + [0, 0.22 * H, 0] + sway[1], 0.16 * min(D, W) * 2, n_vertices // 3),        # torso
             capsule(c + [0, 0.30 * H, 0] + sway[2], c + [0, 0.36 * H, 0] + sway[2], 0.09 * min(D, W) * 2, n_vertices // 8)]           # head
    for sx in (-1, 1):
        parts.append(capsule(c + [0, 0.20 * H, sx * 0.18 * W], c + [0, -0.05 * H, sx * 0.40 * W] + sway[3], 0.05 * W, n_vertices // 8))    # arm
        parts.append(capsule(c + [0, -0.10 * H, sx * 0.10 * W], c + [0, -0.44 * H, sx * 0.14 * W] + sway[4], 0.07 * W, n_vertices // 6))   # leg
    pts = np.clip(np.round(np.concatenate(parts)), 1, np.array([D, H, W]) - 2).astype(np.int64)
    return np.unique(pts, axis=0)


def test_the_builder_is_the_dense_conv3d_pyramid_where_no_voxel_is_shared():
    """spconv v1.2.1 is not in the reference tree, so the sparse pyramid cannot be pinned to it -- but wherever
    no two rows share a voxel NO recalled rule is involved: a submanifold convolution IS a dense conv3d restricted to the active
    sites, the strided one a dense stride-2 conv3d on the sites it can reach (oracle/producers_ref.py items 1-3, 5, 7, 8 reduce to
    torch.nn.functional.conv3d's own definition; item 6, rows sharing a voxel, is the one that stays recall-only).  Fifty random
    person-shaped vertex sets, one row per voxel, random weights and BatchNorm statistics per set: the HIP builder's four dense
    levels against the pyramid computed with F.conv3d in float64 on the densified input -- no sparse code, no rulebook."""
    vol = importlib.import_module("gp-nerf_amd.volume")
    dev = "cuda:0"
    dims = (32, 64, 32)
    worst = 0.0
    for seed in range(50):
        torch.manual_seed(1000 + seed)
        in_dim = (32, 16)[seed % 2]
        net = vol.SparseConvNet(n_layers=4, in_dim=in_dim, out_dim=[32, 32, 32, 32]).eval()
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
        coords = torch.from_numpy(_body_like_coords(seed, dims))
        assert 500 < coords.shape[0] == torch.unique(coords, dim=0).shape[0]
        code = torch.randn((coords.shape[0], in_dim))
        coord4 = torch.cat([torch.zeros((coords.shape[0], 1), dtype=coords.dtype), coords], 1)
        with torch.no_grad():
            hip = net.to(dev).dense_levels_hip(code.to(dev), coord4.to(dev), list(dims))
            net = net.cpu().double()
            # the dense pyramid: x [1,C,D,H,W], mask [1,1,D,H,W] of active sites
            x = torch.zeros((1, in_dim) + dims, dtype=torch.float64)
            x[0, :, coords[:, 0], coords[:, 1], coords[:, 2]] = code.double().t()
            mask = torch.zeros((1, 1) + dims, dtype=torch.float64)
            mask[0, 0, coords[:, 0], coords[:, 1], coords[:, 2]] = 1.0

            def block(seq, x, mask):
                mods = list(seq)
                for i in range(0, len(mods), 3):
                    conv, bn = mods[i], mods[i + 1]
                    w = conv.weight.permute(4, 3, 0, 1, 2)                    # [Cout, Cin, kd, kh, kw]: tap k reads the input at o * stride - pad + k
                    if conv.subm:
                        x = F.conv3d(x, w, padding=1)
                    else:
                        x = F.conv3d(x, w, stride=2, padding=1)
                        mask = (F.conv3d(mask, torch.ones((1, 1, 3, 3, 3), dtype=torch.float64), stride=2, padding=1) > 0).double()
                    x = (x - bn.running_mean.view(1, -1, 1, 1, 1)) / torch.sqrt(bn.running_var.view(1, -1, 1, 1, 1) + bn.eps) * bn.weight.view(1, -1, 1, 1, 1) \
                        + bn.bias.view(1, -1, 1, 1, 1)
                    x = F.relu(x) * mask                                       # BatchNorm1d + ReLU act on the rows = the active sites only
                return x, mask

            x, mask = block(net.net[0], x, mask)
            want = []
            for i in range(net.n_layers):
                x, mask = block(net.net[2 * i + 1], x, mask)
                x, mask = block(net.net[2 * i + 2], x, mask)
                want.append(x)
        for l, (a, b) in enumerate(zip(hip, want)):
            assert a.shape == tuple(b.shape[2:]) + (32,), (seed, l)
            err = float((a.permute(3, 0, 1, 2).cpu().double() - b[0]).abs().max()) / max(1.0, float(b.abs().max()))
            worst = max(worst, err)
            assert err < 1e-4, (seed, l, err)
            assert float((a != 0).float().mean()) > 0
    print(f"50 person-shaped vertex sets without shared voxels: HIP builder vs the dense float64 conv3d pyramid, worst relative max-abs {worst:.2e}")
