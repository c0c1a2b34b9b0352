"""CPU: the rulebook sparse convolutions of oracle/producers_ref.py against a dense conv3d-with-mask formulation.

spconv itself is not available (parity unpinned at this boundary); this pins the oracle's restatement of
its published algorithm to the dense definition it must agree with.  The HIP volume builder is checked against that
restatement on the GPU (tests/test_gpu_renderer.py); the product has no torch path of its own."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from oracle import producers_ref as pref


def _rand_sparse(M, C, shape, seed):
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    keys = torch.randperm(D * H * W, generator=g)[:M]
    coords = torch.stack([keys // (H * W), (keys // W) % H, keys % W], 1)
    feats = torch.randn((M, C), generator=g)
    return pref.SparseTensor(feats, coords, shape)


def test_submanifold_conv_matches_dense_masked_conv():
    vol = importlib.import_module("gp-nerf_amd.volume")
    x = _rand_sparse(400, 6, (12, 16, 10), 1)
    conv = vol._SparseConv3d(6, 5, 3, subm=True)              # the product's parameter container
    y = pref.sparse_conv3d(x, conv.weight, subm=True)
    dense_in = x.dense()
    mask = (pref.SparseTensor(torch.ones(x.coords.shape[0], 1), x.coords, x.shape).dense() > 0).float()
    w = conv.weight.permute(4, 3, 0, 1, 2)                   # [Cout,Cin,kd,kh,kw]
    ref = F.conv3d(dense_in, w, padding=1) * mask
    assert torch.allclose(y.dense(), ref, atol=1e-5)


def test_strided_conv_matches_dense_conv_on_reachable_sites():
    vol = importlib.import_module("gp-nerf_amd.volume")
    x = _rand_sparse(300, 4, (16, 12, 20), 2)
    conv = vol._SparseConv3d(4, 7, 3, stride=2, padding=1)
    y = pref.sparse_conv3d(x, conv.weight, stride=2, padding=1)
    assert y.shape == (8, 6, 10)
    w = conv.weight.permute(4, 3, 0, 1, 2)
    ref = F.conv3d(x.dense(), w, stride=2, padding=1)
    mask_in = pref.SparseTensor(torch.ones(x.coords.shape[0], 1), x.coords, x.shape).dense()
    reach = (F.conv3d(mask_in, torch.ones(1, 1, 3, 3, 3), stride=2, padding=1) > 0).float()
    assert torch.allclose(y.dense(), ref * reach, atol=1e-5)
    got_mask = pref.SparseTensor(torch.ones(y.coords.shape[0], 1), y.coords, y.shape).dense()
    assert torch.equal(got_mask, reach)


def test_a_permuted_tap_order_is_caught():
    """oracle/producers_ref.py items 1-2: the dense conv3d check DISCRIMINATES -- the same rulebook with the taps flipped (true
    convolution instead of cross-correlation), with two kernel axes swapped, or with Cin / Cout transposed misses the dense
    reference by far more than the 1e-5 the right order meets.  (What it cannot tell: whether spconv v1.2.1 itself is the
    cross-correlation -- that is the header's recalled item 2.)"""
    vol = importlib.import_module("gp-nerf_amd.volume")
    x = _rand_sparse(500, 6, (12, 16, 10), 11)
    conv = vol._SparseConv3d(6, 6, 3, subm=True)
    w = conv.weight.detach()                                  # [kd,kh,kw,Cin,Cout]
    mask = (pref.SparseTensor(torch.ones(x.coords.shape[0], 1), x.coords, x.shape).dense() > 0).float()
    ref = F.conv3d(x.dense(), w.permute(4, 3, 0, 1, 2), padding=1) * mask
    right = float((pref.sparse_conv3d(x, w, subm=True).dense() - ref).abs().max())
    assert right < 1e-5
    wrong = {"flipped taps": torch.flip(w, (0, 1, 2)), "kd <-> kw": w.permute(2, 1, 0, 3, 4).contiguous(),
             "kh <-> kw": w.permute(0, 2, 1, 3, 4).contiguous(), "Cin <-> Cout": w.permute(0, 1, 2, 4, 3).contiguous()}
    for name, ww in wrong.items():
        err = float((pref.sparse_conv3d(x, ww, subm=True).dense() - ref).abs().max())
        assert err > 0.1, (name, err)
    # the strided form too
    xs = _rand_sparse(300, 4, (16, 12, 20), 12)
    cs = vol._SparseConv3d(4, 4, 3, stride=2, padding=1)
    ws = cs.weight.detach()
    mask_in = pref.SparseTensor(torch.ones(xs.coords.shape[0], 1), xs.coords, xs.shape).dense()
    reach = (F.conv3d(mask_in, torch.ones(1, 1, 3, 3, 3), stride=2, padding=1) > 0).float()
    refs = F.conv3d(xs.dense(), ws.permute(4, 3, 0, 1, 2), stride=2, padding=1) * reach
    assert float((pref.sparse_conv3d(xs, ws, 2, 1).dense() - refs).abs().max()) < 1e-5
    assert float((pref.sparse_conv3d(xs, torch.flip(ws, (0, 1, 2)), 2, 1).dense() - refs).abs().max()) > 0.1


def test_rows_that_share_a_voxel_follow_the_rulebook():
    """oracle/producers_ref.py items 4-6 on a hand-made case: three rows, two of them in ONE voxel.  The 1 x 1 submanifold
    convolution keeps them separate rows (features @ W); the strided convolution adds BOTH into the coarse site they reach --
    worked out by hand below; the k = 3 lookup finds the highest row of a voxel."""
    feats = torch.tensor([[1.0, 0.0], [0.0, 2.0], [3.0, 1.0]])
    coords = torch.tensor([[2, 2, 2], [2, 2, 2], [5, 4, 3]])
    x = pref.SparseTensor(feats, coords, (8, 8, 8))
    w1 = torch.tensor([[1.0, 2.0], [3.0, 4.0]]).view(1, 1, 1, 2, 2)
    # the k = 3 submanifold convolution, rulebook form: centre tap on every row's own features; the other taps only onto the OWNER
    # (row 1 of the shared voxel), carrying the sum of all rows of the neighbour voxel
    w = torch.zeros(3, 3, 3, 2, 1)
    w[1, 1, 1, :, 0] = torch.tensor([1.0, 1.0])             # centre
    w[1, 1, 0, :, 0] = torch.tensor([100.0, 0.0])           # tap (1,1,0): reads p + (0, 0, -1)
    xs = pref.SparseTensor(torch.tensor([[1.0, 0.0], [0.0, 2.0], [3.0, 1.0], [5.0, 0.0], [7.0, 0.0]]),
                           torch.tensor([[2, 2, 2], [2, 2, 2], [5, 4, 3], [2, 2, 1], [2, 2, 1]]), (8, 8, 8))
    yr = pref.subm_conv3d_rulebook(xs, w).features[:, 0]
    # rows 0, 1 share (2,2,2) [owner 1]; rows 3, 4 share (2,2,1) [owner 4] and are the (0,0,-1) neighbours of (2,2,2)
    assert yr.tolist() == [1.0, 2.0 + 100.0 * (5.0 + 7.0), 4.0, 5.0, 7.0]
    # one representative per voxel instead (rounds 2-4): every row of (2,2,2) sees row 4 alone, and its own voxel's owner at the centre
    ya = pref.sparse_conv3d(xs, w, subm=True).features[:, 0]
    assert ya.tolist() == [2.0 + 700.0, 2.0 + 700.0, 4.0, 7.0, 7.0]
    w3 = torch.zeros(3, 3, 3, 2, 1)
    w3[1, 1, 1, :, 0] = torch.tensor([1.0, 1.0])            # the tap that reads p = 2 o - 1 + (1,1,1) = 2 o
    w3[2, 1, 0, :, 0] = torch.tensor([10.0, 0.0])           # the tap that reads p = 2 o - 1 + (2,1,0) = 2 o + (1, 0, -1)
    y3 = pref.sparse_conv3d(x, w3, stride=2, padding=1)
    d = y3.dense()[0, 0]                                     # [4,4,4]
    # site (1,1,1): centre tap reads p = (2,2,2): both rows there -> (1 + 0) + (0 + 2) = 3;  tap (2,1,0) reads (3,2,1): empty
    assert float(d[1, 1, 1]) == 3.0
    # row 2 at (5,4,3): centre tap needs 2 o = (5,4,3): odd -> none;  tap (2,1,0): 2 o + (1,0,-1) = (5,4,3) -> o = (2,2,2): 10 * 3 = 30
    assert float(d[2, 2, 2]) == 30.0
    assert int((d != 0).sum()) == 2
    sk, order = torch.sort(x.keys(), stable=True)
    assert int(pref._lookup(sk, order, x.keys()[:1])) == 1   # the highest row of the shared voxel


def test_pyramid_shapes_and_keys():
    vol = importlib.import_module("gp-nerf_amd.volume")
    net = vol.SparseConvNet(n_layers=4, in_dim=8, out_dim=[32, 32, 32, 32]).eval()
    g = torch.Generator().manual_seed(3)
    coords = torch.randint(0, 32, (200, 3), generator=g)
    coord4 = torch.cat([torch.zeros(200, 1, dtype=torch.long), coords], 1)
    with torch.no_grad():
        levels = pref.dense_levels(net, torch.randn(200, 8, generator=g), coord4, (32, 64, 32))
    assert [tuple(l.shape) for l in levels] == [(1, 32, 16, 32, 16), (1, 32, 8, 16, 8), (1, 32, 4, 8, 4), (1, 32, 2, 4, 2)]
    keys = set(net.state_dict())
    assert "net.0.0.weight" in keys and "net.0.3.weight" in keys and "net.1.0.weight" in keys and "net.8.4.running_var" in keys


def test_the_product_has_no_torch_path_for_the_producers():
    """no CPU / PyTorch fallback -- CPU tensors and training mode are refused, not silently served."""
    vol = importlib.import_module("gp-nerf_amd.volume")
    enc = importlib.import_module("gp-nerf_amd.encoder")
    head = importlib.import_module("gp-nerf_amd.head")
    L = importlib.import_module("gp-nerf_amd._lib")
    net = vol.SparseConvNet(n_layers=4, in_dim=8, out_dim=[32, 32, 32, 32]).eval()
    assert not hasattr(net, "dense_levels")
    with pytest.raises(L.GpnerfError):
        net.net[0](None)
    with pytest.raises(L.GpnerfError):
        net.dense_levels_hip(torch.randn(10, 8), torch.zeros(10, 4, dtype=torch.long), (32, 64, 32))
    with pytest.raises(L.GpnerfError, match="multiple of 16"):
        net.dense_levels_hip(torch.randn(10, 8), torch.zeros(10, 4, dtype=torch.long), (33, 64, 32))
    att = vol.MultiHeadAttention(4, 16, 4, 4, kv_dim=32, sum=False).eval()
    with pytest.raises(L.GpnerfError):
        att(torch.randn(5, 1, 16), torch.randn(5, 3, 32), torch.randn(5, 3, 32))
    with pytest.raises(L.GpnerfError):
        enc.ResUNet("resnet34", 32).eval()(torch.randn(1, 3, 16, 16))
    h = head.NeRFHead(in_feat_ch=32, code_dim=32).eval()
    with pytest.raises(L.GpnerfError):
        h.sigmahead.build_volumes({"coord": None, "out_sh": (32, 64, 32), "batch_size": 1}, torch.randn(1, 6890, 3, 32))


def test_head_owns_the_reference_state_dict_keys():
    head = importlib.import_module("gp-nerf_amd.head")
    h = head.NeRFHead(in_feat_ch=32, code_dim=32)
    sd = h.state_dict()
    assert len(sd) == 115 and sum(p.numel() for p in h.parameters()) == 646788
    for k, shape in {"sigmahead.c.weight": (6890, 32), "sigmahead.out_geometry_fc.0.weight": (64, 128),
                     "rgbhead.base_fc.0.weight": (64, 105), "rgbhead.rgb_fc.4.bias": (3,),
                     "rgbhead.out_geometry_fc.6.weight": (1, 16), "sigmahead.xyzc_attn.w_ks.weight": (32, 32),
                     "sigmahead.xyzc_attn.layer_norm.bias": (32,), "sigmahead.xyzc_net.net.7.0.weight": (3, 3, 3, 32, 32)}.items():
        assert tuple(sd[k].shape) == shape, k
