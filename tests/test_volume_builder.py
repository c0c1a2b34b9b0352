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
