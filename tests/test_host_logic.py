"""CPU: host-side logic that needs no GPU."""
import importlib

import numpy as np


def test_patch_order_is_a_permutation_of_the_hit_list():
    fm = importlib.import_module("gp-nerf_amd.frame")
    m = np.ones(16 * 64, bool)
    o = fm.patch_order(m, 16, 64)
    assert o.dtype == np.int32 and sorted(o.tolist()) == list(range(1024))
    # first wavefront = 32 pixels of the first row of patch (0,0); second = the row below it
    assert o[:32].tolist() == list(range(32)) and o[32:64].tolist() == list(range(64, 96))
    g = np.random.Generator(np.random.PCG64(2))
    m = g.random(20 * 70) > 0.4
    o = fm.patch_order(m, 20, 70)
    assert sorted(o.tolist()) == list(range(int(m.sum())))


def test_synthetic_scene_is_deterministic_and_follows_the_batch_schema(syn):
    a = syn.make_scene(H=16, W=16, seed=5, aabb_half=(0.12, 0.16, 0.05))
    b = syn.make_scene(H=16, W=16, seed=5, aabb_half=(0.12, 0.16, 0.05))
    for k in ("ray_o", "near", "src_imgs", "featmaps", "coord"):
        assert np.array_equal(a[k], b[k])
    assert a["ray_o"].shape == (1, 256, 3) and a["near"].shape == (1, 256) and a["src_imgs"].shape == (1, 3, 3, 16, 16)
    assert a["out_sh"].shape == (1, 3) and (a["out_sh"] % 32 == 0).all()
    assert [v.shape[1] for v in a["volumes"]] == [32] * 4
    assert (a["far"] > a["near"]).all()
    s = syn.make_scene(H=16, W=16, seed=5, aabb_half=(0.12, 0.16, 0.05), vol_occupancy=0.3)
    assert all((v >= 0).all() for v in s["volumes"]) and 0.05 < (s["volumes"][0] > 0).mean() < 0.6


def test_renderer_refuses_training_mode():
    import pytest
    render = importlib.import_module("gp-nerf_amd.render")
    head = importlib.import_module("gp-nerf_amd.head")
    with pytest.raises(Exception, match="inference-only"):
        render.Renderer(None, head.NeRFHead(code_dim=32), is_train=True)


def test_patch_order_device_equals_host_version():
    """frame.patch_order_device (torch ops, used by the progressive renderer) is the same permutation as patch_order."""
    import importlib
    import numpy as np
    import torch
    fm = importlib.import_module("gp-nerf_amd.frame")
    g = np.random.Generator(np.random.PCG64(1))
    for H, W in ((16, 16), (33, 40), (24, 7)):
        m = g.random((H * W,)) < 0.6
        a = fm.patch_order(m, H, W, patch_w=4, patch_h=8)
        b = fm.patch_order_device(torch.from_numpy(m), H, W).numpy()
        assert np.array_equal(a, b) and sorted(a.tolist()) == list(range(int(m.sum())))
