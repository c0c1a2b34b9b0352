#!/usr/bin/env python3
"""Randomised sweep of the segmented early-termination form (per-ray termination, re-packing) on the GPU: frames of at least
one round of wavefronts with random sizes, sample counts, poses, density biases, neg_ray, both kernel forms, against the CPU
oracle's UNTERMINATED render on a ray sample (bound: 1e-4, term_eps 1e-5), plus invariants over every ray (sum of weights =
acc, weights zero after the stop, same bits in a second run).  The oracle is the checker (tools/ is test infrastructure)."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fm = importlib.import_module("gp-nerf_amd.frame")
syn = importlib.import_module("gp-nerf_amd.synthetic")
from oracle import oracle  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = np.random.Generator(np.random.PCG64(77))
worst = {}
eps = 1e-5
for case in range(n_cases):
    H = int(g.choice([64, 80, 96, 128])); W = int(g.choice([64, 96, 112]))
    S = int(g.choice([17, 32, 48, 64, 100, 128, 200]))
    neg = bool(g.integers(0, 2))
    split = bool(g.integers(0, 2))
    sc = syn.make_scene(H=H, W=W, seed=3000 + case, fill="full", pose=str(g.choice(["random", "identity"])), bias_std=0.15,
                        sigma_bias=float(g.choice([0.0, 0.5, 1.0, 2.0])), neg_cams=neg)
    base = np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32)
    n = int(g.integers(65536, 150000))
    idx = g.integers(0, base.shape[0], size=n)
    rays_h = base[idx]
    fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                  sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
    rays = t(rays_h)
    want = ("weights", "z_vals", "rgb_in", "samples_done")
    kw = dict(neg_ray=neg, early_term=True, term_eps=eps, split_f16=split, want=want)
    a = fm.render_fused(fr, rays, S, **kw)
    b = fm.render_fused(fr, rays, S, **kw)
    for k in a:
        assert torch.equal(a[k].view(torch.int32) if a[k].dtype == torch.float32 else a[k], b[k].view(torch.int32) if b[k].dtype == torch.float32 else b[k]), (case, k, "not deterministic")
    w, done = a["weights"], a["samples_done"].long()
    assert float((w.sum(1) - a["acc_map"]).abs().max()) < 2e-5
    ks = torch.arange(S, device=dev)[None, :]
    assert float((w * (ks >= done[:, None])).abs().max()) == 0.0, "weights after a ray's stop must be zero"
    assert int(done.min()) >= 1 and int(done.max()) <= S
    pick = g.choice(n, size=512, replace=False)
    ref = oracle.render(sc, S, neg_ray=neg, rays=rays_h[pick])
    far = float(rays_h[:, 7].max())
    for k in ("rgb_map", "depth_map", "acc_map", "rgb_in_map"):
        e = float(np.abs(a[k][torch.from_numpy(pick).to(dev)].cpu().numpy().reshape(ref[k].shape) - ref[k]).max())
        key = ("split " if split else "fp32  ") + k
        if e > worst.get(key, (0.0,))[0]:
            worst[key] = (e, case, H, W, S, n, neg, float(done.float().mean()) / S)
    print(f"case {case}: {n} rays x {S} samples, {'split' if split else 'fp32'}, neg_ray {neg}: evaluated {float(done.float().mean()) / S:.2f} of the samples, far {far:.2f}", flush=True)
print("worst max-abs error against the oracle's unterminated render (error, case, H, W, S, rays, neg_ray, evaluated fraction):")
for k in sorted(worst):
    print(f"  {k:18s} {worst[k][0]:.2e}  {worst[k][1:]}")
bad = {k: v for k, v in worst.items() if v[0] > 1e-4}
print("ET SWEEP OK" if not bad else f"ET SWEEP FAILED: {bad}")
sys.exit(0 if not bad else 1)
