#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: the HIP path (fp32 and split-precision forms, with and without neg_ray / occupancy
culling / sample-split geometry) against the CPU oracle over many seeded scenes.  Prints the worst error per output.
The oracle is the checker here (tools/ is test infrastructure, like tests/)."""
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

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = np.random.Generator(np.random.PCG64(2024))
worst = {}
flips = 0
for case in range(n_cases):
    H = int(g.choice([12, 16, 24, 33]))
    W = int(g.choice([12, 16, 20, 40]))
    S = int(g.choice([1, 3, 8, 17, 32, 64, 96]))
    neg = bool(g.integers(0, 2))
    kw = dict(H=H, W=W, seed=1000 + case, fill=str(g.choice(["full", "survey"])), pose=str(g.choice(["random", "identity"])),
              aabb_half=(0.1 + 0.1 * g.random(), 0.12 + 0.1 * g.random(), 0.04 + 0.04 * g.random()), bias_std=0.15,
              sigma_bias=float(g.choice([0.0, 0.5])), neg_cams=neg, focal_mul=float(g.choice([0.6, 1.05, 2.0])))
    sc = syn.make_scene(**kw)
    n = sc["ray_o"].shape[1]
    if n == 0:
        continue
    fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                  sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
    rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
    ref = oracle.render(sc, S, neg_ray=neg)
    for split in (False, True):
        for lb in (False, True):
            got = {k: v.cpu().numpy() for k, v in fm.render_fused(fr, rays, S, neg_ray=neg, split_f16=split, load_balance=lb).items()}
            for k in ("rgb_map", "depth_map", "acc_map", "rgb_in_map", "weights"):
                a, b = got[k].astype(np.float64), ref[k].astype(np.float64)
                ok = ~(np.isnan(a) | np.isnan(b))
                assert np.array_equal(np.isnan(a), np.isnan(b)), (case, k, "NaN pattern")
                e = float(np.abs(a[ok] - b[ok]).max()) if ok.any() else 0.0
                key = ("split " if split else "fp32  ") + k
                if e > worst.get(key, (0.0,))[0]:
                    worst[key] = (e, case, kw["seed"], H, W, S, neg, lb)
            flips += int((got["ray_mask"] != ref["ray_mask"]).sum())
print(f"{n_cases} scenes; worst max-abs error per output (error, case, seed, H, W, S, neg_ray, load_balance):")
for k in sorted(worst):
    print(f"  {k:18s} {worst[k][0]:.2e}  {worst[k][1:]}")
print("ray_mask flips:", flips)
bad = {k: v for k, v in worst.items() if v[0] > 1e-4}
print("SWEEP OK" if not bad and flips == 0 else f"SWEEP FAILED: {bad}")
sys.exit(0 if not bad and flips == 0 else 1)
