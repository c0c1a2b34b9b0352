#!/usr/bin/env python3
"""Randomised sweep on the GPU: the fp32 forms' default launches (colour branch deferred sample by sample, empty-space exit of the
sigma feature layer) against the same launches with every layer evaluated (GPNERF_FLAG_NO_EXITS), bit for bit, over many seeded
scenes, sizes, sample counts and launch shapes.  usage: defer_sweep.py [n_cases]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fm = importlib.import_module("gp-nerf_amd.frame")
syn = importlib.import_module("gp-nerf_amd.synthetic")

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = np.random.Generator(np.random.PCG64(515))
launches = bad = 0
skipped = []
for case in range(n_cases):
    H = int(g.choice([12, 24, 33, 64, 96, 160, 300]))
    W = int(g.choice([12, 20, 40, 64, 96, 160, 300]))
    S = int(g.choice([1, 3, 8, 17, 32, 64, 96, 128]))
    neg = bool(g.integers(0, 2))
    body = str(g.choice(["box", "capsules"]))
    kw = dict(H=H, W=W, seed=7000 + case, fill=str(g.choice(["full", "survey"])), pose=str(g.choice(["random", "identity"])),
              bias_std=0.15, sigma_bias=float(g.choice([-1.0, -0.3, 0.0, 0.5, 2.0, 60.0])), neg_cams=neg, body=body,
              vol_occupancy=(None if g.random() < 0.5 else float(g.choice([0.05, 0.3]))))
    if body == "box":
        kw["aabb_half"] = (0.1 + 0.1 * g.random(), 0.12 + 0.1 * g.random(), 0.04 + 0.04 * g.random())
    sc = syn.make_scene(**kw)
    n = sc["ray_o"].shape[1]
    if n == 0:
        continue
    fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                  sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
    rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
    order = torch.from_numpy(g.permutation(n).astype(np.int32)).to(dev)
    for fold in (False, True):
        for extra in ({}, {"load_balance": False}, {"early_term": True, "term_eps": 1e-4}, {"occ_cull": True}, {"ray_order": order},
                      {"early_term": True, "term_eps": 1e-4, "load_balance": False}):
            k2 = dict(neg_ray=neg, fold=fold, want=("weights", "z_vals", "rgb_in", "ray_mask", "step_stats"), **extra)
            a = fm.render_fused(fr, rays, S, **k2)
            b = fm.render_fused(fr, rays, S, exits=False, **k2)
            st = a.pop("step_stats").cpu().numpy()
            b.pop("step_stats")
            launches += 1
            skipped.append(st[2] / max(1, st[0]))
            for k in a:
                if not torch.equal(torch.nan_to_num(a[k].float()), torch.nan_to_num(b[k].float())):
                    bad += 1
                    print("DIFFERS", case, kw, "fold" if fold else "ref", extra if "ray_order" not in extra else "ray_order", k,
                          float((torch.nan_to_num(a[k].float()) - torch.nan_to_num(b[k].float())).abs().max()))
sk = np.array(skipped)
print(f"{n_cases} scenes, {launches} pairs of launches (reference-order and folded form; plain, unbalanced, early termination, culling, "
      f"permuted order): {bad} outputs differ; colour passes skipped per launch: min {sk.min():.2f} median {np.median(sk):.2f} max {sk.max():.2f}")
print("DEFER SWEEP OK" if bad == 0 else "DEFER SWEEP FAILED")
sys.exit(0 if bad == 0 else 1)
