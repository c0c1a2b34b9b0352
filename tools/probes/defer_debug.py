"""Diagnostic: deferred colour branch against the launch that evaluates everything, form by form (where do they differ?)."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
L = importlib.import_module("gp-nerf_amd._lib")
if os.environ.get("GPNERF_DIAG_LIB"):          # a diagnostic build in place of the product library, this process only
    L.LIB_PATH = os.path.join(ROOT, "gp-nerf_amd", "csrc", os.environ["GPNERF_DIAG_LIB"])
fm = importlib.import_module("gp-nerf_amd.frame")
syn = importlib.import_module("gp-nerf_amd.synthetic")
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sc = syn.make_scene(H=72, W=72, seed=91, fill="full", pose="random", aabb_half=(0.2, 0.3, 0.12), bias_std=0.1, sigma_bias=-0.3)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
for name, kw in (("ref", {}), ("fold", {"fold": True}), ("split", {"split_f16": True, "guard": False}), ("guard", {"split_f16": True})):
    for n, S in ((5, 1), (64, 1), (64, 8), (2000, 64)):
        r = rays[:n].contiguous()
        a = fm.render_fused(r, r, S, **kw) if False else fm.render_fused(fr, r, S, want=("weights", "guard_tiles") if "split_f16" in kw and kw.get("guard") is not False else ("weights",), **kw)
        b = fm.render_fused(fr, r, S, want=("weights", "raw"), **kw)        # `raw` keeps the colour branch in the step
        d = (a["rgb_map"] - b["rgb_map"]).abs()
        rel = d / (b["rgb_map"].abs() + 1e-12)
        print(f"{name:6s} n={n:5d} S={S:3d}: rays differing {int((d > 0).any(1).sum()):5d}, max abs {float(d.max()):.3e}, max rel {float(rel.max()):.3e}, "
              f"weights equal {torch.equal(a['weights'], b['weights'])}, guard_tiles {int(a['guard_tiles']) if 'guard_tiles' in a else '-'}")
