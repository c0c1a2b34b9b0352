#!/usr/bin/env python3
"""Single-GPU time of ONE rank's share of the strong-scaling plan (round-robin bands of the patch-ordered ray list), i.e. what a rank
of `bench.py --gpus N` renders per frame, without the all-gather.  usage: share_time.py [size] [samples]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fm = importlib.import_module("gp-nerf_amd.frame")
syn = importlib.import_module("gp-nerf_amd.synthetic")
par = importlib.import_module("gp-nerf_amd.parallel")

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
sc = syn.make_scene(H=size, W=size, seed=0, fill="full", pose="identity")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
patch = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], size, size)).to(dev)
rays_all = rays.index_select(0, patch.long())


def timed(r, reps=20):
    for _ in range(3):
        fm.render_fused(fr, r, S, want=(), fold=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fm.render_fused(fr, r, S, want=(), fold=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


full = timed(rays_all)
print(f"{size}x{size}x{S}: whole frame {full:.3f} ms")
for world in (2, 4, 8):
    plan = par.plan_for(rays_all.shape[0], world, dev)
    ts = [timed(plan.take(rays_all, rank).contiguous()) for rank in (0, world - 1)]
    print(f"  world {world}: a rank's share ({plan.share} rays) {ts[0]:.3f} / {ts[1]:.3f} ms (rank 0 / last) -> {full / max(ts):.2f}x before the gather")
