#!/usr/bin/env python3
"""End-to-end wall time of Renderer.render(batch) (the reference's call, BaseTrainer.py:267) on the synthetic 512x512x64
scene: per-frame producers (encoder, volume builder, frame build) + the per-ray kernel, as `etime` / `rtime` report it."""
import importlib
import os
import sys
import time
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gp-nerf_amd", "plugins"))
syn = importlib.import_module("gp-nerf_amd.synthetic")
hip_render = importlib.import_module("hip_render")

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = 64
cfg = NS(encoder=NS(file="hip_encoder", name="resnet34", out_ch=32),
         head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32] * 4)),
         dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3),
         train=NS(n_rays=1024, n_samples=S), test=NS(mesh_th=50))
dev = "cuda:0"
torch.manual_seed(0)            # (random-initialised head and encoder: the densities, and with them the time, follow the seed)
r = hip_render.build_render(cfg).to(dev).eval()
sc = syn.make_scene(H=size, W=size, seed=0, fill="full", pose="identity", make_volumes=False)
keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk")
batch = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in keys if k in sc}


def run(b, n=8):
    with torch.no_grad():
        for _ in range(3):
            r.render(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = rt = 0.0
        for _ in range(n):
            o = r.render(b)
            e += o["etime"]; rt += o["rtime"]
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, e / n * 1e3, rt / n * 1e3


wall, e, rt = run(batch)
print(f"full path (encoder + vertex features + sparse volume builder + frame + {size}x{size}x{S} rays): wall {wall:.2f} ms, etime {e:.2f} ms, rtime {rt:.2f} ms")
with torch.no_grad():
    fr = r.build_frame(batch)
    b2 = dict(batch, featmaps=r.encoder(batch["src_imgs"][0]), volumes=[v for v in fr.vols])
for v in b2["volumes"]:
    v._gpnerf_ndhwc = True
wall, e, rt = run(b2)
print(f"featmaps + volumes given: wall {wall:.2f} ms, etime {e:.2f} ms, rtime {rt:.2f} ms")

# the README command's renderer (render.file hip_demo_render): volume builder -> occupancy -> ray selection -> culled render
hip_demo = importlib.import_module("hip_demo_render")
rp = hip_demo.build_render(cfg).to(dev).eval()
rp.load_state_dict(r.state_dict(), strict=True)
bp = dict(batch)
for k in ("target_K", "target_pose", "target_K_inv"):
    bp[k] = torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev)
with torch.no_grad():
    for _ in range(3):
        o = rp.render(bp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    acc = {}
    for _ in range(n):
        o = rp.render(bp)
        for k, v in o["time_slots"].items():
            acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
print(f"progressive renderer: wall {wall:.2f} ms, etime {o['etime']*1e3:.2f} ms, rtime {o['rtime']*1e3:.2f} ms, {int(o['mask_at_box'].sum())} rays; "
      + ", ".join(f"{k} {acc[k] / n * 1e3:.2f}" for k in ("frame", "ray_select", "render", "bc_render")))
