#!/usr/bin/env python3
"""Wall time of Renderer.render(batch) with the per-frame producers on the ZJU-sized frame of SURVEY.md 8d (512x512 sources,
literal f = 1.05 W camera: ~74 k rays x 64 samples).  usage: time_survey_api.py [n_calls]   (rocprofv3 --kernel-trace friendly)"""
import importlib
import os
import sys
import time
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gp-nerf_amd", "plugins")]
syn = importlib.import_module("gp-nerf_amd.synthetic")
hip_render = importlib.import_module("hip_render")
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
S = 64
cfg = NS(encoder=NS(file="hip_encoder", name="resnet34", out_ch=32),
         head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32] * 4)),
         dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3),
         train=NS(n_rays=1024, n_samples=S), test=NS(mesh_th=50))
dev = "cuda:0"
torch.manual_seed(0)            # (the renderer's head and encoder are random-initialised: the densities, and with them the time, follow the seed)
r = hip_render.build_render(cfg).to(dev).eval()
sc = syn.make_scene(H=512, W=512, seed=0, fill="survey", pose="identity", make_volumes=False)
keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk", "mask_at_box")
b = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in keys}
with torch.no_grad():
    for _ in range(3):
        r.render(b)
    ts, et = [], []
    for _ in range(n_calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = r.render(b)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        et.append(o["etime"] * 1e3)
print(f"survey frame: {sc['ray_o'].shape[1]} rays x {S}: Renderer.render wall median {np.median(ts):.3f} ms (min {min(ts):.3f}), etime {np.median(et):.3f} ms")
