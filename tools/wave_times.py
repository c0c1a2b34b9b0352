#!/usr/bin/env python3
"""Diagnostic: where a launch's fixed cost goes.  Per wavefront of ONE fused-kernel launch: when it entered the kernel, had its
weights in LDS, ended its first sample step and left (build: make -C gp-nerf_amd/csrc/diag libgpnerf_hip_wavetimes.so; loaded IN PLACE
of the product library for this process only).  usage: wave_times.py [size] [samples]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("gp-nerf_amd._lib")
L.LIB_PATH = os.path.join(ROOT, "gp-nerf_amd", "csrc", "diag", "libgpnerf_hip_wavetimes.so")
fm = importlib.import_module("gp-nerf_amd.frame")
syn = importlib.import_module("gp-nerf_amd.synthetic")

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
chain = len(sys.argv) > 3 and sys.argv[3] == "chain"       # segmented form with nothing terminating: the LAST segment's launch is what is read
kw = dict(early_term=True, term_eps=0.0) if chain else {}
REPS = 4
dev = torch.device("cuda:0")
sc = syn.make_scene(H=size, W=size, seed=0, fill="full", pose="identity")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
order = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], size, size)).to(dev)
lib = L.lib()
lib.gpnerf_debug_read_wavetimes.argtypes = [C.c_void_p, C.c_int]
n_waves = 256 * 8
for _ in range(REPS - 1):
    fm.render_fused(fr, rays, S, want=(), ray_order=order, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fm.render_fused(fr, rays, S, want=(), ray_order=order, **kw)
e1.record()
torch.cuda.synchronize()
buf = np.zeros((n_waves, 4), np.uint64)
assert lib.gpnerf_debug_read_wavetimes(buf.ctypes.data, n_waves) == 0
visits = (buf[:, 2] >> np.uint64(48)).astype(np.int64)
buf[:, 2] &= np.uint64((1 << 48) - 1)
w = buf.astype(np.int64)
visits = visits[w[:, 0] > 0]
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
us = (w - t0) / 100.0                   # 100 MHz
q = lambda a: " ".join(f"{np.percentile(a, p):8.1f}" for p in (0, 10, 50, 90, 100))
print(f"{rays.shape[0]} rays x {S} samples, {len(w)} wavefronts, launch {e0.elapsed_time(e1) * 1e3:.0f} us (events)")
print("                                   min      p10      p50      p90      max   [us]")
print("enters the kernel               ", q(us[:, 0]))
print("weights staged                  ", q(us[:, 1]))
print("first step done                 ", q(us[:, 2]))
print("leaves (last tile done)         ", q(us[:, 3]))
print("staging duration                ", q(us[:, 1] - us[:, 0]))
print("first step duration             ", q(us[:, 2] - us[:, 1]))
launches = REPS * (((S + 15) // 16) if chain else 1)        # the visit counters add up over every launch of the process
visits = visits / launches
print('tile visits per wavefront and launch:', np.unique(np.round(visits, 2), return_counts=True))
steps = (16 if chain and S > 16 else S) * visits          # (sample steps per wavefront: a tile visit walks all S)
print("mean step (staged -> exit)      ", q((us[:, 3] - us[:, 1]) / np.maximum(1, steps)))
print(f"span: first entry -> last exit {us[:, 3].max():.1f} us; last exit - median exit {us[:, 3].max() - np.median(us[:, 3]):.1f} us")
step = (us[:, 3] - us[:, 1]) / np.maximum(1, steps)
print(f'wavefront steps in total {steps.sum()}, per SIMD step-equivalent {us[:, 3].max() * len(w) / 2 / steps.sum():.2f} us')
idx = np.arange(n_waves)[buf[:, 0] > 0]
wave_in_wg, block = idx % 8, idx // 8
print("mean step by wave of the workgroup:", " ".join(f"{step[wave_in_wg == i].mean():6.1f}" for i in range(8)))
print("mean step by XCD (block % 8):      ", " ".join(f"{step[block % 8 == i].mean():6.1f}" for i in range(8)))
per_wg = np.array([step[block == b].mean() for b in np.unique(block)])
print(f"mean step per workgroup: min {per_wg.min():.1f} p10 {np.percentile(per_wg, 10):.1f} p50 {np.median(per_wg):.1f} p90 {np.percentile(per_wg, 90):.1f} max {per_wg.max():.1f}")
within = np.array([step[block == b].max() - step[block == b].min() for b in np.unique(block)])
print(f"spread inside a workgroup (max - min step): p50 {np.median(within):.1f} max {within.max():.1f}")
pair = np.array([[step[(block == b) & (wave_in_wg == i)].mean() + step[(block == b) & (wave_in_wg == i + 4)].mean() for i in range(4)] for b in np.unique(block)[:64]])
print(f"sum of the steps of waves i and i+4 (first 64 workgroups): mean {pair.mean():.1f} std {pair.std():.1f}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", f"wave_times_{size}_{S}.npy"), us)
