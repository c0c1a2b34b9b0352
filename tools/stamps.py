#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused kernel (build: make -C gp-nerf_amd/csrc/diag libgpnerf_hip_stamps.so).
Loads the stamped library IN PLACE of the product library for this process only.  Shares, not run times."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("gp-nerf_amd._lib")
L.LIB_PATH = os.path.join(ROOT, "gp-nerf_amd", "csrc", "diag", os.environ.get("GPNERF_DIAG_LIB", "libgpnerf_hip_stamps.so"))
fm = importlib.import_module("gp-nerf_amd.frame")
syn = importlib.import_module("gp-nerf_amd.synthetic")

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
split = len(sys.argv) > 2 and sys.argv[2] == "split"
chain = len(sys.argv) > 2 and sys.argv[2] == "chain"          # segmented early-termination form, nothing terminating (term_eps 0)
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
kw = dict(early_term=True, term_eps=0.0) if chain else {}
dev = torch.device("cuda:0")
sc = syn.make_scene(H=size, W=size, seed=0, fill="full", pose="identity")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
lib = L.lib()
lib.gpnerf_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
fm.render_fused(fr, rays, S, want=(), split_f16=split, **kw)
torch.cuda.synchronize()
lib.gpnerf_debug_read_stamps(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fm.render_fused(fr, rays, S, want=(), split_f16=split, **kw)
e1.record()
torch.cuda.synchronize()
lib.gpnerf_debug_read_stamps(buf)
names = ["0 sample+volume gather", "1 geo MFMA+ELU", "2 view gather", "3 density branch", "4 colour base/vis x3", "5 rgb_fc tail", "6 composite+stores"]
tot = sum(buf[i] for i in range(7))
waves = (rays.shape[0] + 31) // 32
print(f"stamped kernel {e0.elapsed_time(e1):.2f} ms; per wave per sample cycles (s_memtime ticks):")
for i, n in enumerate(names):
    print(f"  {n:28s} {buf[i] / waves / S:9.0f}  {100.0 * buf[i] / tot:5.1f}%")
print(f"  total {tot / waves / S:.0f}")
print(f"  deferred colour passes (per wave per sample step): regather {buf[14] / waves / S:.0f}, hand-back {buf[15] / waves / S:.0f}")
print(f"  volume-gather phase of a visit's first step {buf[10] / max(1, buf[12]):.0f}, second step {buf[11] / max(1, buf[12]):.0f}")
print(f"  per tile visit: prologue (entry -> first sample) {buf[7] / max(1, buf[12]):.0f}, whole visit {buf[8] / max(1, buf[12]):.0f}, "
      f"queue pop {buf[9] / max(1, buf[12]):.0f}; loop phases {tot / max(1, buf[12]):.0f}; visits {buf[12]}")
