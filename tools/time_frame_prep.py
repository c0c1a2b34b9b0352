#!/usr/bin/env python3
"""Per-frame producers timed on the GPU: sparse volume builder (gp-nerf_amd/volume.py), re-layout, occupancy, image encoder."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
head = importlib.import_module("gp-nerf_amd.head")
fm = importlib.import_module("gp-nerf_amd.frame")
syn = importlib.import_module("gp-nerf_amd.synthetic")
dev = torch.device("cuda:0")
sc = syn.make_scene(H=512, W=512, seed=0, fill="full", pose="identity", make_volumes=False)
h = head.NeRFHead(code_dim=32).to(dev).eval()
coord = torch.from_numpy(sc["coord"][0]).to(dev)
coord4 = torch.cat([torch.zeros((coord.shape[0], 1), dtype=coord.dtype, device=dev), coord], 1)
sp = {"coord": coord4, "out_sh": [int(x) for x in sc["out_sh"][0]], "batch_size": 1}
feat = torch.randn((1, 6890, 3, 32), device=dev)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


with torch.no_grad():
    ms, vols = timed(lambda: h.sigmahead.build_volumes(sp, feat))
print(f"volume builder (HIP sparse conv): {ms:.1f} ms; levels {[tuple(v.shape) for v in vols]}; active level-1 voxels {(vols[0].abs().sum(-1) > 0).sum().item()}")
blob = fm.pack_head({k: v for k, v in h.per_ray_state().items()}, dev)
ms, fr = timed(lambda: fm.Frame.for_volumes(vols, blob))
print(f"re-layout of the pyramid: {ms:.2f} ms")
ms, _ = timed(lambda: fr.build_occupancy())
print(f"occupancy volume: {ms:.2f} ms")

# image encoder (gp-nerf_amd/encoder.py): 3 source views 512x512 -> [3,32,128,128], channels-last out
enc = importlib.import_module("gp-nerf_amd.encoder")
net = enc.ResUNet().to(dev).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(0).items()}, strict=True)
imgs = torch.from_numpy(syn.make_encoder_images(512, 512, 0)).to(dev)
with torch.no_grad():
    for _ in range(3):
        net(imgs)                      # MIOpen picks its algorithms on the first calls
    ms, out = timed(lambda: net(imgs), n=10)
print(f"encoder (ResUNet, MIOpen channels-last): {ms:.2f} ms per frame; out {tuple(out.shape)} strides {out.stride()}")
