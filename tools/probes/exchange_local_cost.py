"""The single-GPU part of the strong-scaling exchange (parallel.gather_frame without the collective): pack the share's maps
(torch.cat), copy into the gathered buffer's slot, un-permute the gathered buffer into ray order (index_select) and split the
columns -- device time per frame, for the `pixels` payload (16 B/ray) and the full dict, at 512x512 and 1024x1024 with an
8-rank plan.  Replaces DESIGN.md section 6's assumed ~0.1 ms.   usage: python tools/probes/exchange_local_cost.py"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
par = importlib.import_module("gp-nerf_amd.parallel")
dev = torch.device("cuda:0")


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for size in (512, 1024):
    n, S = size * size, 64
    for world in (2, 4, 8):
        plan = par.plan_for(n, world, dev)
        for label, keys, shapes in (("pixels (rgb+depth, 16 B/ray)", par.PIXEL_KEYS, {"rgb_map": 3, "depth_map": 1}),
                                    ("all maps (rgb,depth,acc,disp,weights,z_vals,rgb_in: 588 B/ray)",
                                     ("rgb_map", "depth_map", "acc_map", "disp_map", "weights", "z_vals", "rgb_in_map"),
                                     {"rgb_map": 3, "depth_map": 1, "acc_map": 1, "disp_map": 1, "weights": S, "z_vals": S, "rgb_in_map": 9})):
            local = {k: (torch.randn((plan.share, c), device=dev) if c > 1 else torch.randn((plan.share,), device=dev)) for k, c in shapes.items()}
            C = sum(shapes.values())
            buf = torch.empty((world * plan.share, C), device=dev)

            def step():
                packed, cols = par.pack_maps(local, keys)
                buf[:plan.share].copy_(packed)                    # stands in for this rank's slot of the all-gather (same bytes written)
                full = plan.unpermute(buf)
                return {k: (full[:, a:b] if nd > 1 else full[:, a]) for k, (a, b, nd, dt) in cols.items()}

            ms = timed(step)
            print(f"{size}x{size} world {world} {label}: pack + own slot + un-permute = {ms:.3f} ms  (gathered buffer {buf.numel() * 4 / 1e6:.1f} MB)")
