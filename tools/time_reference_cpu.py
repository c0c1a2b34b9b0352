#!/usr/bin/env python3
"""Wall time of the REFERENCE's own CPU formulation on the bench frame (BASELINE.json configs[1]: 512x512 rays x 64 samples):
libs/renders/BaseRender.py Renderer.render -> batchify_rays (:160-184, test chunk 2000, configs/default.py:65) -> render_rays
(:110-157), imported from /root/reference exactly as tests/golden/make_golden.py imports it (import-time stubs for
spconv / cv2 / mcubes / trimesh; the 4 dense levels and the feature maps are inputs, as in every fixture).

Build container only: /root/reference does not exist on the GPU box, so this number cannot be taken on the host bench.py runs on --
it is the regenerable counterpart of BASELINE.md §2's figure, to be read beside `cpu_baseline` (the blocked C twin on the GPU box's
host).  Prints one JSON line.   usage: python tools/time_reference_cpu.py [--size 512] [--samples 64] [--chunk 2000] [--threads N] [--rays N]"""
import argparse
import importlib
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=2000)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--rays", type=int, default=None, help="time only the first N rays of the frame (default: all of them)")
    ap.add_argument("--repeats", type=int, default=1)
    a = ap.parse_args()
    import make_golden as mg
    mg._install_stubs()
    mg._paths()
    torch.set_num_threads(a.threads)
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    scene = syn.make_scene(H=a.size, W=a.size, seed=0, fill="full", pose="identity")        # bench.py's frame
    if a.rays:
        for k in ("ray_o", "ray_d", "near", "far", "body_msk"):
            scene[k] = scene[k][:, :a.rays]
    r, _, _ = mg.build_reference_renderer(scene, a.samples, False)
    r.chunk = a.chunk
    batch = mg.to_batch(scene)
    n = int(batch["ray_o"].shape[1])
    with torch.no_grad():
        r.render({k: (v[:, :min(n, 2 * a.chunk)] if k in ("ray_o", "ray_d", "near", "far") else v) for k, v in batch.items()})   # warm-up
        ts = []
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            ret = r.render(batch)
            ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    print(json.dumps({"what": "libs/renders/BaseRender.py Renderer.render (the reference itself, torch CPU fp32)", "rays": n, "samples": a.samples,
                      "chunk": a.chunk, "threads": a.threads, "cpu": cpu_model(), "torch": torch.__version__, "seconds": dt,
                      "rays_per_sec": n / dt, "ms_per_512x512_frame": 262144 / (n / dt) * 1e3,
                      "rgb_mean": float(ret["rgb_map"].mean())}))


if __name__ == "__main__":
    main()
