#!/usr/bin/env python3
"""bench.py over image sizes x samples per ray (both kernel forms); one table row per configuration."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print("size samples     rays | fp32 ms  Mrays/s  frac | split+guard ms  Mrays/s")
for size in (64, 128, 256, 512, 1024):
    for S in (32, 64, 128):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", str(size), "--samples", str(S),
                              "--no-cpu-baseline", "--steps", "10", "--warmup", "3"], capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        s = (d.get("beside_headline") or {}).get("split_f16_api_outputs_patch_order") or {}
        print(f"{size:4d} {S:7d} {d['config']['rays_per_gpu']:8d} | {d['ms_per_step']:7.3f} {d['value'] / 1e6:8.2f} {d['roofline']['frac']:5.3f} |"
              f" {s.get('kernel_ms', 0):14.3f} {s.get('rays_per_sec', 0) / 1e6:8.2f}", flush=True)
