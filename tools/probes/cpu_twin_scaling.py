"""Thread scaling of the blocked CPU twin on this host (diagnostic; uses oracle/)."""
import importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from oracle import blocked, oracle
syn = importlib.import_module("gp-nerf_amd.synthetic")
print(subprocess.run("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node\\(s\\)|L3|L2'", shell=True, capture_output=True, text=True).stdout)
sc = syn.make_scene(H=512, W=512, seed=0, fill="full", pose="identity")
rays = oracle.rays_of(sc)
fr = blocked.Frame(sc)
for th in (1, 8, 32, 64, 128):
    n = min(rays.shape[0], 4096 * th)
    sub = rays[:: max(1, rays.shape[0] // n)][:n]
    blocked.render(fr, sub[:256 * th], 64, want=(), n_threads=th)
    t0 = time.perf_counter(); blocked.render(fr, sub, 64, want=(), n_threads=th); dt = time.perf_counter() - t0
    print(f"threads {th:4d}: {n} rays {dt:.3f} s  {n/dt:10.0f} rays/s  {n/dt/th:8.0f} /thread  {n/dt*64*110848/1e9:8.1f} GFLOP/s", flush=True)
