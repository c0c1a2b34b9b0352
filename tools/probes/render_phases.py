"""Phases of one Renderer.render call on the survey frame: device time (stream events) and host time (perf_counter) per phase."""
import importlib, os, sys, time
from types import SimpleNamespace as NS
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gp-nerf_amd", "plugins")]
syn = importlib.import_module("gp-nerf_amd.synthetic"); hip_render = importlib.import_module("hip_render")
R = importlib.import_module("gp-nerf_amd.render"); F_ = importlib.import_module("gp-nerf_amd.frame")
cfg = NS(encoder=NS(file="hip_encoder", name="resnet34", out_ch=32),
         head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32] * 4)),
         dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3),
         train=NS(n_rays=1024, n_samples=64), test=NS(mesh_th=50))
dev = "cuda:0"
r = hip_render.build_render(cfg).to(dev).eval()
sc = syn.make_scene(H=512, W=512, seed=0, fill=sys.argv[1] if len(sys.argv) > 1 else "survey", pose="identity", make_volumes=False)
keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk", "mask_at_box")
b = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in keys}
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, time.perf_counter(), e))
orig_encode, orig_build, orig_fused = r.encode, r.build_frame, F_.render_fused
def encode(batch, **kw): mark("encode:start"); o = orig_encode(batch, **kw); mark("encode:end"); return o
def build(*a, **k): mark("frame:start"); o = orig_build(*a, **k); mark("frame:end"); return o
def fused(*a, **k): mark("fused:start"); o = orig_fused(*a, **k); mark("fused:end"); return o
r.encode, r.build_frame, F_.render_fused = encode, build, fused
with torch.no_grad():
    for _ in range(3): r.render(b)
    rows = []
    for _ in range(9):
        marks.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        r.render(b); torch.cuda.synchronize(); t1 = time.perf_counter()
        rows.append([(n, (t - t0) * 1e3, e0.elapsed_time(e)) for n, t, e in marks] + [("return", (t1 - t0) * 1e3, None)])
rows.sort(key=lambda r: r[-1][1])          # the call with the median wall time of nine
print(f"walls of the nine calls: {' '.join(f'{r[-1][1]:.3f}' for r in rows)} ms; shown: the median one")
for n, th, td in rows[len(rows) // 2]:
    print(f"{n:14s} host {th:7.3f} ms   device {td if td is None else round(td, 3)}")
