"""Where the HOST spends its time inside one Renderer.render call on the survey frame (perf_counter around the call's pieces while
the device runs beside it): medians over 20 calls."""
import importlib, os, sys, time
from types import SimpleNamespace as NS
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gp-nerf_amd", "plugins")]
syn = importlib.import_module("gp-nerf_amd.synthetic"); F_ = importlib.import_module("gp-nerf_amd.frame")
V_ = importlib.import_module("gp-nerf_amd.volume"); hip_render = importlib.import_module("hip_render")
S = 64
cfg = NS(encoder=NS(file="hip_encoder", name="resnet34", out_ch=32),
         head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32] * 4)),
         dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3),
         train=NS(n_rays=1024, n_samples=S), test=NS(mesh_th=50))
dev = "cuda:0"
r = hip_render.build_render(cfg).to(dev).eval()
sc = syn.make_scene(H=512, W=512, seed=0, fill="survey", pose="identity", make_volumes=False)
keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk", "mask_at_box")
b = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in keys}
acc = {}
def timed(obj, name, label=None):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); o = f(*a, **k); acc.setdefault(label or name, []).append((time.perf_counter() - t) * 1e3); return o
    setattr(obj, name, w)
R_ = importlib.import_module("gp-nerf_amd.render"); timed(R_, "_record_on")
timed(r, "encode"); timed(r, "build_frame"); timed(r, "prepare_builder_inputs"); timed(F_, "render_fused"); timed(F_, "patch_order_device")
timed(F_.Frame, "consts_of_batch"); timed(F_, "relayout_images"); timed(F_, "project_gather"); timed(F_.Frame, "from_batch", "Frame()")
net = r.nerfhead.sigmahead.xyzc_net
timed(net, "plan_levels"); timed(net, "dense_levels_hip"); timed(r.nerfhead.sigmahead, "build_volumes")
walls = []
with torch.no_grad():
    for _ in range(3): r.render(b)
    for k in acc: acc[k].clear()
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r.render(b); torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) * 1e3)
print(f"wall median {np.median(walls):.3f} ms")
for k, v in acc.items():
    print(f"  {k:26s} host median {np.median(v):7.3f} ms  (x{len(v) // 20} per call)")

if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable()
        for _ in range(50): r.render(b)
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
