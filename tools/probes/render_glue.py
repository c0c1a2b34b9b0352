"""What Renderer.render(batch) spends beyond the per-ray kernel on the bench frame with the products in the batch (bench.py's
renderer_api.products_in_batch leg: feature maps NCHW + the four dense levels NCDHW, as the reference hands them over): wall against
the kernel alone, and a cProfile of the host side (`profile`).  Under `rocprofv3 --kernel-trace` + tools/trace_frames.py: the device timeline."""
import importlib, os, sys, time
from types import SimpleNamespace as NS
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gp-nerf_amd", "plugins")]
import numpy as np, torch
import bench
argv = sys.argv[1:]
sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline"]
args = bench.parse()
fm = importlib.import_module("gp-nerf_amd.frame")
hip_render = importlib.import_module("hip_render")
dev = torch.device("cuda", 0)
wl = bench.Workload(args, 512, 64, dev, 0.0)
cfg = NS(encoder=NS(file="hip_encoder", name="resnet34", out_ch=32),
         head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32, 32, 32, 32])),
         dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3),
         train=NS(n_rays=1024, n_samples=64), test=NS(mesh_th=50))
r = hip_render.build_render(cfg).to(dev).eval()
sd = r.state_dict()
for k, v in wl.sc["head"].items():
    sd["nerfhead." + k] = torch.from_numpy(v.copy())
r.load_state_dict(sd, strict=True)
keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk", "mask_at_box")
b = {k: torch.from_numpy(np.ascontiguousarray(wl.sc[k])).to(dev) for k in keys}
b["featmaps"] = torch.from_numpy(wl.sc["featmaps"]).to(dev)
b["volumes"] = wl.vols_dev
k_ms, _ = bench.time_launches(lambda: fm.render_fused(wl.frame, wl.rays, 64, want=bench.API_OUTPUTS, ray_order=wl.patch), 10, 3)
with torch.no_grad():
    for _ in range(3):
        r.render(b)
    ts = []
    for _ in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ret = r.render(b); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"products_in_batch: wall median {np.median(ts):.3f} ms (min {np.min(ts):.3f}); per-ray kernel alone {k_ms:.3f} ms -> glue {np.median(ts) - k_ms:.3f} ms")
if "profile" in argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable()
        for _ in range(40):
            r.render(b)
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("cumtime").print_stats(45)
