"""Per-call device time of gpnerf_render_fused over many calls (one event pair each): min / median / p99 / max.  A launch whose own
wavefronts evaluate its colour list waits on device-side flags and counters; this is the check that no call is ever slow for it.
usage: step_time_spread.py [size] [samples] [calls] [survey]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fm = importlib.import_module("gp-nerf_amd.frame"); syn = importlib.import_module("gp-nerf_amd.synthetic")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 300
fill = "survey" if len(sys.argv) > 4 else "full"
dev = torch.device("cuda:0")
sc = syn.make_scene(H=size, W=size, seed=0, fill=fill, pose="identity")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
order = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], size, size)).to(dev)
first = fm.render_fused(fr, rays, S, ray_order=order)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(calls)]
same = True
for a, b in ev:
    a.record()
    out = fm.render_fused(fr, rays, S, ray_order=order)
    b.record()
    same = same and bool(torch.equal(out["rgb_map"], first["rgb_map"]))
torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in ev])
print(f"{rays.shape[0]} rays x {S}, {calls} calls: min {ms.min():.3f} median {np.median(ms):.3f} p99 {np.percentile(ms, 99):.3f} max {ms.max():.3f} ms; "
      f"every call's rgb_map the first call's bits: {same}")
