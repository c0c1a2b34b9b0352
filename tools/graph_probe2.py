import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
t = importlib.import_module("test_gpu_guard")
fm = importlib.import_module("gp-nerf_amd.frame"); syn = importlib.import_module("gp-nerf_amd.synthetic")
sc = syn.make_scene(H=96, W=96, seed=11, fill="full", pose="identity", sigma_bias=1.0)
fr = t.build_frame(fm, sc)
base = t.rays_of(sc)
rays = base[torch.arange(int(os.environ.get("NRAYS", "70000")), device=base.device) % base.shape[0]].contiguous()
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
kw = {"fold": False} if os.environ.get("NOFOLD") else {}
ref = fm.render_fused(fr, rays, 48, **kw)
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    fm.render_fused(fr, rays, 48, **kw)
s.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = fm.render_fused(fr, rays, 48, **kw)
for it in range(3):
    if mode != "nozero":
        for v in out.values():
            v.zero_()
    g.replay()
    torch.cuda.synchronize()
    if mode == "eq2":
        print(it, [bool((out[k] == ref[k]).all().item()) for k in ("rgb_map", "depth_map", "acc_map", "weights", "z_vals")], flush=True)
        continue
    if mode == "eq1":
        print(it, bool(torch.equal(out["rgb_map"], ref["rgb_map"])), flush=True)
        continue
    if mode == "eqw":
        print(it, bool(torch.equal(out["weights"], ref["weights"])), flush=True)
        continue
    if mode == "eq":
        print(it, [bool(torch.equal(out[k], ref[k])) for k in ("rgb_map", "depth_map", "acc_map", "weights", "z_vals")], flush=True)
        bad = (out["rgb_map"] != ref["rgb_map"]).any(1)
        if bad.any():
            i = bad.nonzero().flatten()
            print("   bad rows", int(bad.sum()), int(i[0]), int(i[-1]), "zero rows among them", int((out["rgb_map"][i] == 0).all(1).sum()),
                  "z_vals bad rows", int((out["z_vals"] != ref["z_vals"]).any(1).sum()), flush=True)
        continue
    bad = (out["rgb_map"] != ref["rgb_map"]).any(1)
    print(mode, it, "bad rows", int(bad.sum()), (int(bad.nonzero()[0]), int(bad.nonzero()[-1])) if bad.any() else None, flush=True)
