"""debug (GPNERF_DEBUG=1 GPNERF_LIB_PATH=.../libgpnerf_hip_dump.so): intermediate registers of the reference-order form"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_cases import load, scene_of
from oracle import oracle
fm = importlib.import_module("gp-nerf_amd.frame")
dev = torch.device("cuda:0")
z, meta = load("base_s8"); sc = scene_of(meta); S = meta["n_samples"]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(oracle.rays_of(sc))
ref = oracle.render(sc, S, neg_ray=meta["neg_ray"], stages=True)
o = fm.render_fused(fr, rays, S, neg_ray=meta["neg_ray"], want=("raw",), fold=False)
raw = o["raw"].cpu().numpy()
W, b = sc["head"]["sigmahead.out_geometry_fc.0.weight"], sc["head"]["sigmahead.out_geometry_fc.0.bias"]
vf = ref["st_vol_feat"]                       # [N,S,128]
pre = vf @ W.T + b
sf = np.where(pre > 0, pre, np.expm1(pre))
for r in (0, 5, 100):
    for k in (0, 3):
        v = vf[r, k]
        print(f"ray {r} k {k}: dump {raw[r, k]}  expect l0c0 {v[0]:.6f} l0c2 {v[2]:.6f} l0c16 {v[16]:.6f} l1c0 {v[32]:.6f}   l0 first 6: {np.round(v[:6], 5)}")
