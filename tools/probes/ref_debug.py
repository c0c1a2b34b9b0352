"""debug: the reference-order form's per-sample raw output against the oracle's on one golden case"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_cases import load, scene_of
from oracle import oracle
fm = importlib.import_module("gp-nerf_amd.frame")
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "base_s8"
z, meta = load(name); sc = scene_of(meta); S = meta["n_samples"]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(oracle.rays_of(sc))
ref = oracle.render(sc, S, neg_ray=meta["neg_ray"], stages=True)
for tag, kw in (("ref-order", dict(fold=False)), ("folded", dict(fold=True))):
    o = fm.render_fused(fr, rays, S, neg_ray=meta["neg_ray"], want=("raw", "weights"), **kw)
    raw = o["raw"].cpu().numpy()
    d = np.abs(raw - ref["st_raw"])
    print(tag, "raw max-abs", d.max(axis=(0, 1)), "nan", np.isnan(raw).sum(), "rgb_map", float(np.abs(o["rgb_map"].cpu().numpy() - ref["rgb_map"]).max()))
    print("  hip ", raw[0, :3].ravel())
    print("  orac", ref["st_raw"][0, :3].ravel())
