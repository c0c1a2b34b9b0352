#!/usr/bin/env python3
"""How much would per-ray (instead of per-32-ray-tile) early termination save on the configs[2] bench frame?
Per-ray stop index = first sample after which T = 1 - cumsum(weights) < term_eps, from an unterminated render."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
fm = importlib.import_module("gp-nerf_amd.frame"); syn = importlib.import_module("gp-nerf_amd.synthetic")
dev = torch.device("cuda:0"); S, eps = 128, 1e-5
sc = syn.make_scene(H=512, W=512, seed=0, fill="full", pose="identity", sigma_bias=1.0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))
order = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], 512, 512, patch_w=32, patch_h=8)).to(dev)
w = fm.render_fused(fr, rays, S, want=("weights",), ray_order=order)["weights"].double()
T = 1.0 - torch.cumsum(w, 1)
stop = torch.where((T < eps).any(1), (T < eps).double().argmax(1) + 1, torch.full((w.shape[0],), S, device=dev)).float()
for name, perm in (("32x1 rows (patch order 32x8)", order.long()), ("4x8 patches", torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], 512, 512, patch_w=4, patch_h=8)).to(dev).long())):
    s = stop[perm].view(-1, 32)
    print(f"{name}: per-ray mean {float(stop.mean()) / S:.3f} of S, per-tile (max over 32) mean {float(s.max(1)[0].mean()) / S:.3f}")
np.save(os.path.join(ROOT, "gpurun_out", "stop_rays.npy"), stop.cpu().numpy().astype(np.int16))
