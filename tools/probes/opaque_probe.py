"""Diagnostic: the exact-opacity exit on frames whose every ray is opaque at once (density bias + 60): step statistics by frame size
and launch shape."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
fm = importlib.import_module("gp-nerf_amd.frame"); syn = importlib.import_module("gp-nerf_amd.synthetic")
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for size in (96, 272, 512):      # 272^2 = 74 k rays: one round of wavefronts + a remainder (the chained form's single launch)
    sc = syn.make_scene(H=size, W=size, seed=94, fill="full", pose="random", aabb_half=(0.2, 0.3, 0.12), bias_std=0.1, sigma_bias=60.0)
    fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                  sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
    rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
    for lb in (True, False):
        a = fm.render_fused(fr, rays, 64, want=("weights", "step_stats", "samples_done") if not lb else ("weights", "step_stats"), load_balance=lb)
        w = a["weights"]
        print(size, "load_balance", lb, "stats", a["step_stats"].cpu().numpy(), "nonzero weights per ray", float((w != 0).sum(1).float().mean()), "acc", float(a["acc_map"].mean()))
