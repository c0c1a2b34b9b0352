import importlib, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
fm = importlib.import_module("gp-nerf_amd.frame"); syn = importlib.import_module("gp-nerf_amd.synthetic")
dev = torch.device("cuda:0")
for size, S, neg in ((64, 32, False), (96, 48, True), (272, 24, False)):
    sc = syn.make_scene(H=size, W=size, seed=3, fill="full", pose="random", aabb_half=(0.2, 0.3, 0.12), bias_std=0.1, neg_cams=neg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                  sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
    rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1))
    for kw in ({}, {"early_term": True, "term_eps": 1e-5}, {"occ_cull": True}):
        a = fm.render_fused(fr, rays, S, neg_ray=neg, want=("weights", "raw"), fold=False, **kw)
        b = fm.render_fused(fr, rays, S, neg_ray=neg, want=("weights", "raw"), fold=True, **kw)
        print(size, S, neg, kw, {k: float((a[k].float() - b[k].float()).abs().nan_to_num().max()) for k in ("rgb_map", "depth_map", "acc_map", "weights", "raw")})
