import importlib, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
fm = importlib.import_module("gp-nerf_amd.frame"); syn = importlib.import_module("gp-nerf_amd.synthetic")
dev = torch.device("cuda:0")
def build(sc):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                  sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
    rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))
    return fr, rays
for size, S in ((64, 32), (512, 64)):
    sc = syn.make_scene(H=size, W=size, seed=3, fill="full", pose="identity")
    fr, rays = build(sc)
    a = fm.render_fused(fr, rays, S)
    b = fm.render_fused(fr, rays, S, split_f16=True, guard=False)
    c = fm.render_fused(fr, rays, S, split_f16=True, want=("weights", "z_vals", "rgb_in", "guard_tiles"))
    print(size, "plain scene: guard tiles", int(c["guard_tiles"]), "split vs f32", float((a["rgb_map"]-b["rgb_map"]).abs().max()), "guarded vs unguarded bit-equal", bool(torch.equal(b["rgb_map"], c["rgb_map"])))
    # blow up the features a part of the image sees: operands beyond the f16 range there
    sc2 = dict(sc); f2 = sc["featmaps"].copy(); f2[:, :, : f2.shape[2] // 3] *= 3.0e5; sc2["featmaps"] = f2
    fr2, _ = build(sc2)
    a = fm.render_fused(fr2, rays, S)
    b = fm.render_fused(fr2, rays, S, split_f16=True, guard=False)
    c = fm.render_fused(fr2, rays, S, split_f16=True, want=("weights", "z_vals", "rgb_in", "guard_tiles"))
    torch.cuda.synchronize()
    nt = (rays.shape[0] + 31) // 32
    print(size, "overflow scene: guard tiles", int(c["guard_tiles"]), "of", nt, "| unguarded split vs f32", float((a["rgb_map"]-b["rgb_map"]).abs().max()),
          "| guarded vs f32", float((a["rgb_map"]-c["rgb_map"]).abs().max()), float((a["depth_map"]-c["depth_map"]).abs().max()), float((a["weights"]-c["weights"]).abs().max()))
import time
sc = syn.make_scene(H=512, W=512, seed=0, fill="full", pose="identity"); fr, rays = build(sc)
for kw in (dict(), dict(split_f16=True, guard=False), dict(split_f16=True, guard=True)):
    for _ in range(3): fm.render_fused(fr, rays, 64, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fm.render_fused(fr, rays, 64, **kw)
    torch.cuda.synchronize(); print(kw, round((time.perf_counter() - t0) * 100, 3), "ms")
