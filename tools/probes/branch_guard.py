"""Does the split form's range guard written WITH a branch (-DGPNERF_X_BRANCHGUARD, build/ab/libbranch.so) still give
run-to-run different results?  Renders the same frame N times with the guarded split form and compares bits; run once with
the product library and once with GPNERF_DEBUG=1 GPNERF_LIB_PATH=build/ab/libbranch.so."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
fm = importlib.import_module("gp-nerf_amd.frame"); syn = importlib.import_module("gp-nerf_amd.synthetic"); L = importlib.import_module("gp-nerf_amd._lib")
print("library:", L.LIB_PATH)
dev = torch.device("cuda:0")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = syn.make_scene(H=size, W=size, seed=0, fill="full", pose="identity")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
blob = fm.pack_head(sc["head"], dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]), sc["Rh"][0], sc["Th"][0],
              sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], blob)
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))
for guard in (True, False):
    outs = []
    for i in range(8):
        o = fm.render_fused(fr, rays, 64, split_f16=True, guard=guard, want=("weights",))
        outs.append({k: v.clone() for k, v in o.items() if k in ("rgb_map", "depth_map", "weights")})
    torch.cuda.synchronize()
    diffs = [max(float((outs[i][k] - outs[0][k]).abs().max()) for k in outs[0]) for i in range(1, 8)]
    nbad = [int((outs[i]["rgb_map"] != outs[0]["rgb_map"]).any(dim=1).sum()) for i in range(1, 8)]
    print(f"guard={guard}: max |run_i - run_0| = {max(diffs):.3e}; rays that differ per run: {nbad}")
