#!/usr/bin/env python3
"""How far is tile-level sample culling from per-sample culling on the bench's sparse frame (--occ-cull --occupancy 0.1)?
keep[r, k] = trilinear occupancy > 0 at sample k of ray r (the renderer's rule, demo_render.py:270-283), evaluated here with
torch's grid_sample (diagnostic).  A 4x8-pixel wavefront tile costs, per frame of S steps:
  union   the steps at which ANY of its 32 rays keeps its sample          (what the kernel does)
  cursor  max over its rays of the ray's kept samples                      (every lane walks its own ray's kept samples)
  sorted  the same after sorting the rays by their kept count              (re-packing: ~ sum of kept samples / 32)"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
fm = importlib.import_module("gp-nerf_amd.frame"); syn = importlib.import_module("gp-nerf_amd.synthetic")
dev = torch.device("cuda:0"); S = 64
occf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
sc = syn.make_scene(H=512, W=512, seed=0, fill="full", pose="identity", vol_occupancy=occf)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
occ = fr.build_occupancy()
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))
_, _, grid = fm.sample_points(fr, rays, S)                                   # [N,S,3] normalised xyz (the dense renderer's voxel size)
v = torch.nn.functional.grid_sample(occ[None, None], grid.view(1, 1, 1, -1, 3), mode="bilinear", padding_mode="zeros", align_corners=True)
keep = (v.view(-1, S) > 0)
order = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], 512, 512, patch_w=4, patch_h=8)).to(dev).long()
k = keep[order].view(-1, 32, S)
cnt = keep.sum(1).float()
union = k.any(1).sum(1).float().mean() / S
cursor = k.sum(2).max(1)[0].float().mean() / S
srt = torch.sort(cnt)[0].view(-1, 32).max(1)[0].mean() / S
print(f"occupancy {occf}: samples kept {float(keep.float().mean()):.3f}; per 4x8 tile: union {float(union):.3f}, per-lane cursor {float(cursor):.3f}, "
      f"rays sorted by count {float(srt):.3f} of S")
# per-tile step counts (union) and what the order of the tiles costs on 2048 wavefront slots
u = k.any(1).sum(1).cpu().numpy()                      # steps per 4x8 tile, in the launch's tile order
print(f"steps per tile: mean {u.mean():.1f}, max {u.max()}, 90th percentile {np.percentile(u, 90):.0f}; tiles with no step {int((u == 0).sum())} of {len(u)}")
import heapq
def makespan(lengths, workers=2048):
    h = [0.0] * workers
    heapq.heapify(h)
    for L in lengths:
        heapq.heappush(h, heapq.heappop(h) + L)
    return max(h)
print(f"makespan in steps on 2048 slots: launch order {makespan(u):.0f}, longest first {makespan(np.sort(u)[::-1]):.0f}, ideal {u.sum() / 2048:.1f}")
