"""How much do the two readings of spconv's treatment of vertices that SHARE a voxel differ in the volumes the renderer samples?
(oracle/producers_ref.py header item 6.)  CPU only.  Body-like vertices (synthetic.body_vertices: 93 shared voxels, 193 of 6 890
rows) and the uniform-in-the-box vertices, random-initialised SparseConvNet with non-trivial BatchNorm statistics:
  A. one representative row per voxel for every submanifold lookup, every row computes through lookups (the product, gpnerf_volume.hip
     and sparse_conv3d(subm=True));
  B. the rulebook as recalled from spconv v1.2.1's CPU path (subm_conv3d_rulebook): owners receive all rows of their neighbours,
     non-owners only their centre term."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import producers_ref as pref
syn = importlib.import_module("gp-nerf_amd.synthetic")
vol = importlib.import_module("gp-nerf_amd.volume")
torch.manual_seed(0)
net = vol.SparseConvNet(n_layers=4, in_dim=32, out_dim=[32, 32, 32, 32]).eval()
for m in net.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
for body in ("capsules", "box"):
    sc = syn.make_scene(H=16, W=16, seed=0, fill="survey", pose="identity", body=body, make_volumes=False)
    coord = torch.from_numpy(sc["coord"][0]).long()
    u, cnt = np.unique(sc["coord"][0], axis=0, return_counts=True)
    coord4 = torch.cat([torch.zeros((coord.shape[0], 1), dtype=torch.long), coord], 1)
    code = torch.randn((coord.shape[0], 32))
    with torch.no_grad():
        a = pref.dense_levels(net, code, coord4, [int(v) for v in sc["out_sh"][0]])
        b = pref.dense_levels(net, code, coord4, [int(v) for v in sc["out_sh"][0]], rulebook_duplicates=True)
    print(f"{body}: {coord.shape[0]} vertices, {int((cnt > 1).sum())} shared voxels ({int(cnt[cnt > 1].sum())} rows)")
    for l, (x, y) in enumerate(zip(a, b)):
        d = (x - y).abs()
        touched = (d.amax(1) > 1e-6)
        sites = (x.abs().amax(1) > 0)
        print(f"   level {l}: {int(sites.sum())} sites, {int(touched.sum())} differ; max-abs difference {float(d.max()):.3f} on values of max {float(x.abs().max()):.2f}, "
              f"mean |value| {float(x[x != 0].abs().mean()):.3f}")
