#!/usr/bin/env python3
"""Parity sweep of the per-frame producers on random shapes and seeds (the HIP kernels against the float64 run of the oracle's
torch-operator restatements, which the golden vectors pin to the reference): image encoder at random sizes, sparse volume
pyramid on random vertex clouds, vertex attention.  usage: producers_sweep.py [n_cases]   (writes one line per case + a summary)"""
import copy, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from oracle import producers_ref as ref
enc = importlib.import_module("gp-nerf_amd.encoder"); vol = importlib.import_module("gp-nerf_amd.volume"); syn = importlib.import_module("gp-nerf_amd.synthetic")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = "cuda:0"
rng = np.random.default_rng(2026)
worst = {"encoder": 0.0, "pyramid": 0.0, "attention": 0.0}

for i in range(n_cases):
    # ---- encoder: random size (multiples of 4 .. odd), random weights
    H, W = int(rng.integers(40, 300)), int(rng.integers(40, 300))
    seed = int(rng.integers(1 << 30))
    net = enc.ResUNet().eval()
    state = syn.make_encoder_weights(seed % 1000)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    imgs = torch.from_numpy(syn.make_encoder_images(H, W, seed % 997))
    with torch.no_grad():
        want = ref.encoder(copy.deepcopy(net).double(), imgs.double()).float()
        got = net.to(dev)(imgs.to(dev)).cpu()
    e = float((got - want).abs().max())
    worst["encoder"] = max(worst["encoder"], e)
    line = f"case {i:3d}: encoder {H:3d}x{W:3d} max-abs {e:.2e}"

    # ---- pyramid: random vertex cloud in a random box, random BatchNorm statistics
    torch.manual_seed(seed)
    in_dim = int(rng.choice([16, 32]))
    sp = vol.SparseConvNet(n_layers=4, in_dim=in_dim, out_dim=[32, 32, 32, 32]).to(dev).eval()
    for m in sp.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    out_sh = [int(16 * rng.integers(1, 5)), int(16 * rng.integers(1, 7)), int(16 * rng.integers(1, 5))]
    nv = int(rng.integers(50, 4000))
    coord = torch.from_numpy(np.stack([rng.integers(0, out_sh[0], nv), rng.integers(0, out_sh[1], nv), rng.integers(0, out_sh[2], nv)], 1)).to(dev)
    coord4 = torch.cat([torch.zeros((nv, 1), dtype=coord.dtype, device=dev), coord], 1)
    code = torch.randn((nv, in_dim), device=dev)
    with torch.no_grad():
        hip = sp.dense_levels_hip(code, coord4, out_sh)
        refl = ref.dense_levels(copy.deepcopy(sp).cpu().double(), code.cpu().double(), coord4.cpu(), out_sh)
    pe = 0.0
    for a, b in zip(hip, refl):
        pe = max(pe, float((a.permute(3, 0, 1, 2).cpu().double() - b[0]).abs().max()) / max(1.0, float(b.abs().max())))
    worst["pyramid"] = max(worst["pyramid"], pe)
    line += f" | pyramid {nv:4d} vertices in {out_sh}, in_dim {in_dim}: rel {pe:.2e}"

    # ---- attention
    nh = int(rng.choice([1, 2, 4, 8]))
    att = vol.MultiHeadAttention(nh, 32, 32 // nh, 32 // nh, kv_dim=32, sum=False).to(dev).eval()
    if True:
        n = int(rng.integers(1, 7000)); V = 3
        c, f = torch.randn((n, 32), device=dev), torch.randn((n, V, 32), device=dev)
        with torch.no_grad():
            got = att.fuse_vertices(c, f).cpu().double()
            want = ref.attention(copy.deepcopy(att).cpu().double(), c.cpu().double().unsqueeze(1), f.cpu().double(), f.cpu().double())
            want = (want[0] if isinstance(want, tuple) else want).reshape(n, 32)
        ae = float((got - want).abs().max()) / max(1.0, float(want.abs().max()))
        worst["attention"] = max(worst["attention"], ae)
        line += f" | attention {n:4d} x {att.n_head} heads: rel {ae:.2e}"
    print(line, flush=True)
print("worst over", n_cases, "cases:", ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
assert worst["encoder"] < 1e-4 and worst["pyramid"] < 1e-4 and worst["attention"] < 1e-4
