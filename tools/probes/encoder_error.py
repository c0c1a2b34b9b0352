"""Encoder error budget: hip_encoder against the torch-CPU restatement in float32 (= the reference's arithmetic) and float64
(the exact value), at 64x64 and 512x512.  Diagnostic (uses oracle/)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from oracle import producers_ref as ref
syn = importlib.import_module("gp-nerf_amd.synthetic"); enc = importlib.import_module("gp-nerf_amd.encoder"); L = importlib.import_module("gp-nerf_amd._lib")
print("library:", os.path.basename(L.LIB_PATH), "stats:", os.environ.get("GPNERF_ENC_STATS", "epilogue"))
for size, seed in ((64, 31), (512, 33)):
    state = {k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(seed).items()}
    imgs = torch.from_numpy(syn.make_encoder_images(size, size, seed))
    net = enc.ResUNet(); net.load_state_dict(state); net.eval()
    with torch.no_grad():
        f32 = ref.encoder(net, imgs)
        f64 = ref.encoder(enc.ResUNet().double().eval().requires_grad_(False).load_state_dict({k: v.double() for k, v in state.items()}) or None, imgs.double()) if False else None
        n64 = enc.ResUNet(); n64.load_state_dict(state); n64 = n64.double().eval()
        f64 = ref.encoder(n64, imgs.double())
        g = net.to("cuda:0")(imgs.to("cuda:0")).cpu()
    e = lambda a, b: (float((a.double() - b.double()).abs().max()), float((a.double() - b.double()).abs().mean()))
    print(f"{size}x{size}: gpu vs cpu32 max {e(g, f32)[0]:.3e} mean {e(g, f32)[1]:.3e} | gpu vs fp64 max {e(g, f64)[0]:.3e} mean {e(g, f64)[1]:.3e} | cpu32 vs fp64 max {e(f32, f64)[0]:.3e} mean {e(f32, f64)[1]:.3e}")
