"""Encoder accuracy against the reference vector and its time at 512x512, under whatever MIOpen algorithm switches the
environment sets (diagnostic)."""
import importlib, sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
from golden_cases import load
syn = importlib.import_module("gp-nerf_amd.synthetic"); enc = importlib.import_module("gp-nerf_amd.encoder")
z, meta = load("encoder_72x88")
net = enc.ResUNet("resnet34", 32); net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(meta["seed"]).items()}, strict=True); net = net.eval().to("cuda:0")
imgs = torch.from_numpy(syn.make_encoder_images(meta["H"], meta["W"], meta["seed"])).to("cuda:0")
with torch.no_grad():
    out = net(imgs)
print("env", os.environ.get("MIOPEN_DEBUG_CONV_WINOGRAD"), os.environ.get("MIOPEN_DEBUG_CONV_FFT"), "max err", float(np.abs(out.cpu().numpy() - z["featmaps"]).max()))
big = torch.rand((3, 3, 512, 512), device="cuda:0") * 2 - 1
with torch.no_grad():
    for _ in range(3): net(big)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net(big)
    torch.cuda.synchronize(); print("512x512 encoder ms", (time.perf_counter() - t0) * 100)
