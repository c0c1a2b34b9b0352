"""Device time of the encoder at 3x512x512 (events, 30 calls) + error against the float32 / float64 torch-CPU restatement."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from oracle import producers_ref as ref
syn = importlib.import_module("gp-nerf_amd.synthetic"); enc = importlib.import_module("gp-nerf_amd.encoder")
dev = torch.device("cuda:0")
state = {k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(33).items()}
net = enc.ResUNet(); net.load_state_dict(state); net.eval()
imgs = torch.from_numpy(syn.make_encoder_images(512, 512, 33))
with torch.no_grad():
    f32 = ref.encoder(net, imgs)
    n64 = enc.ResUNet(); n64.load_state_dict(state); f64 = ref.encoder(n64.double().eval(), imgs.double())
    gnet, gi = net.to(dev), imgs.to(dev)
    out = gnet(gi)
    for _ in range(5): gnet(gi)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): gnet(gi)
    e1.record(); torch.cuda.synchronize()
    again = gnet(gi)
g = out.cpu().double()
print(f"encoder 3x512x512: {e0.elapsed_time(e1) / 30:.3f} ms/call; deterministic {bool(torch.equal(out, again))}; vs cpu32 max {float((g - f32.double()).abs().max()):.3e} "
      f"mean {float((g - f32.double()).abs().mean()):.3e}; vs fp64 mean {float((g - f64).abs().mean()):.3e} (cpu32 vs fp64 mean {float((f32.double() - f64).abs().mean()):.3e})")
