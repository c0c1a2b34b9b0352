"""Device time of the encoder's HIP-graph replay for 1, 2 and 3 views of 512x512 (what the owner of a view runs under a shard
group: Renderer.encode -> parallel.encode_views_sharded(encode_fn=forward_graphed)), and of the eager launch chain through Python
for comparison (host-bound).  Events around 30 calls each."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
syn = importlib.import_module("gp-nerf_amd.synthetic"); enc = importlib.import_module("gp-nerf_amd.encoder")
dev = torch.device("cuda:0")
net = enc.ResUNet(); net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(33).items()}); net = net.eval().to(dev)
imgs = torch.from_numpy(syn.make_encoder_images(512, 512, 33)).to(dev)
with torch.no_grad():
    for v in (1, 2, 3):
        x = imgs[:v].contiguous()
        for name, fn in (("graph replay", lambda: enc.forward_graphed(net, x)), ("eager launches", lambda: net(x))):
            for _ in range(5): fn()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            for _ in range(30): fn()
            e1.record(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 30 * 1e3
            print(f"{v} view(s) of 512x512, {name:14s}: {e0.elapsed_time(e1) / 30:.3f} ms per call on the stream ({wall:.3f} ms wall)")
