"""Can the encoder's ~110 launches be captured in a HIP graph (torch.cuda.CUDAGraph around the ctypes launches), does the replay
give the eager bits, and what does it save?  3x512x512."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
syn = importlib.import_module("gp-nerf_amd.synthetic"); enc = importlib.import_module("gp-nerf_amd.encoder")
dev = torch.device("cuda:0")
net = enc.ResUNet(); net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(3).items()}); net = net.to(dev).eval()
imgs = torch.from_numpy(syn.make_encoder_images(512, 512, 3)).to(dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    ref = net(imgs).clone()
    print("eager: device %.3f ms/call, wall %.3f ms/call" % timeit(lambda: net(imgs)))
    static_in = imgs.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): net(static_in)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = net(static_in)
    g.replay(); torch.cuda.synchronize()
    print("graph replay == eager bits:", bool(torch.equal(static_out, ref)))
    static_in.copy_(imgs.flip(0)); g.replay(); torch.cuda.synchronize()
    print("second input through the same graph == eager:", bool(torch.equal(static_out, net(imgs.flip(0)))))
    static_in.copy_(imgs)
    print("graph: device %.3f ms/call, wall %.3f ms/call" % timeit(lambda: g.replay()))
