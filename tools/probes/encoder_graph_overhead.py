"""What the encoder's HIP-graph replay costs beyond its kernels: events around copy-in / replay / clone-out, 30 calls."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
syn = importlib.import_module("gp-nerf_amd.synthetic"); enc = importlib.import_module("gp-nerf_amd.encoder")
dev = torch.device("cuda:0")
net = enc.ResUNet(); net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(3).items()}); net = net.to(dev).eval()
imgs = torch.from_numpy(syn.make_encoder_images(512, 512, 3)).to(dev)
with torch.no_grad():
    enc.forward_graphed(net, imgs)
    g = net.__dict__["_gpnerf_graph_f32" if net.precision == "fp32" else "_gpnerf_graph"][1]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    acc = [0.0, 0.0, 0.0, 0.0]
    n = 30
    for _ in range(n):
        torch.cuda.synchronize()
        e = [ev() for _ in range(5)]
        e[0].record(); g.static_in.copy_(imgs); e[1].record(); g.graph.replay(); e[2].record(); out = g.static_out.clone(); e[3].record()
        torch.cuda.synchronize()
        acc[0] += e[0].elapsed_time(e[1]); acc[1] += e[1].elapsed_time(e[2]); acc[2] += e[2].elapsed_time(e[3]); acc[3] += e[0].elapsed_time(e[3])
    print("isolated call (device idle before it): copy-in %.3f  replay %.3f  clone %.3f  total %.3f ms" % tuple(a / n for a in acc))
    torch.cuda.synchronize(); a, b = ev(), ev(); a.record()
    for _ in range(n): enc.forward_graphed(net, imgs)
    b.record(); torch.cuda.synchronize()
    print("back to back: %.3f ms per call" % (a.elapsed_time(b) / n))
    a, b = ev(), ev(); torch.cuda.synchronize(); a.record()
    for _ in range(n): net(imgs)
    b.record(); torch.cuda.synchronize()
    print("eager back to back: %.3f ms per call" % (a.elapsed_time(b) / n))
