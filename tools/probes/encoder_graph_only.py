"""30 HIP-graph replays of the 3x512x512 encoder and nothing else (the target of tools/gpu_prof_encoder.sh's kernel trace: an eager
call's timeline shows the HOST's launch latency between kernels, a replay's shows the device's)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
syn = importlib.import_module("gp-nerf_amd.synthetic"); enc = importlib.import_module("gp-nerf_amd.encoder")
dev = torch.device("cuda:0")
net = enc.ResUNet(); net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(33).items()}); net = net.eval().to(dev)
x = torch.from_numpy(syn.make_encoder_images(512, 512, 33)).to(dev)
with torch.no_grad():
    for _ in range(5): enc.forward_graphed(net, x)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): enc.forward_graphed(net, x)
    e1.record(); torch.cuda.synchronize()
print(f"encoder 3x512x512, graph replay: {e0.elapsed_time(e1) / 30:.3f} ms per call")
