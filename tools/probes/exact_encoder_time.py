"""ResUNet.forward_exact on 3 x 512 x 512: the tiled fp32-MFMA convolutions (round 5) against the untiled form (GPNERF_DEBUG=1
GPNERF_EXACT_UNTILED=1), the split-f16 graph for scale, and the distance of each from the float64 torch-CPU restatement."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
E = importlib.import_module("gp-nerf_amd.encoder")
syn = importlib.import_module("gp-nerf_amd.synthetic")
from oracle import producers_ref as ref
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = E.ResUNet(encoder="resnet34", out_ch=32).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(11).items()}, strict=True)
imgs = torch.from_numpy(syn.make_encoder_images(size, size, 11))
with torch.no_grad():
    want = ref.encoder(importlib.import_module("copy").deepcopy(net).double(), imgs.double()).numpy()
net = net.to(dev)
x = imgs.to(dev)


def timed(fn, n=5):
    with torch.no_grad():
        out = fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms, out = timed(lambda: net.forward_exact(x))
print(f"forward_exact ({'untiled' if os.environ.get('GPNERF_EXACT_UNTILED') == '1' else 'tiled'}) 3x{size}x{size}: {ms:.2f} ms; max-abs vs float64 {np.abs(out.cpu().numpy() - want).max():.2e} "
      f"(output range {np.abs(want).max():.2f})")
ms2, out2 = timed(lambda: E.forward_graphed(net, x), 20)
print(f"split-f16 graph: {ms2:.2f} ms; max-abs vs float64 {np.abs(out2.cpu().numpy() - want).max():.2e}")
