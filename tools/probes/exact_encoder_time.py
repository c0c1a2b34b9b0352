"""The image encoder's two arithmetic forms on 3 x 512 x 512 (ResUNet.precision): "fp32" (default: fp32 operands on the fp32 MFMA, the
fused launch chain as one HIP graph) against "split" (f16 hi/lo operands), and the distance of each from the float64 torch-CPU
restatement; the fp32 form against the independent per-operand restatement gpnerf_conv2d_nhwc_exact on one layer."""
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
    want32 = ref.encoder(net, imgs).numpy()
net = net.to(dev)
x = imgs.to(dev)


def timed(fn, n=20):
    with torch.no_grad():
        out = fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


print(f"reference arithmetic (torch CPU float32) vs float64: max {np.abs(want32 - want).max():.2e} mean {np.abs(want32 - want).mean():.2e}")
for prec in ("fp32", "split"):
    net.precision = prec
    ms_e, out_e = timed(lambda: net(x), 5)
    ms, out = timed(lambda: E.forward_graphed(net, x))
    o = out.cpu().numpy()
    print(f"precision {prec}: graph {ms:.3f} ms, eager {ms_e:.2f} ms; vs float64 max {np.abs(o - want).max():.2e} mean {np.abs(o - want).mean():.2e}; "
          f"vs the reference's float32 arithmetic max {np.abs(o - want32).max():.2e} mean {np.abs(o - want32).mean():.2e}; graph == eager {bool(torch.equal(out, out_e))}")
