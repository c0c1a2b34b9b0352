"""Device time of single 3x3 stride-1 convolutions of the encoder's three stages.  The calls are captured in a HIP graph (20 per
graph) and replayed, so that the time is the device's and not the ~16 us the host needs per Python launch."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
enc = importlib.import_module("gp-nerf_amd.encoder"); L = importlib.import_module("gp-nerf_amd._lib")
dev = "cuda:0"


def t(fn, per_graph=20, replays=10):
    with torch.no_grad():
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(per_graph): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (per_graph * replays) * 1e3


out = [os.path.basename(L.LIB_PATH)]
for c, hw in ((64, 128), (128, 64), (256, 32)):
    conv = torch.nn.Conv2d(c, c, 3, padding=1, bias=False, padding_mode="reflect").to(dev)
    norm = torch.nn.InstanceNorm2d(c, affine=True).to(dev)
    x = torch.randn((3, c, hw, hw), device=dev).contiguous(memory_format=torch.channels_last)
    small = x[:, :4, :2, :2].contiguous(memory_format=torch.channels_last)
    out.append(f"{c}ch@{hw}: conv {t(lambda: enc._conv(conv, x)):.1f} +sums {t(lambda: enc._conv(conv, x, stats=True)):.1f} +finalize {t(lambda: enc._conv_norm(conv, norm, x)):.1f} "
               f"(empty launch {t(lambda: enc._upsample2x(small)):.1f}) us")
print(" | ".join(out))

# fixed cost vs cost per 16-channel block at the third stage's geometry (3 x 32 x 32 pixels, 256 output channels)
line = ["cin sweep @32, cout 256 (conv only):"]
for cin in (32, 64, 128, 256, 512):
    conv = torch.nn.Conv2d(cin, 256, 3, padding=1, bias=False, padding_mode="reflect").to(dev)
    x = torch.randn((3, cin, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
    line.append(f"{cin}: {t(lambda: enc._conv(conv, x)):.1f}")
print(" ".join(line) + " us")
