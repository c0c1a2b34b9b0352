"""Target of tools/probes/conv_pmc.sh: the 3x3 kernel on 3 x 32 x 32 pixels, 512 -> 256 channels (32 channel blocks: the loop
dominates), 20 eager launches."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
enc = importlib.import_module("gp-nerf_amd.encoder")
dev = "cuda:0"
cin = int(os.environ.get("CIN", "512"))
conv = torch.nn.Conv2d(cin, 256, 3, padding=1, bias=False, padding_mode="reflect").to(dev)
x = torch.randn((3, cin, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(20): enc._conv(conv, x)
torch.cuda.synchronize()
