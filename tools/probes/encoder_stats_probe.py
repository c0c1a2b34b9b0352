"""Which part of the fp32 encoder's distance from the reference's float32 arithmetic is the InstanceNorm statistics?  The fused chain
as shipped (tables from the convolutions' tile sums) against the same chain with every table recomputed by the double-precision
pass (gpnerf_instance_norm_act_nhwc's statistics kernels), on the 3 x 512 x 512 reference vector's weights and images."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
E = importlib.import_module("gp-nerf_amd.encoder")
L = E.L
syn = importlib.import_module("gp-nerf_amd.synthetic")
from oracle import producers_ref as ref
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
net = E.ResUNet(encoder="resnet34", out_ch=32).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(11).items()}, strict=True)
imgs = torch.from_numpy(syn.make_encoder_images(size, size, 11))
with torch.no_grad():
    want = ref.encoder(importlib.import_module("copy").deepcopy(net).double(), imgs.double()).numpy()
    want32 = ref.encoder(net, imgs).numpy()
net = net.to(dev)
x = imgs.to(dev)
orig_cn, orig_cat = E._conv_norm, E._conv_norm_cat


def table_by_pass(norm, y):
    lib = L.lib()
    n, c, h, w = y.shape
    nbytes = int(lib.gpnerf_instance_norm_nhwc_scratch_bytes(n, h * w, c))
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=y.device)
    dummy = torch.empty_like(y, memory_format=torch.channels_last)
    L.check(lib.gpnerf_instance_norm_act_nhwc(y.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), None, n, h * w, c, float(norm.eps), 0,
                                              dummy.data_ptr(), scratch.data_ptr(), E._st(y)), "norm")
    return scratch[nbytes - n * 3 * c * 4:].view(torch.float32).view(n, 3, c).clone()


def cn(conv, norm, xx, in_tab=None, in_act=0):
    y, tab = orig_cn(conv, norm, xx, in_tab=in_tab, in_act=in_act)
    t2 = table_by_pass(norm, y)
    cn.diff = max(getattr(cn, "diff", 0.0), float(((t2[:, 1] - tab[:, 1]).abs() / t2[:, 1].abs().clamp_min(1e-30)).max()))
    return y, t2


def cat(conv, norm, xa, xb):
    y, tab = orig_cat(conv, norm, xa, xb)
    return y, table_by_pass(norm, y)


def report(tag, o):
    o = o.cpu().numpy()
    print(f"{tag}: vs float64 max {np.abs(o - want).max():.2e} mean {np.abs(o - want).mean():.2e}; vs torch-CPU float32 max {np.abs(o - want32).max():.2e} mean {np.abs(o - want32).mean():.2e}")


with torch.no_grad():
    for prec in ("fp32", "split"):
        net.precision = prec
        E._conv_norm, E._conv_norm_cat = orig_cn, orig_cat
        report(f"{prec}, tables from the convolutions' tile sums", net(x))
        E._conv_norm, E._conv_norm_cat = cn, cat
        report(f"{prec}, tables from a double-precision pass   ", net(x))
        print("   largest relative difference of a scale (gamma * rstd) between the two tables:", cn.diff)
        cn.diff = 0.0
