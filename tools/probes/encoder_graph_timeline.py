"""One replay of the fp32 encoder graph under `rocprofv3 --kernel-trace`: run this, then parse the trace with `parse <csv>`:
every launch of the last replay with its duration, grid and workgroup size -- where the 1.9 ms sit and which launches leave CUs idle."""
import csv, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "parse":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    # the last replay: from its stem kernel on
    stems = [i for i, r in enumerate(rows) if "conv7x7_s2_stem" in r["Kernel_Name"]]
    a = stems[-1]
    t0 = int(rows[a]["Start_Timestamp"]); prev = t0; tot = 0
    for r in rows[a:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gx, gy, gz = (int(r.get(f"Grid_Size_{a}", 1)) for a in "XYZ")
        wx, wy, wz = (int(r.get(f"Workgroup_Size_{a}", 1)) for a in "XYZ")
        wgs = (gx // max(1, wx)) * (gy // max(1, wy)) * (gz // max(1, wz))
        tot += e - s
        print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:5.1f}  workgroups {wgs:6d} x {wx * wy * wz:4d}  {r['Kernel_Name'].replace('(anonymous namespace)::', '')[:70]}")
        prev = e
    print(f"span {(prev - t0) / 1e3:.1f} us, kernels {tot / 1e3:.1f} us")
    sys.exit(0)
import numpy as np, torch
E = importlib.import_module("gp-nerf_amd.encoder")
syn = importlib.import_module("gp-nerf_amd.synthetic")
dev = torch.device("cuda:0")
net = E.ResUNet(encoder="resnet34", out_ch=32).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(11).items()}, strict=True)
net = net.to(dev)
x = torch.from_numpy(syn.make_encoder_images(512, 512, 11)).to(dev)
with torch.no_grad():
    for _ in range(6):
        out = E.forward_graphed(net, x)
torch.cuda.synchronize()
print("ok", tuple(out.shape))
