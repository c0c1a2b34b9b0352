"""Does the next frame's encoder run BESIDE the per-ray kernel when that kernel leaves a few CUs free?  Per-ray kernel of the
ZJU-sized survey frame on the current stream, the image encoder's HIP graph (3 x 512 x 512) on a second stream, for reserve_cus =
0 / 8 / 16 / 32: time of the kernel alone, the encoder alone, and both enqueued together (wall, host-synchronised)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fm = importlib.import_module("gp-nerf_amd.frame")
syn = importlib.import_module("gp-nerf_amd.synthetic")
E = importlib.import_module("gp-nerf_amd.encoder")
dev = torch.device("cuda:0")
sc = syn.make_scene(H=512, W=512, seed=0, fill="survey", pose="identity")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
              sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))
order = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], 512, 512, patch_w=32, patch_h=8)).to(dev)
net = E.ResUNet(encoder="resnet34", out_ch=32).to(dev).eval()
imgs = t(sc["src_imgs"][0])
side = torch.cuda.Stream(device=dev)
with torch.no_grad():
    E.forward_graphed(net, imgs)


def wall(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def enc():
    with torch.no_grad(), torch.cuda.stream(side):
        E.forward_graphed(net, imgs)


print(f"rays {rays.shape[0]}")
print(f"encoder alone: {wall(enc):.3f} ms")
for R in (0, 8, 16, 32, 64):
    k = lambda: fm.render_fused(fr, rays, 64, want=("weights", "z_vals", "rgb_in"), ray_order=order, reserve_cus=R)
    both = lambda: (enc(), k())
    both2 = lambda: (k(), enc())
    print(f"reserve {R:3d}: kernel alone {wall(k):.3f} ms   encoder then kernel enqueued {wall(both):.3f} ms   kernel then encoder {wall(both2):.3f} ms")
