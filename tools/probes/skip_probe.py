"""The reference-order form's bit-exact exits (empty-space sigma layer; colour branch deferred and run only for samples with a
non-zero weight) on a PERSON-SHAPED frame rendered by the dense renderer: kernel time and bits with the exits on (default) and
off (GPNERF_FLAG_NO_EXITS, a second process), the fraction of zero-density samples, the same on the dense synthetic bench frame,
and on that frame with every density positive (nothing to skip: the deferral's own cost).
usage: python tools/probes/skip_probe.py            (parent)        |   ... child <out.npz>"""
import importlib, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SCENES = {"body (capsule figure, sparse levels, sigma_bias -1.5)": dict(H=512, W=512, seed=0, fill="survey", pose="identity", body="capsules", sigma_bias=-1.5, bias_std=0.1, vol_scale=2.0),
          "bench frame (dense random levels)": dict(H=512, W=512, seed=0, fill="full", pose="identity"),
          # nothing to skip: what the queue and the regather of the deferred colour branch cost by themselves
          "bench frame, density bias +1 (zero only where no view sees the sample)": dict(H=512, W=512, seed=0, fill="full", pose="identity", sigma_bias=1.0),
          # the trained-like parameter distributions of tests/golden/trained_h1 / h3 (heads x 1 / x 3 with biases, heavy-tailed x 4
          # features, ReLU-sparse levels) at the bench frame's size: a full-size box, 512 x 512 rays
          "trained-like x 1, 512 x 512 (the distributions of trained_h1_s64)": dict(H=512, W=512, seed=46, fill="full", pose="random", bias_std=0.1, sigma_bias=-10.0,
                                                                                head_scale=1.0, feat_scale=4.0, feat_tail=0.5, vol_scale=4.0, vol_relu=True),
          "trained-like x 3, 512 x 512 (the distributions of trained_h3_s64)": dict(H=512, W=512, seed=52, fill="full", pose="random", bias_std=0.5, sigma_bias=-16.0,
                                                                                head_scale=3.0, feat_scale=4.0, feat_tail=0.5, vol_scale=4.0, vol_relu=True),
          # every ray opaque behind its first few samples: transmittance underflows to exactly 0 and the rest of the ray is skipped
          "bench frame, density bias +60 (opaque at once)": dict(H=512, W=512, seed=0, fill="full", pose="identity", sigma_bias=60.0)}


def child(path):
    import torch
    fm = importlib.import_module("gp-nerf_amd.frame")
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    dev = torch.device("cuda:0")
    out = {}
    for name, kw in SCENES.items():
        sc = syn.make_scene(**kw)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                      sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
        rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))
        pw, ph = (int(v) for v in os.environ.get("SKIP_PROBE_PATCH", "32x8").split("x"))       # pixels per wavefront: W x H
        order = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], 512, 512, patch_w=pw, patch_h=ph)).to(dev)
        want = ("weights", "z_vals", "rgb_in")
        f = lambda: fm.render_fused(fr, rays, 64, want=want, ray_order=order, exits=os.environ.get("SKIP_PROBE_EXITS", "1") == "1")
        for _ in range(3):
            o = f()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record(); o = f(); b.record()
        torch.cuda.synchronize()
        out[name + "|ms"] = np.float64(np.median([a.elapsed_time(b) for a, b in ev]))
        for k in ("rgb_map", "depth_map", "acc_map", "weights", "rgb_in_map"):
            out[name + "|" + k] = o[k].cpu().numpy()
        raw = fm.render_fused(fr, rays, 64, want=("raw",), ray_order=order)["raw"].cpu().numpy()
        out[name + "|sigma_zero_frac"] = np.float64((raw[..., 3] == 0).mean())
        out[name + "|n"] = np.int64(rays.shape[0])
    np.savez(path, **out)


if len(sys.argv) > 2 and sys.argv[1] == "child":
    child(sys.argv[2])
    sys.exit(0)
res = {}
for tag, env in (("exits on (default)", {}), ("exits off", {"SKIP_PROBE_EXITS": "0"})):
    p = f"/tmp/skip_{len(res)}.npz"
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", p], env=dict(os.environ, **env))
    res[tag] = np.load(p)
a, b = res["exits on (default)"], res["exits off"]
for name in SCENES:
    same = all(np.array_equal(a[name + "|" + k], b[name + "|" + k], equal_nan=True) for k in ("rgb_map", "depth_map", "acc_map", "weights", "rgb_in_map"))
    print(f"{name}: {int(a[name + '|n'])} rays x 64; samples with density exactly 0: {float(a[name + '|sigma_zero_frac']) * 100:.1f} %; "
          f"kernel {float(b[name + '|ms']):.3f} ms with the exits off -> {float(a[name + '|ms']):.3f} ms with them on; every map bit-identical: {same}")
