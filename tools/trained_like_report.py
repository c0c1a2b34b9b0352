"""How far the oracle and the HIP path sit from the reference on the "trained-like" fixtures (tests/golden/trained_*.npz: head
weights x 1 / 1.5 / 2 / 3 with biases, features x 4 with log-normal tails, ReLU-sparse levels), next to the reference's OWN
float32 rounding noise on the same inputs (`spread_*`: its float32 run against its head evaluated in float64, make_golden.py
_Head64).  Prints one table; with a GPU also the fused kernel's forms.   usage: python tools/trained_like_report.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_cases import load, scene_of, trained_case_names  # noqa: E402
from oracle import oracle  # noqa: E402

KEYS = ("rgb_map", "depth_map", "acc_map")


def errs(res, z):
    return {k: (float(np.abs(np.asarray(res[k], np.float64) - z[k]).max()), float(np.abs(np.asarray(res[k], np.float64) - z[k + "_head64"]).max()))
            for k in KEYS}


def main():
    oracle.build()
    import torch
    gpu = torch.cuda.is_available()
    if gpu:
        fm = importlib.import_module("gp-nerf_amd.frame")
        dev = torch.device("cuda:0")
    print(f"{'case':18s} {'path':30s} " + " ".join(f"{k + ' vs ref32 / ref-head64':>34s}" for k in KEYS))
    for name in trained_case_names():
        z, meta = load(name)
        sc = scene_of(meta)
        S = meta["n_samples"]
        rows = [("reference f32 vs its f64 head", {k: (float(z["spread_" + k]), 0.0) for k in KEYS}),
                ("C oracle", errs(oracle.render(sc, S), z))]
        twins = {}
        for tag, mask in (("CPU twin of the ref-order form", oracle.KO_KERNEL_REF), ("CPU twin of the folded form", oracle.KO_KERNEL_R4)):
            with oracle.kernel_order(mask):          # oracle/kernel_order.inc: the kernel's arithmetic order restated on the CPU
                twins[tag] = oracle.render(sc, S, want_weights=False)
            rows.append((tag, errs(twins[tag], z)))
        if gpu:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                          sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
            rays = t(oracle.rays_of(sc))
            for tag, kw, twin in (("HIP fp32 reference order", dict(fold=False), "CPU twin of the ref-order form"),
                                  ("HIP fp32 folded", dict(fold=True), "CPU twin of the folded form"), ("HIP split-f16 + guard", dict(split_f16=True), None)):
                o = {k: v.cpu().numpy() for k, v in fm.render_fused(fr, rays, S, **kw).items() if k in KEYS}
                rows.append((tag, errs(o, z)))
                if twin:       # how well the CPU twin predicts the kernel: max-abs between the two (everything but v_exp_f32's last bit is modelled)
                    rows.append(("   ... vs its CPU twin", {k: (float(np.abs(o[k].astype(np.float64) - twins[twin][k]).max()), 0.0) for k in KEYS}))
        for tag, e in rows:
            print(f"{name:18s} {tag:30s} " + " ".join(f"{e[k][0]:16.2e} / {e[k][1]:<15.2e}" for k in KEYS))


if __name__ == "__main__":
    main()
