"""Attribute the fused kernel's distance from the reference on the trained-like fixtures, on the CPU.

oracle/kernel_order.inc restates the HIP kernel's arithmetic ORDER deviation by deviation on top of the op-for-op oracle; this
tool renders every tests/golden/trained_*.npz fixture with (a) the oracle itself, (b) each deviation alone, (c) all of the
round-4 kernel's deviations together (what the kernel does, modulo v_exp_f32's last bit), (d) all but one, and prints max-abs
distances from the reference's float32 maps.  usage: python tools/kernel_order_report.py [mask ...]   (extra masks to try)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_cases import load, scene_of, trained_case_names  # noqa: E402
from oracle import oracle  # noqa: E402

KEYS = ("rgb_map", "depth_map", "acc_map")


def name_of(mask):
    return "+".join(n for i, n in enumerate(oracle.KO_BITS) if mask >> i & 1) or "oracle"


def dist(res, z):
    return [float(np.abs(np.asarray(res[k], np.float64) - z[k]).max()) for k in KEYS]


def main():
    extra = [int(a, 0) for a in sys.argv[1:]]
    ALL = oracle.KO_KERNEL_R4
    bits = [1 << i for i in range(len(oracle.KO_BITS)) if ALL >> i & 1]
    masks = [0] + bits + [ALL] + [ALL & ~b for b in bits] + extra
    cases = trained_case_names()
    table = {}
    for name in cases:
        z, meta = load(name)
        sc = scene_of(meta)
        S = meta["n_samples"]
        for m in masks:
            with oracle.kernel_order(m):
                table[(name, m)] = dist(oracle.render(sc, S, want_weights=False), z)
        table[(name, "spread")] = [float(z["spread_" + k]) for k in KEYS]
    print(f"{'variant':58s} " + " ".join(f"{c.replace('trained_', '').replace('_s64', ''):>26s}" for c in cases))
    print(f"{'':58s} " + " ".join(f"{'rgb':>8s} {'depth':>8s} {'acc':>8s}" for _ in cases))
    rows = [("reference f32 vs its own f64 head", "spread")] + [(("kernel r4 = " if m == ALL else "") + (("all but " + name_of(ALL & ~m)) if (m != ALL and bin(m).count("1") > 1 and m in [ALL & ~b for b in bits]) else name_of(m)), m) for m in masks]
    for label, m in rows:
        print(f"{label[:58]:58s} " + " ".join(" ".join(f"{v:8.1e}" for v in table[(c, m)]) for c in cases))


if __name__ == "__main__":
    main()
