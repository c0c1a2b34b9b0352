"""Attribute the fused kernel's distance from the reference on the trained-like fixtures, on the CPU.

oracle/kernel_order.inc restates the HIP kernel's arithmetic ORDER deviation by deviation on top of the op-for-op oracle; this
tool renders every tests/golden/trained_*.npz fixture with (a) the oracle itself, (b) each deviation alone, (c) all of the
round-4 kernel's deviations together (what the folded form does, modulo v_exp_f32's last bit), (d) all but one, (e) the
reference-order form of round 5 and (f) that form with one of the old deviations put back, and prints max-abs distances from the
reference's float32 maps.  ~8 minutes on 8 cores.  usage: python tools/kernel_order_report.py [mask ...]   (extra masks to try)"""
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
    R4, REF = oracle.KO_KERNEL_R4, oracle.KO_KERNEL_REF
    bits = [1 << i for i in range(len(oracle.KO_BITS)) if R4 >> i & 1]
    # TAILS and TAILS2 are two forms of the same two small layers: adding round 4's TAILS to the reference-order form replaces its TAILS2
    plus = lambda b: (REF & ~4096 | b) if b == 256 else (REF | b)
    masks = [0] + bits + [R4] + [R4 & ~b for b in bits] + [REF] + [plus(b) for b in bits if not REF & b] + extra
    labels = {0: "oracle (op for op; pinned to the reference)", R4: "ALL = the folded fp32 form (round 4's kernel)", REF: "the reference-order form (round 5's default)"}
    for b in bits:
        labels.setdefault(b, "oracle + " + name_of(b))
        labels.setdefault(R4 & ~b, "folded form - " + name_of(b))
        if not REF & b:
            labels.setdefault(plus(b), "reference-order form + " + name_of(b))
    cases = trained_case_names()
    table = {}
    for name in cases:
        z, meta = load(name)
        sc = scene_of(meta)
        S = meta["n_samples"]
        for m in dict.fromkeys(masks):
            with oracle.kernel_order(m):
                table[(name, m)] = dist(oracle.render(sc, S, want_weights=False), z)
        table[(name, "spread")] = [float(z["spread_" + k]) for k in KEYS]
    print("max-abs distance from the reference's float32 maps (tests/golden/trained_*.npz: head x 1 / 1.5 / 2 / 3), CPU twin of the kernel")
    print(f"{'variant':52s} " + " ".join(f"{c.replace('trained_', '').replace('_s64', ''):>26s}" for c in cases))
    print(f"{'':52s} " + " ".join(f"{'rgb':>8s} {'depth':>8s} {'acc':>8s}" for _ in cases))
    rows = [("reference f32 vs its own f64 head (yardstick)", "spread")] + [(labels.get(m, name_of(m)), m) for m in dict.fromkeys(masks)]
    for label, m in rows:
        print(f"{label[:52]:52s} " + " ".join(" ".join(f"{v:8.1e}" for v in table[(c, m)]) for c in cases))


if __name__ == "__main__":
    main()
