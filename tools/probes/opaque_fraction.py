"""CPU: how many samples of the reference-made trained-like fixtures lie behind a transmittance of exactly 0 (what the exact-opacity
exit of the sample loop can skip), computed from the oracle's staged densities in the compositing's own float32 order."""
import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from golden_cases import load, scene_of
from oracle import oracle
for name in ("trained_h1_s64","trained_h1p5_s64","trained_h2_s64","trained_h3_s64","demo_body_s64"):
    try:
        z, meta = load(name)
    except Exception as e:
        print(name, "ERR", e); continue
    sc = scene_of(meta); S = meta["n_samples"]
    r = oracle.render(sc, S, neg_ray=meta.get("neg_ray", False), stages=True)
    sig = r["st_raw"][..., 3].astype(np.float32)
    if meta.get("neg_ray", False): sig = sig[:, ::-1]
    alpha = (np.float32(1) - np.exp(-sig)).astype(np.float32)
    T = np.ones(sig.shape[0], np.float32); behind = 0; dead_rays = 0
    for k in range(S):
        behind += int((T == 0).sum())
        T = (T * ((np.float32(1) - alpha[:, k]) + np.float32(1e-10))).astype(np.float32)
    print(f"{name}: rays {sig.shape[0]}, samples behind an exactly-zero transmittance {behind / sig.size:.3f}, rays ending at T == 0: {(T == 0).mean():.3f}, max sigma {sig.max():.1f}, acc mean {r['acc_map'].mean():.3f}")
