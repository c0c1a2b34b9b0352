"""Plugin shim: `render.file hip_render_fold` = hip_render with round 4's fp32 form: the two coarse volume levels folded into the
sigma feature layer once per frame (gpnerf_fold_volumes) and log2(e)-scaled dense layers.  Layer for layer ~8 % faster than the
default reference-order form, but since both defer the colour branch they render the bench frame in the
same time; the same <= 1e-5 at initialisation scale, 5-10 x further from the reference on trained-like parameters (section 5).
Kept for comparison with round 4."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_m = importlib.import_module("gp-nerf_amd.render")
Renderer = _m.Renderer


def build_render(cfg):
    r = _m.build_render(cfg)
    r.fold_levels = True
    return r
