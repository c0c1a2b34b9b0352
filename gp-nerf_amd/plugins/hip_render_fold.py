"""Plugin shim: `render.file hip_render_fold` = hip_render with the fast fp32 form of round 4: the two coarse volume levels
folded into the sigma feature layer once per frame (gpnerf_fold_volumes) and log2(e)-scaled dense layers.  ~8 % faster than
the default reference-order form; the same <= 1e-5 at initialisation scale, 5-10 x further from the reference on trained-like
parameters."""
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
