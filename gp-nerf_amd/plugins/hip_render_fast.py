"""Plugin shim: `render.file hip_render_fast` = hip_render with the split-precision dense layers
(GPNERF_FLAG_SPLIT_F16: f16 hi/lo operand pairs on the f16 MFMA, f32 accumulation; same 1e-4 parity bound)."""
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
    r.split_f16 = True
    return r
