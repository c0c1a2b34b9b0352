"""Plugin shim: `render.file hip_demo_render` = the inference renderer of the README command
(libs/renders/demo_render.py: progressive ray selection + sample culling, returns `pred_img`), on the HIP path."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_m = importlib.import_module("gp-nerf_amd.render")
Renderer = _m.Renderer


def build_render(cfg):
    return _m.build_render(cfg, progressive=True)
