"""Plugin shim: `encoder.file hip_encoder` on the reference's command line (see hip_render.py)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_m = importlib.import_module("gp-nerf_amd.encoder")
ResUNet = _m.ResUNet
build_encoder = _m.build_encoder
