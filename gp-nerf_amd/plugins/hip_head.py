"""Plugin shim: `head.file hip_head` on the reference's command line (see hip_render.py)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_m = importlib.import_module("gp-nerf_amd.head")
NeRFHead = _m.NeRFHead
NeRFSigmaHead = _m.NeRFSigmaHead
NeRFRGBHead = _m.NeRFRGBHead
build_head = _m.build_head
