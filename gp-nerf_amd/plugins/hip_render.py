"""Plugin shim: `render.file hip_render` on the reference's command line.

The reference resolves plugins with importlib.import_module(cfg.render.file) from sys.path
(tools/inference.py:61, tools/_init_paths.py:29-37); put this directory on PYTHONPATH and pass
`render.file hip_render head.file hip_head` -- no edit to tools/ is needed.
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_m = importlib.import_module("gp-nerf_amd.render")
Renderer = _m.Renderer
build_render = _m.build_render
