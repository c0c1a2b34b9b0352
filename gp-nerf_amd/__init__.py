"""MI355X-native per-ray render path of GP-NeRF (libs/renders + libs/nerfheads of sail-sg/GP-Nerf).

The directory name carries a hyphen; import it with ``importlib.import_module("gp-nerf_amd")``.
Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every piece of
arithmetic on the path runs in hand-written gfx950 kernels behind the C ABI of include/gpnerf_hip.h.
"""
from . import _lib  # noqa: F401
from ._lib import GpnerfError  # noqa: F401

__all__ = ["GpnerfError", "_lib"]
