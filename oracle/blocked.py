"""ctypes binding of gpnerf_cpu_blocked.c, the THROUGHPUT twin of the oracle (bench.py `cpu_baseline`, kind "port-blocked").

Test / bench infrastructure only (see the C file's header): the product never imports this.  The library is compiled with
-march=native, so it is built on the host it runs on, under a name that carries a hash of that host's CPU flags (a build
made in the container is not loaded on the GPU box's host)."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

from . import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "gpnerf_cpu_blocked.c")
FP = C.POINTER(C.c_float)
NV, NC, NL = 3, 32, 4


class BlockedFrame(C.Structure):
    _fields_ = ([("imgs", FP), ("featmaps", FP), ("vol", FP * NL), ("vol_dhw", (C.c_int32 * 3) * NL),
                 ("img_h", C.c_int32), ("img_w", C.c_int32), ("feat_h", C.c_int32), ("feat_w", C.c_int32),
                 ("K4P4", (C.c_float * 16) * NV), ("Rh", C.c_float * 9), ("Th", C.c_float * 3), ("bounds_min", C.c_float * 3),
                 ("voxel", C.c_float * 3), ("out_sh", C.c_int32 * 3)]
                + [(n + s, FP) for n in ("geo", "b1", "b2", "v1", "v2", "r1", "r2", "r3") for s in ("_w", "_b")]
                + [(n + s, FP) for n in ("d1", "d2", "d3", "d4") for s in ("_w", "_b")])


class BlockedOut(C.Structure):
    _fields_ = [("rgb", FP), ("depth", FP), ("acc", FP), ("disp", FP), ("weights", FP), ("rgb_in", FP), ("ray_mask", C.POINTER(C.c_uint8))]


def _host_tag():
    flags = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = line
                break
    except OSError:
        pass
    return hashlib.sha256((flags + open(SRC).read()).encode()).hexdigest()[:10]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, f"libgpnerf_cpu_blocked.{_host_tag()}.so")
        if not os.path.exists(path):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-std=gnu11", "-Wall",
                                   "-o", path, SRC, "-lm"])
        _lib = C.CDLL(path)
        _lib.blocked_render.restype = C.c_int
        _lib.blocked_render.argtypes = [C.POINTER(BlockedFrame), FP, C.c_int64, C.c_int, C.c_int, C.POINTER(BlockedOut), C.c_int]
        _lib.blocked_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return int(lib().blocked_max_threads())


_f32, _p = O._f32, O._p


class Frame:
    """Per-frame preparation: the scene's tensors in channels-last order (outside any timed call)."""

    def __init__(self, scene):
        f = BlockedFrame()
        imgs = _f32(scene["src_imgs"][0] * 0.5 + 0.5)                       # BaseRender.py:231
        V, _, H, W = imgs.shape
        imgs4 = np.zeros((V, H, W, 4), np.float32)
        imgs4[..., :3] = imgs.transpose(0, 2, 3, 1)
        fm = _f32(np.asarray(scene["featmaps"]).transpose(0, 2, 3, 1))
        self.keep = [imgs4, fm]
        f.imgs, f.featmaps = _p(imgs4), _p(fm)
        f.img_h, f.img_w, f.feat_h, f.feat_w = H, W, fm.shape[1], fm.shape[2]
        for l, v in enumerate(scene["volumes"]):
            v = _f32(v[0].transpose(1, 2, 3, 0))
            self.keep.append(v)
            f.vol[l] = _p(v)
            for a in range(3):
                f.vol_dhw[l][a] = v.shape[a]
        M = O.k4p4(scene["src_Ks"][0], scene["src_poses"][0])
        for v in range(NV):
            for i in range(16):
                f.K4P4[v][i] = float(M[v].ravel()[i])
        Rh, Th, bm = _f32(scene["Rh"][0]).ravel(), _f32(scene["Th"][0]).ravel(), _f32(scene["bounds"][0][0]).ravel()
        for i in range(9):
            f.Rh[i] = float(Rh[i])
        for i in range(3):
            f.Th[i], f.bounds_min[i] = float(Th[i]), float(bm[i])
            f.voxel[i] = float(np.float32(scene["voxel_size"][i]))
            f.out_sh[i] = int(scene["out_sh"][0][i])
        for short, name in O._W.items():
            w, b = _f32(scene["head"][name + ".weight"]), _f32(scene["head"][name + ".bias"])
            self.keep += [w, b]
            setattr(f, short + "_w", _p(w))
            setattr(f, short + "_b", _p(b))
        self.c = f


def render(frame, rays, n_samples, neg_ray=False, want=("rgb_in", "weights"), n_threads=0):
    """Every ray of `rays` [N,8] through the blocked CPU path; returns the same maps as oracle.render."""
    rays = _f32(rays)
    N, S = rays.shape[0], int(n_samples)
    res = {"rgb_map": np.zeros((N, 3), np.float32), "depth_map": np.zeros(N, np.float32), "acc_map": np.zeros(N, np.float32),
           "disp_map": np.zeros(N, np.float32), "ray_mask": np.zeros(N, np.uint8)}
    o = BlockedOut()
    o.rgb, o.depth, o.acc, o.disp = _p(res["rgb_map"]), _p(res["depth_map"]), _p(res["acc_map"]), _p(res["disp_map"])
    o.ray_mask = res["ray_mask"].ctypes.data_as(C.POINTER(C.c_uint8))
    if "rgb_in" in want:
        res["rgb_in_map"] = np.zeros((N, 9), np.float32)
        o.rgb_in = _p(res["rgb_in_map"])
    if "weights" in want:
        res["weights"] = np.zeros((N, S), np.float32)
        o.weights = _p(res["weights"])
    rc = lib().blocked_render(C.byref(frame.c), _p(rays), N, S, 3 if neg_ray else 0, C.byref(o), int(n_threads))
    assert rc == 0, rc
    return res
