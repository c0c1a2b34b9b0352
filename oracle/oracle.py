"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE -- see gpnerf_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgpnerf_oracle.so")
NV, NC, NL = 3, 32, 4
FP = C.POINTER(C.c_float)


class OracleFrame(C.Structure):
    _fields_ = [
        ("imgs", FP), ("featmaps", FP), ("vol", FP * NL), ("vol_dhw", (C.c_int32 * 3) * NL),
        ("img_h", C.c_int32), ("img_w", C.c_int32), ("feat_h", C.c_int32), ("feat_w", C.c_int32),
        ("K4P4", (C.c_float * 16) * NV), ("Rh", C.c_float * 9), ("Th", C.c_float * 3),
        ("bounds_min", C.c_float * 3), ("voxel", C.c_float * 3), ("out_sh", C.c_int32 * 3),
        ("geo_w", FP), ("geo_b", FP),
        ("b1_w", FP), ("b1_b", FP), ("b2_w", FP), ("b2_b", FP),
        ("v1_w", FP), ("v1_b", FP), ("v2_w", FP), ("v2_b", FP),
        ("r1_w", FP), ("r1_b", FP), ("r2_w", FP), ("r2_b", FP), ("r3_w", FP), ("r3_b", FP),
        ("d1_w", FP), ("d1_b", FP), ("d2_w", FP), ("d2_b", FP), ("d3_w", FP), ("d3_b", FP), ("d4_w", FP), ("d4_b", FP),
        ("occ", FP),
    ]


class OracleOut(C.Structure):
    _fields_ = [
        ("rgb", FP), ("depth", FP), ("acc", FP), ("disp", FP), ("weights", FP), ("z_vals", FP), ("rgb_in", FP),
        ("ray_mask", C.POINTER(C.c_uint8)),
        ("st_grid", FP), ("st_vol_feat", FP), ("st_rgb_feat", FP), ("st_mask", FP), ("st_raw", FP),
        ("samples_done", C.POINTER(C.c_int32)), ("term_eps", C.c_float),
    ]


_W = {
    "geo": "sigmahead.out_geometry_fc.0", "b1": "rgbhead.base_fc.0", "b2": "rgbhead.base_fc.2",
    "v1": "rgbhead.vis_fc.0", "v2": "rgbhead.vis_fc.2", "r1": "rgbhead.rgb_fc.0", "r2": "rgbhead.rgb_fc.2",
    "r3": "rgbhead.rgb_fc.4", "d1": "rgbhead.out_geometry_fc.0", "d2": "rgbhead.out_geometry_fc.2",
    "d3": "rgbhead.out_geometry_fc.4", "d4": "rgbhead.out_geometry_fc.6",
}


def build(force=False):
    src = os.path.join(HERE, "gpnerf_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB


_lib = None
_main_lib = None
KO_LIB = os.path.join(HERE, "libgpnerf_kernel_order.so")
KO_BITS = ("DIV3", "BIASFIRST", "KORDER", "SCALED", "EXP2ELU", "TRIFMA", "FOLD", "V1SCALE", "TAILS", "EXPS", "ELU1ULP", "DIV3FMA", "TAILS2")
KO_KERNEL_R4 = 1 | 2 | 4 | 8 | 32 | 64 | 128 | 256 | 512      # the folded fp32 form's order (round 4's kernel; kernel_order.inc)
KO_KERNEL_REF = 16 | 512 | 2048 | 4096                        # the reference-order form (round 5): only exp2-based exps, x/3 by FMA, split tails


class kernel_order:
    """Diagnostic (tools/kernel_order_report.py): inside the `with`, render() / head_forward() go through the oracle's twin
    built with kernel_order.inc, the HIP kernel's deviations from the reference's arithmetic order switched on by `mask`
    (bit i = KO_BITS[i]; 0 = the oracle itself).  Not a checker."""

    def __init__(self, mask):
        self.mask = int(mask)

    def __enter__(self):
        global _lib, _main_lib
        src = [os.path.join(HERE, n) for n in ("gpnerf_oracle.c", "kernel_order.inc")]
        if not os.path.exists(KO_LIB) or os.path.getmtime(KO_LIB) < max(os.path.getmtime(s) for s in src):
            subprocess.check_call(["make", "-C", HERE, "-B", "kernel_order"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _main_lib = lib()
        _lib = _bind(C.CDLL(KO_LIB))
        _lib.kernel_order_set.argtypes = [C.c_int]
        _lib.kernel_order_set(self.mask)
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = _main_lib
        return False


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(LIB))
    return _lib


def _bind(_lib):
    if True:
        _lib.oracle_render.restype = C.c_int
        _lib.oracle_render.argtypes = [C.POINTER(OracleFrame), FP, C.c_int64, C.c_int, C.c_int, C.POINTER(OracleOut), C.c_int]
        _lib.oracle_make_rays.restype = C.c_int64
        DP = C.POINTER(C.c_double)
        _lib.oracle_make_rays.argtypes = [C.c_int, C.c_int, DP, DP, DP, FP, FP, FP, FP, FP, C.POINTER(C.c_uint8)]
        _lib.oracle_composite.restype = C.c_int
        _lib.oracle_composite.argtypes = [FP, FP, FP, C.c_int64, C.c_int, C.c_int, FP, FP, FP, FP, FP, C.POINTER(C.c_uint8)]
        _lib.oracle_max_threads.restype = C.c_int
        _lib.oracle_select_rays.restype = C.c_int64
        _lib.oracle_select_rays.argtypes = [FP, C.c_int, C.c_int, C.c_int, C.c_float, FP, FP, FP, FP, FP, FP, FP, C.c_int, C.c_int,
                                            C.c_int, FP, FP, FP, FP, C.POINTER(C.c_uint8)]
        _lib.oracle_build_occupancy.restype = C.c_int
        _lib.oracle_build_occupancy.argtypes = [C.POINTER(OracleFrame), FP]
        _lib.oracle_head_forward.restype = C.c_int
        _lib.oracle_head_forward.argtypes = [C.POINTER(OracleFrame), FP, FP, FP, C.c_int64, FP]
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(FP) if a is not None else FP()


def k4p4(src_Ks, src_poses):
    """train_intrinsics.bmm(train_poses) on the 4x4 embeddings, fp32 (BaseRender.py:233-247,314).  For matrices this small
    torch.bmm runs its native loop: every element a float32 multiply-add chain over k = 0..3, NO fused multiply-adds (checked
    bit for bit on 6 000 elements; numpy's `@` goes through a BLAS whose FMAs differ in the last bit of most products)."""
    V = src_Ks.shape[0]
    out = np.zeros((V, 4, 4), np.float32)
    for v in range(V):
        K4 = np.eye(4, dtype=np.float32)
        K4[:3, :3] = src_Ks[v].astype(np.float32)
        P4 = np.eye(4, dtype=np.float32)
        P4[:3, :4] = src_poses[v].astype(np.float32)
        acc = np.zeros((4, 4), np.float32)
        for k in range(4):
            acc = (acc + (K4[:, k:k + 1] * P4[k:k + 1, :]).astype(np.float32)).astype(np.float32)
        out[v] = acc
    return out


def head_only_frame(head):
    """OracleFrame carrying only the MLP parameters (for head_forward)."""
    fr = Frame.__new__(Frame)
    fr.keep = []
    f = OracleFrame()
    for short, name in _W.items():
        w = _f32(head[name + ".weight"])
        b = _f32(head[name + ".bias"])
        fr.keep += [w, b]
        setattr(f, short + "_w", _p(w))
        setattr(f, short + "_b", _p(b))
    fr.c = f
    return fr


def head_forward(head, vol_feat, rgb_feat, mask):
    """NeRFHead.forward on gathered features: [P,128], [P,V,35], [P,V] -> raw [P,4]."""
    fr = head_only_frame(head)
    vol_feat, rgb_feat, mask = _f32(vol_feat), _f32(rgb_feat), _f32(mask)
    P = vol_feat.shape[0]
    raw = np.zeros((P, 4), np.float32)
    assert lib().oracle_head_forward(C.byref(fr.c), _p(vol_feat), _p(rgb_feat), _p(mask), P, _p(raw)) == 0
    return raw


class Frame:
    """Holds numpy arrays alive and exposes the C struct."""

    def __init__(self, scene):
        self.keep = []
        f = OracleFrame()
        imgs = _f32(scene["src_imgs"][0] * 0.5 + 0.5)  # BaseRender.py:231
        fm = _f32(scene["featmaps"])
        self.keep += [imgs, fm]
        f.imgs, f.featmaps = _p(imgs), _p(fm)
        f.img_h, f.img_w = imgs.shape[-2:]
        f.feat_h, f.feat_w = fm.shape[-2:]
        for l, v in enumerate(scene["volumes"]):
            v = _f32(v[0])
            self.keep.append(v)
            f.vol[l] = _p(v)
            for a in range(3):
                f.vol_dhw[l][a] = v.shape[1 + a]
        M = k4p4(scene["src_Ks"][0], scene["src_poses"][0])
        for v in range(NV):
            for i in range(16):
                f.K4P4[v][i] = float(M[v].ravel()[i])
        Rh = _f32(scene["Rh"][0]).ravel()
        Th = _f32(scene["Th"][0]).ravel()
        bm = _f32(scene["bounds"][0][0]).ravel()
        for i in range(9):
            f.Rh[i] = float(Rh[i])
        for i in range(3):
            f.Th[i] = float(Th[i])
            f.bounds_min[i] = float(bm[i])
            f.voxel[i] = float(np.float32(scene["voxel_size"][i]))
            f.out_sh[i] = int(scene["out_sh"][0][i])
        for short, name in _W.items():
            w = _f32(scene["head"][name + ".weight"])
            b = _f32(scene["head"][name + ".bias"])
            self.keep += [w, b]
            setattr(f, short + "_w", _p(w))
            setattr(f, short + "_b", _p(b))
        self.c = f


def rays_of(scene):
    return _f32(np.concatenate([scene["ray_o"][0], scene["ray_d"][0], scene["near"][0][:, None], scene["far"][0][:, None]], 1))


def build_occupancy(scene):
    """masks3d of SparseConvNet.encode (SparseConvNet.py:135-139) from the scene's dense levels."""
    fr = Frame(scene)
    D, H, W = scene["volumes"][0].shape[-3:]
    occ = np.zeros((D, H, W), np.float32)
    assert lib().oracle_build_occupancy(C.byref(fr.c), _p(occ)) == 0
    return occ


def render(scene, n_samples, neg_ray=False, stages=False, n_threads=0, rays=None, want_weights=True, occ=None, flip=None, term_eps=0.0):
    """Run the oracle over all rays of a synthetic scene; returns a dict of numpy arrays.
    neg_ray: the Projector's front test (h_z < 0).  flip: raw2outputs(neg=True); default = neg_ray for the dense renderer
    and False for the progressive one, which never flips (demo_render.py:329-344).
    occ: masks3d -> the progressive renderer's per-sample rules (pinned by tests/golden/demo_*.npz).
    term_eps > 0: the build's per-ray early termination (not in the reference): a ray stops at the first sample at whose start its
    transmittance is below term_eps; adds `samples_done`."""
    if flip is None:
        flip = bool(neg_ray) and occ is None
    fr = Frame(scene)
    if occ is not None:
        occ = _f32(occ)
        fr.keep.append(occ)
        fr.c.occ = _p(occ)
    rays = rays_of(scene) if rays is None else _f32(rays)
    N, S = rays.shape[0], int(n_samples)
    res = {
        "rgb_map": np.zeros((N, 3), np.float32), "depth_map": np.zeros(N, np.float32),
        "acc_map": np.zeros(N, np.float32), "disp_map": np.zeros(N, np.float32),
        "rgb_in_map": np.zeros((N, 9), np.float32), "ray_mask": np.zeros(N, np.uint8),
    }
    if want_weights:
        res["weights"] = np.zeros((N, S), np.float32)
        res["z_vals"] = np.zeros((N, S), np.float32)
    if stages:
        res.update({
            "st_grid": np.zeros((N, S, 3), np.float32), "st_vol_feat": np.zeros((N, S, 128), np.float32),
            "st_rgb_feat": np.zeros((N, S, NV, NC + 3), np.float32), "st_mask": np.zeros((N, S, NV), np.float32),
            "st_raw": np.zeros((N, S, 4), np.float32),
        })
    o = OracleOut()
    o.rgb, o.depth, o.acc, o.disp = _p(res["rgb_map"]), _p(res["depth_map"]), _p(res["acc_map"]), _p(res["disp_map"])
    o.rgb_in = _p(res["rgb_in_map"])
    o.ray_mask = res["ray_mask"].ctypes.data_as(C.POINTER(C.c_uint8))
    if want_weights:
        o.weights, o.z_vals = _p(res["weights"]), _p(res["z_vals"])
    if term_eps > 0:
        res["samples_done"] = np.zeros(N, np.int32)
        o.samples_done = res["samples_done"].ctypes.data_as(C.POINTER(C.c_int32))
        o.term_eps = float(term_eps)
    if stages:
        for k in ("st_grid", "st_vol_feat", "st_rgb_feat", "st_mask", "st_raw"):
            setattr(o, k, _p(res[k]))
    rc = lib().oracle_render(C.byref(fr.c), _p(rays), N, S, int(bool(neg_ray)) | (2 if flip else 0) | (4 if occ is not None else 0), C.byref(o), int(n_threads))
    assert rc == 0
    return res


def camera_inverses(K, R, T):
    """What get_rays derives from the camera (data_utils.py:49-51,57), in the dtype it is handed (the dataset: float64),
    returned as float64 for the C side: inv(K), inv(R), -inv(R) @ T."""
    K, R, T = np.asarray(K), np.asarray(R), np.asarray(T).reshape(3, 1)
    R_inv = np.linalg.inv(R)
    o = (-R_inv @ T).ravel()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    return f64(np.linalg.inv(K)), f64(R_inv), f64(o)


def make_rays(H, W, K, R, T, bounds):
    Kinv, Rinv, o = camera_inverses(K, R, T)
    bounds = _f32(bounds)
    n = H * W
    ro, rd = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    near, far = np.zeros(n, np.float32), np.zeros(n, np.float32)
    mask = np.zeros(n, np.uint8)
    DP = C.POINTER(C.c_double)
    k = lib().oracle_make_rays(H, W, Kinv.ctypes.data_as(DP), Rinv.ctypes.data_as(DP), o.ctypes.data_as(DP), _p(bounds), _p(ro), _p(rd),
                               _p(near), _p(far), mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    return ro[:k], rd[:k], near[:k], far[:k], mask.astype(bool)


def composite(raw, z, nvalid, neg=False):
    raw, z = _f32(raw), _f32(z)
    N, S = z.shape
    nv = _f32(nvalid) if nvalid is not None else None
    rgb, depth, acc, disp = np.zeros((N, 3), np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32)
    w = np.zeros((N, S), np.float32)
    rm = np.zeros(N, np.uint8)
    lib().oracle_composite(_p(raw), _p(z), _p(nv), N, S, int(neg), _p(rgb), _p(depth), _p(acc), _p(disp), _p(w),
                           rm.ctypes.data_as(C.POINTER(C.c_uint8)))
    return {"rgb_map": rgb, "depth_map": depth, "acc_map": acc, "disp_map": disp, "weights": w, "ray_mask": rm.astype(bool)}


def max_threads():
    return int(lib().oracle_max_threads())


def select_rays(occ, voxel, bmin, Rh, Th, pose, K, ih, iw, neg_ray=False, thr=0.1, Kinv=None):
    """demo_render.py:166-247: occupied voxels -> pixel set -> rays / near / far (pinned by tests/golden/demo_*.npz)."""
    occ, voxel, bmin, Rh, Th, pose, K = (_f32(a) for a in (occ, voxel, bmin, Rh, Th, pose, K))
    # batch['target_K_inv'] = np.linalg.inv(target_K) on the float32 K (ZjumocapDataset.py:480)
    Kinv = _f32(np.linalg.inv(K)) if Kinv is None else _f32(Kinv)
    n = ih * iw
    ro, rd, near, far = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    mask = np.zeros(n, np.uint8)
    D, H, W = occ.shape
    k = lib().oracle_select_rays(_p(occ), D, H, W, float(thr), _p(voxel.ravel()), _p(bmin.ravel()), _p(Rh.ravel()), _p(Th.ravel()),
                                 _p(pose.ravel()), _p(K.ravel()), _p(Kinv.ravel()), ih, iw, int(bool(neg_ray)), _p(ro), _p(rd),
                                 _p(near), _p(far), mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    return ro[:k], rd[:k], near[:k], far[:k], mask.astype(bool)
