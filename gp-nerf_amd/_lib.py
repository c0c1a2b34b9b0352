"""ctypes binding of include/gpnerf_hip.h.

The shared library is built in-tree (gp-nerf_amd/csrc/libgpnerf_hip.so) by
``__graft_entry__.build()`` / ``make -C gp-nerf_amd/csrc``.  There is NO fallback: if the
library is missing or fails to load, every entry point raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree build, always.  GPNERF_LIB_PATH (a differently built library for the A/B measurements of
# tools/) is honoured only together with GPNERF_DEBUG=1, the switch that also enables the launcher's experiment knobs.
_DEBUG = os.environ.get("GPNERF_DEBUG", "0") == "1"
LIB_PATH = (os.environ.get("GPNERF_LIB_PATH") if _DEBUG else None) or os.path.join(HERE, "csrc", "libgpnerf_hip.so")

VIEWS, CH, LEVELS = 3, 32, 4
FP = C.POINTER(C.c_float)
DP = C.POINTER(C.c_double)
U8P = C.POINTER(C.c_uint8)


class GpnerfFrame(C.Structure):
    _fields_ = [
        ("vol", C.c_void_p * LEVELS),
        ("vol_dhw", (C.c_int32 * 3) * LEVELS),
        ("featmaps", C.c_void_p),
        ("feat_h", C.c_int32), ("feat_w", C.c_int32),
        ("imgs", C.c_void_p),
        ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("proj", (C.c_float * 12) * VIEWS),
        ("Rh", C.c_float * 9), ("Th", C.c_float * 3),
        ("bounds_min", C.c_float * 3), ("voxel", C.c_float * 3),
        ("out_sh", C.c_int32 * 3),
        ("head_blob", C.c_void_p),
        ("head_blob_split", C.c_void_p),
        ("occ", C.c_void_p),
        ("vol_folded", C.c_void_p * LEVELS),
        ("head_blob_ref", C.c_void_p),
    ]


PYRAMID_MAX_LEVELS = 4


class GpnerfSparseConv(C.Structure):
    """include/gpnerf_hip.h: one convolution of the sparse pyramid (gpnerf_sparse_pyramid_run)"""
    _fields_ = [("strided", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("form", C.c_int32),
                ("weight", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("weight_raw", C.c_void_p)]


class GpnerfPyramid(C.Structure):
    """include/gpnerf_hip.h: the buffers of one frame's pyramid (gpnerf_sparse_pyramid_plan / _run)"""
    _fields_ = [("n_levels", C.c_int32), ("m0", C.c_int32), ("dims0", C.c_int32 * 3),
                ("dims", (C.c_int32 * 3) * PYRAMID_MAX_LEVELS), ("cap", C.c_int32 * PYRAMID_MAX_LEVELS), ("ch", C.c_int32 * PYRAMID_MAX_LEVELS),
                ("coords0", C.c_void_p), ("grid0", C.c_void_p), ("dup_scratch", C.c_void_p),
                ("grid", C.c_void_p * PYRAMID_MAX_LEVELS), ("coords", C.c_void_p * PYRAMID_MAX_LEVELS), ("m", C.c_void_p * PYRAMID_MAX_LEVELS),
                ("vol", C.c_void_p * PYRAMID_MAX_LEVELS), ("feat_a", C.c_void_p), ("feat_b", C.c_void_p), ("feat_c", C.c_void_p)]



HEAD_FIELDS = [
    ("geo", "sigmahead.out_geometry_fc.0"), ("b1", "rgbhead.base_fc.0"), ("b2", "rgbhead.base_fc.2"),
    ("v1", "rgbhead.vis_fc.0"), ("v2", "rgbhead.vis_fc.2"), ("r1", "rgbhead.rgb_fc.0"),
    ("r2", "rgbhead.rgb_fc.2"), ("r3", "rgbhead.rgb_fc.4"), ("d1", "rgbhead.out_geometry_fc.0"),
    ("d2", "rgbhead.out_geometry_fc.2"), ("d3", "rgbhead.out_geometry_fc.4"), ("d4", "rgbhead.out_geometry_fc.6"),
]
HEAD_SHAPES = {
    "geo": (64, 128), "b1": (64, 105), "b2": (32, 64), "v1": (32, 32), "v2": (32, 32), "r1": (32, 96),
    "r2": (16, 32), "r3": (3, 16), "d1": (64, 134), "d2": (32, 64), "d3": (16, 32), "d4": (1, 16),
}


class GpnerfHeadParams(C.Structure):
    _fields_ = [f for s, _ in HEAD_FIELDS for f in ((s + "_w", FP), (s + "_b", FP))]


class GpnerfOutputs(C.Structure):
    _fields_ = [
        ("rgb", C.c_void_p), ("depth", C.c_void_p), ("acc", C.c_void_p), ("disp", C.c_void_p),
        ("weights", C.c_void_p), ("z_vals", C.c_void_p), ("rgb_in", C.c_void_p), ("ray_mask", C.c_void_p),
        ("raw", C.c_void_p), ("samples_done", C.c_void_p), ("step_stats", C.c_void_p),
    ]


FLAG_NEG_RAY = 1
FLAG_EARLY_TERM = 2
FLAG_OCC_CULL = 4
FLAG_SPLIT_F16 = 8
FLAG_FLIP_SAMPLES = 16
FLAG_SPLIT_GUARD = 32
FLAG_REF_ORDER = 64
FLAG_NO_EXITS = 128
FLAG_SHARED_DEVICE = 256
FOLD_FIRST_LEVEL = 2

# every symbol include/gpnerf_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "gpnerf_head_blob_floats": (C.c_int64, []),
    "gpnerf_pack_head": (C.c_int, [C.POINTER(GpnerfHeadParams), FP]),
    "gpnerf_pack_head_ref": (C.c_int, [C.POINTER(GpnerfHeadParams), FP]),
    "gpnerf_head_blob_split_floats": (C.c_int64, []),
    "gpnerf_pack_head_split": (C.c_int, [C.POINTER(GpnerfHeadParams), FP]),
    "gpnerf_render_fused": (C.c_int, [C.POINTER(GpnerfFrame), C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_float,
                                      C.c_void_p, C.POINTER(GpnerfOutputs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "gpnerf_fold_volumes": (C.c_int, [C.POINTER(GpnerfFrame), C.POINTER(C.c_void_p), C.c_void_p]),
    "gpnerf_render_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "gpnerf_render_guard_bytes": (C.c_size_t, [C.c_int64]),
    "gpnerf_sample_points": (C.c_int, [C.POINTER(GpnerfFrame), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "gpnerf_sample_volume": (C.c_int, [C.POINTER(GpnerfFrame), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gpnerf_project_gather": (C.c_int, [C.POINTER(GpnerfFrame), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "gpnerf_head_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gpnerf_sigma_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_rgb_head_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gpnerf_composite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                   C.POINTER(GpnerfOutputs), C.c_void_p]),
    "gpnerf_make_rays": (C.c_int, [C.c_int32, C.c_int32, DP, DP, DP, FP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_select_pixels": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, FP, FP, FP, FP, FP, FP, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_make_rays_demo": (C.c_int, [C.c_int32, C.c_int32, FP, FP, FP, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_conv_packed_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "gpnerf_conv_pack_weight": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gpnerf_conv_out_tiles": (C.c_int32, [C.c_int32] * 5),
    "gpnerf_conv2d_nhwc": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gpnerf_conv2d_nhwc_exact": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_void_p]),
    "gpnerf_conv2d_norm_nhwc": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gpnerf_conv2d_norm_cat_nhwc": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_void_p]),
    "gpnerf_norm_apply_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                         C.c_void_p]),
    "gpnerf_instance_norm_nhwc_scratch_bytes": (C.c_int64, [C.c_int32, C.c_int64, C.c_int32]),
    "gpnerf_instance_norm_act_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                                C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_upsample2x_nhwc": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gpnerf_vertex_attention": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p]),
    "gpnerf_build_occupancy": (C.c_int, [C.POINTER(GpnerfFrame), C.c_void_p, C.c_void_p]),
    "gpnerf_sparse_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "gpnerf_sparse_conv3": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_sparse_conv3_mfma": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_sparse_conv3_mfma16": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                             C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_sparse_packed_weight16_bytes": (C.c_int64, [C.c_int32]),
    "gpnerf_sparse_pack_weight16": (C.c_int, [FP, C.c_int32, C.c_int32, C.c_void_p]),
    "gpnerf_sparse_packed_weight_floats": (C.c_int64, [C.c_int32]),
    "gpnerf_sparse_pack_weight": (C.c_int, [FP, C.c_int32, C.c_int32, FP]),
    "gpnerf_sparse_pyramid_plan": (C.c_int, [C.POINTER(GpnerfPyramid), C.c_void_p]),
    "gpnerf_sparse_pyramid_run": (C.c_int, [C.POINTER(GpnerfPyramid), C.c_void_p, C.c_int32, C.POINTER(GpnerfSparseConv), C.c_int32, C.c_void_p]),
    "gpnerf_sparse_merge_duplicates": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                                 C.c_void_p, C.c_void_p]),
    "gpnerf_sparse_down_sites": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_void_p]),
    "gpnerf_sparse_to_dense": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "gpnerf_zero_volume": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]),
    "gpnerf_sparse_scatter_dense": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                              C.POINTER(C.c_int32), C.c_void_p, C.c_int32, C.c_void_p]),
    "gpnerf_relayout_volume": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gpnerf_relayout_featmaps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gpnerf_relayout_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gpnerf_patch_order_scratch_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gpnerf_patch_order": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpnerf_head_layout": (C.c_int, [C.POINTER(C.c_int32)]),
    "gpnerf_strerror": (C.c_char_p, [C.c_int]),
    "gpnerf_rays_per_tile": (C.c_int32, []),
    "gpnerf_build_info": (C.c_char_p, []),
}

_lib = None


class GpnerfError(RuntimeError):
    pass


def lib():
    """Load the HIP library; fails loudly when it is absent (no CPU / PyTorch fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GpnerfError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gp-nerf_amd/csrc` (hipcc --offload-arch=gfx950). There is no fallback path.")
        # torch first: it bundles its own HIP runtime, and the one that is loaded first serves the whole process.  Loaded the
        # other way round, this library binds to the system runtime while torch's streams and allocations live in another
        # one, and every launch fails.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the ABI and the header drift apart
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        raise GpnerfError(f"{what} failed: {lib().gpnerf_strerror(code).decode()} ({code})")
