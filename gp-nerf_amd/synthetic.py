"""Deterministic synthetic frames for the per-ray render path.

Everything here is numpy + PCG64 so the GPU box regenerates byte-identical inputs
without the reference tree.  The dict returned by :func:`make_scene` follows the
reference batch schema (libs/datasets/ZjumocapDataset.py:464-517) plus the two
per-frame products the hot path consumes but does not compute itself:

* ``featmaps``  [V,32,H/4,W/4]  -- what ``encoder(src_imgs)`` would return
  (libs/renders/BaseRender.py:222)
* ``volumes``   4 x [1,32,D/2^k,H/2^k,W/2^k] -- what ``SparseConvNet.forward`` makes
  dense before sampling it (libs/nerfheads/networks/SparseConvNet.py:105-116)

and the per-ray MLP parameters under the reference's state_dict names
(SURVEY.md Appendix B).

``rays_numpy`` restates get_rays/get_near_far
(libs/datasets/data_utils.py:47-63,96-130) for scene construction only; the
product's ray kernel is checked against the C oracle, not against this file.
"""
import math
from collections import OrderedDict

import numpy as np

N_VIEWS = 3
FEAT_CH = 32
N_LEVELS = 4

# (name, out, in) of every Linear on the per-ray path, reference parameter names
# (libs/nerfheads/trainhead.py:39-40,85-110)
HEAD_LAYERS = [
    ("sigmahead.out_geometry_fc.0", 64, 128),
    ("rgbhead.base_fc.0", 64, 105),
    ("rgbhead.base_fc.2", 32, 64),
    ("rgbhead.vis_fc.0", 32, 32),
    ("rgbhead.vis_fc.2", 32, 32),
    ("rgbhead.rgb_fc.0", 32, 96),
    ("rgbhead.rgb_fc.2", 16, 32),
    ("rgbhead.rgb_fc.4", 3, 16),
    ("rgbhead.out_geometry_fc.0", 64, 134),
    ("rgbhead.out_geometry_fc.2", 32, 64),
    ("rgbhead.out_geometry_fc.4", 16, 32),
    ("rgbhead.out_geometry_fc.6", 1, 16),
]


def _rng(seed, stream):
    return np.random.Generator(np.random.PCG64([int(seed), int(stream)]))


def make_head_weights(seed=0, bias_std=0.0, sigma_bias=0.0):
    """kaiming-normal W (std = sqrt(2/fan_in)), zero b: trainhead.py:13-17.

    ``bias_std`` > 0 draws non-zero biases so tests exercise the bias path;
    ``sigma_bias`` shifts the last density bias so that a useful share of
    samples has sigma > 0 (random-init nets otherwise sit near ReLU(0)).
    """
    g = _rng(seed, 101)
    sd = OrderedDict()
    for name, n_out, n_in in HEAD_LAYERS:
        w = g.standard_normal((n_out, n_in), dtype=np.float32) * np.float32(math.sqrt(2.0 / n_in))
        if bias_std > 0:
            b = g.standard_normal((n_out,), dtype=np.float32) * np.float32(bias_std)
        else:
            b = np.zeros((n_out,), np.float32)
        sd[name + ".weight"] = w.astype(np.float32)
        sd[name + ".bias"] = b.astype(np.float32)
    if sigma_bias != 0.0:
        sd["rgbhead.out_geometry_fc.6.bias"] = sd["rgbhead.out_geometry_fc.6.bias"] + np.float32(sigma_bias)
    return sd


def rodrigues(rvec):
    rvec = np.asarray(rvec, np.float64)
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3, dtype=np.float32)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)
    return R.astype(np.float32)


def rot_y(theta):
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def rays_numpy(H, W, K, R, T, world_bounds):
    """Pixel rays + AABB slab test, restating data_utils.py:47-63 and :96-130.

    Returns ray_o [N,3], ray_d [N,3] (un-normalised, with the +1e-5 clamp applied
    in place as the reference does), near [N], far [N], mask_at_box [H*W] bool.
    """
    K = np.asarray(K, np.float64)
    R = np.asarray(R, np.float64)
    T = np.asarray(T, np.float64).reshape(3, 1)
    R_inv = np.linalg.inv(R)
    Tw = -R_inv @ T
    rays_o = Tw.ravel()
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pixel_camera = np.dot(xy1, np.linalg.inv(K).T)
    pixel_world = (pixel_camera @ R_inv.T) + Tw.ravel()[None, None]
    rays_d = pixel_world - rays_o[None, None]
    rays_o = np.broadcast_to(rays_o, rays_d.shape)
    ray_o = rays_o.reshape(-1, 3).astype(np.float32)
    ray_d = rays_d.reshape(-1, 3).astype(np.float32)

    bounds = np.asarray(world_bounds, np.float32) + np.array([-0.01, 0.01], np.float64)[:, None]
    nominator = bounds[None] - ray_o[:, None]
    ray_d = ray_d.copy()
    ray_d[np.abs(ray_d) < 1e-5] = 1e-5
    d_intersect = (nominator / ray_d[:, None]).reshape(-1, 6)
    p_intersect = d_intersect[..., None] * ray_d[:, None] + ray_o[:, None]
    min_x, min_y, min_z, max_x, max_y, max_z = bounds.ravel()
    eps = 1e-6
    p_mask = (
        (p_intersect[..., 0] >= (min_x - eps))
        * (p_intersect[..., 0] <= (max_x + eps))
        * (p_intersect[..., 1] >= (min_y - eps))
        * (p_intersect[..., 1] <= (max_y + eps))
        * (p_intersect[..., 2] >= (min_z - eps))
        * (p_intersect[..., 2] <= (max_z + eps))
    )
    mask_at_box = p_mask.sum(-1) == 2
    p_int = p_intersect[mask_at_box][p_mask[mask_at_box]].reshape(-1, 2, 3)
    o = ray_o[mask_at_box]
    d = ray_d[mask_at_box]
    norm_ray = np.linalg.norm(d, axis=1)
    sign = np.array(((p_int[:, 0] - o) * d).sum(axis=1) < 0.0, dtype=np.int64) * -2 + 1
    d0 = np.linalg.norm(p_int[:, 0] - o, axis=1) / norm_ray * sign
    d1 = np.linalg.norm(p_int[:, 1] - o, axis=1) / norm_ray * sign
    near = np.minimum(d0, d1).astype(np.float32)
    far = np.maximum(d0, d1).astype(np.float32)
    return o.astype(np.float32), d.astype(np.float32), near, far, mask_at_box


def out_shape_dhw(bounds_smpl, voxel):
    """ZjumocapDataset.py:243-254: ceil(extent/voxel) rounded up to the next x32."""
    mn = bounds_smpl[0][[2, 1, 0]].astype(np.float64)
    mx = bounds_smpl[1][[2, 1, 0]].astype(np.float64)
    sh = np.ceil((mx - mn) / np.asarray(voxel, np.float64)).astype(np.int32)
    return ((sh | 31) + 1).astype(np.int32)


def make_scene(
    H=64,
    W=64,
    seed=0,
    focal_mul=None,
    fill="full",
    aabb_half=(0.5, 0.9, 0.25),
    voxel=0.005,
    pose="identity",
    n_verts=6890,
    bias_std=0.0,
    sigma_bias=0.0,
    vol_scale=1.0,
    max_rays=None,
    make_volumes=True,
    neg_cams=False,
    vol_occupancy=None,
):
    """Build one synthetic frame.

    fill="full": focal length chosen so that every pixel's ray crosses the SMPL
    AABB (N = H*W, the accounting SURVEY.md §8d uses for config 2);
    fill="survey": f = 1.05*W as written in §8d (about a fifth of the pixels hit).
    pose="identity": Rh=I, Th=0 (§8d); pose="random": a non-trivial Rh/Th so the
    world->SMPL transform (BaseRender.py:52-60) is exercised.
    """
    g = _rng(seed, 7)
    hx, hy, hz = [float(a) for a in aabb_half]

    # SMPL-frame vertices ~ U(AABB) and bounds with z -/+ 0.05 (ZjumocapDataset.py:236-240)
    verts = (g.random((n_verts, 3), dtype=np.float32) * 2 - 1) * np.array([hx, hy, hz], np.float32)
    # pin the extremes so the bounds are exactly the nominal box
    verts[0] = [-hx, -hy, -hz]
    verts[1] = [hx, hy, hz]
    bounds = np.stack([verts.min(0), verts.max(0)], 0).astype(np.float32)
    bounds[0, 2] -= 0.05
    bounds[1, 2] += 0.05

    if pose == "identity":
        Rh = np.eye(3, dtype=np.float32)
        Th = np.zeros((1, 3), np.float32)
    else:
        Rh = rodrigues([0.15, -0.35, 0.1])
        Th = np.array([[0.07, -0.04, 0.11]], np.float32)

    # world-frame vertices: xyz @ Rh^T + Th  (BaseRender.py:128-131)
    verts_world = verts @ Rh.T + Th
    can_bounds = np.stack([verts_world.min(0), verts_world.max(0)], 0).astype(np.float32)
    can_bounds[0, 2] -= 0.05
    can_bounds[1, 2] += 0.05

    voxel_size = np.array([voxel] * 3, np.float32)
    out_sh = out_shape_dhw(bounds, voxel_size)

    # target camera
    if focal_mul is None:
        if fill == "full":
            # front face of the (padded) world box at z_cam = 3 - |z|max; need W/2/f * z_far_face <= x half extent
            zc = 3.0 + float(np.abs(can_bounds[:, 2]).max()) + 0.01
            need_x = (W / 2.0) * zc / (float(min(-can_bounds[0, 0], can_bounds[1, 0])) - 0.0)
            need_y = (H / 2.0) * zc / (float(min(-can_bounds[0, 1], can_bounds[1, 1])) - 0.0)
            focal = 1.02 * max(need_x, need_y)
        else:
            focal = 1.05 * W
    else:
        focal = focal_mul * W
    K = np.array([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1]], np.float32)
    R = np.eye(3, dtype=np.float32)
    T = np.array([[0.0], [0.0], [3.0]], np.float32)

    ray_o, ray_d, near, far, mask_at_box = rays_numpy(H, W, K, R, T, can_bounds)
    if max_rays is not None and ray_o.shape[0] > max_rays:
        ray_o, ray_d, near, far = ray_o[:max_rays], ray_d[:max_rays], near[:max_rays], far[:max_rays]

    # source views: the target rig rotated about y by -30, 0, +30 degrees
    src_Ks = np.stack([K] * N_VIEWS, 0).astype(np.float32)
    src_poses = []
    for deg in (-30.0, 0.0, 30.0):
        Rv = rot_y(-math.radians(deg)).astype(np.float32)
        src_poses.append(np.concatenate([Rv, T], 1))
    src_poses = np.stack(src_poses, 0).astype(np.float32)
    if neg_cams:
        # THuman-style convention exercised by neg_ray (BaseRender.py:317-320): camera-space
        # coordinates negated, so visible points have z < 0 and project to the same pixels
        src_poses = -src_poses

    src_imgs01 = g.random((N_VIEWS, 3, H, W), dtype=np.float32)
    src_imgs = (src_imgs01 - 0.5) / 0.5  # dataset normalisation (transform.py:349-373)
    featmaps = g.standard_normal((N_VIEWS, FEAT_CH, H // 4, W // 4), dtype=np.float32)

    volumes = []
    if make_volumes:
        occ_coarse = None
        if vol_occupancy is not None:
            # sparse, non-negative pyramid like the sparse conv net's ReLU outputs: a random block mask at the
            # coarsest level, nearest-upsampled to the finer ones, times |N(0,1)|
            d4, h4, w4 = [int(s) >> N_LEVELS for s in out_sh]
            occ_coarse = _rng(seed, 300).random((d4, h4, w4)) < float(vol_occupancy)
        for k in range(1, N_LEVELS + 1):
            d, h, w = [int(s) >> k for s in out_sh]
            gv = _rng(seed, 200 + k)
            v = gv.standard_normal((1, FEAT_CH, d, h, w), dtype=np.float32) * np.float32(vol_scale)
            if occ_coarse is not None:
                r = 1 << (N_LEVELS - k)
                m = np.repeat(np.repeat(np.repeat(occ_coarse, r, 0), r, 1), r, 2)
                v = np.abs(v) * m[None, None].astype(np.float32)
            volumes.append(v)

    # voxel index of every vertex, dhw order (ZjumocapDataset.py:243-247); only the
    # (out-of-scope) sparse volume builder reads it, kept for schema completeness
    dhw = verts[:, [2, 1, 0]]
    coord = np.round((dhw - bounds[0][[2, 1, 0]]) / voxel_size).astype(np.int32)

    n = ray_o.shape[0]
    scene = {
        "ray_o": ray_o[None],
        "ray_d": ray_d[None],
        "near": near[None],
        "far": far[None],
        "mask_at_box": mask_at_box[None],
        "body_msk": np.ones((1, n), np.float32),
        "src_imgs": src_imgs[None].astype(np.float32),
        "src_Ks": src_Ks[None],
        "src_poses": src_poses[None],
        "target_K": K[None],
        "target_pose": np.concatenate([R, T], 1)[None].astype(np.float32),
        "feature": np.concatenate([verts, np.zeros_like(verts)], 1)[None].astype(np.float32),
        "coord": coord[None],
        "bounds": bounds[None],
        "can_bounds": can_bounds[None],
        "out_sh": out_sh[None],
        "Rh": Rh[None],
        "R": Rh[None],
        "Th": Th[None],
        "voxel_size": voxel_size,
        "featmaps": featmaps,
        "volumes": volumes,
        "H": H,
        "W": W,
    }
    scene["head"] = make_head_weights(seed, bias_std=bias_std, sigma_bias=sigma_bias)
    return scene
