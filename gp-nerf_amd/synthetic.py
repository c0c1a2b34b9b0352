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


def make_head_weights(seed=0, bias_std=0.0, sigma_bias=0.0, head_scale=1.0):
    """kaiming-normal W (std = sqrt(2/fan_in)), zero b: trainhead.py:13-17.

    ``bias_std`` > 0 draws non-zero biases so tests exercise the bias path;
    ``sigma_bias`` shifts the last density bias so that a useful share of
    samples has sigma > 0 (random-init nets otherwise sit near ReLU(0));
    ``head_scale`` multiplies every weight matrix (a trained head is not at its initialisation's scale: the "trained-like"
    golden cases use 2 - 3, where the ELU chains leave their linear range and sigmoid / exp(-sigma) saturate).
    """
    g = _rng(seed, 101)
    sd = OrderedDict()
    for name, n_out, n_in in HEAD_LAYERS:
        w = g.standard_normal((n_out, n_in), dtype=np.float32) * np.float32(math.sqrt(2.0 / n_in))
        if head_scale != 1.0:
            w = w * np.float32(head_scale)
        if bias_std > 0:
            b = g.standard_normal((n_out,), dtype=np.float32) * np.float32(bias_std)
        else:
            b = np.zeros((n_out,), np.float32)
        sd[name + ".weight"] = w.astype(np.float32)
        sd[name + ".bias"] = b.astype(np.float32)
    if sigma_bias != 0.0:
        sd["rgbhead.out_geometry_fc.6.bias"] = sd["rgbhead.out_geometry_fc.6.bias"] + np.float32(sigma_bias)
    return sd



def encoder_param_shapes(out_ch=32):
    """(state_dict key, shape) of every parameter of the image encoder, in the checkpoint's naming
    (libs/encoders/UNet.py:154-176): stem, three residual stages of 3/4/6 units, two decoder steps, output conv."""
    t = [("conv1.weight", (64, 3, 7, 7)), ("bn1.weight", (64,)), ("bn1.bias", (64,))]
    cin = 64
    for stage, (width, units) in enumerate(((64, 3), (128, 4), (256, 6)), start=1):
        for u in range(units):
            p = f"layer{stage}.{u}."
            t += [(p + "conv1.weight", (width, cin if u == 0 else width, 3, 3)), (p + "bn1.weight", (width,)),
                  (p + "bn1.bias", (width,)), (p + "conv2.weight", (width, width, 3, 3)), (p + "bn2.weight", (width,)),
                  (p + "bn2.bias", (width,))]
            if u == 0:       # every stage is entered with stride 2 -> projected shortcut
                t += [(p + "downsample.0.weight", (width, cin, 1, 1)), (p + "downsample.1.weight", (width,)),
                      (p + "downsample.1.bias", (width,))]
        cin = width
    for name, ci, co in (("upconv3.conv", 256, 128), ("iconv3", 256, 128), ("upconv2.conv", 128, 64), ("iconv2", 128, out_ch)):
        t += [(name + ".conv.weight", (co, ci, 3, 3)), (name + ".conv.bias", (co,)), (name + ".bn.weight", (co,)),
              (name + ".bn.bias", (co,))]
    t += [("out_conv.weight", (out_ch, out_ch, 1, 1)), ("out_conv.bias", (out_ch,))]
    return t


def make_encoder_weights(seed=0, out_ch=32, gamma_range=None, beta_std=0.1, conv_scale=1.0):
    """Seeded encoder parameters: conv kernels ~ N(0, 2/fan_in), norm scales ~ 1 + 0.1 N, every bias ~ 0.1 N (so that the
    affine and bias paths are exercised).  Drawn in `encoder_param_shapes()` order from PCG64([seed, 303]).
    "Trained-like" variants: gamma_range (lo, hi) draws every InstanceNorm scale from U(lo, hi) (the standard normal draw mapped
    through its CDF, so the stream stays aligned), beta_std widens the biases, conv_scale multiplies the kernels."""
    g = _rng(seed, 303)
    sd = OrderedDict()
    for key, shape in encoder_param_shapes(out_ch):
        x = g.standard_normal(shape, dtype=np.float32)
        if len(shape) == 4:
            x *= np.float32(math.sqrt(2.0 / (shape[1] * shape[2] * shape[3])) * conv_scale)
        elif key.endswith("weight"):
            if gamma_range is None:
                x = np.float32(1.0) + np.float32(0.1) * x
            else:
                u = 0.5 * (1.0 + np.vectorize(math.erf)(x.astype(np.float64) / math.sqrt(2.0)))
                x = (gamma_range[0] + (gamma_range[1] - gamma_range[0]) * u).astype(np.float32)
        else:
            x *= np.float32(beta_std)
        sd[key] = x.astype(np.float32)
    return sd


def make_encoder_images(H, W, seed=0, views=3):
    """Source images as the dataset hands them over: [V,3,H,W] in [-1,1], smooth + noise so that InstanceNorm sees structure."""
    g = _rng(seed, 304)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H, dtype=np.float32), np.linspace(-1, 1, W, dtype=np.float32), indexing="ij")
    out = np.empty((views, 3, H, W), np.float32)
    for v in range(views):
        for c in range(3):
            a, b, ph = g.uniform(1.0, 6.0, 3).astype(np.float32)
            out[v, c] = 0.6 * np.sin(a * xx + b * yy + ph) + 0.4 * g.uniform(-1, 1, (H, W)).astype(np.float32)
    return np.clip(out, -1, 1).astype(np.float32)



def make_attention_case(n, code_dim, seed=0, views=3, kv_dim=32):
    """Seeded inputs of the vertex-code attention (trainhead.py:48-52): parameters by state_dict key, vertex codes [n,d],
    per-view vertex features [n,V,32]."""
    g = _rng(seed, 404)
    sd = OrderedDict()
    for key, shape in (("w_qs.weight", (code_dim, code_dim)), ("w_ks.weight", (code_dim, kv_dim)), ("w_vs.weight", (code_dim, kv_dim)),
                       ("fc.weight", (code_dim, code_dim))):
        sd[key] = (g.standard_normal(shape, dtype=np.float32) * np.float32(math.sqrt(1.0 / shape[1]))).astype(np.float32)
    sd["layer_norm.weight"] = np.ones((code_dim,), np.float32)
    sd["layer_norm.bias"] = np.zeros((code_dim,), np.float32)
    code = g.standard_normal((n, code_dim), dtype=np.float32)
    feat = (g.standard_normal((n, views, kv_dim), dtype=np.float32) * np.float32(1.5)).astype(np.float32)
    return sd, code, feat


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
    """Pixel rays and their entry / exit distances through the padded world AABB, with the semantics of
    data_utils.py:47-63 (get_rays) and :96-130 (get_near_far): un-normalised directions, +1e-5 clamp of small direction
    components, box padded by 1 cm, a ray is kept iff exactly two of its six plane hits lie on the box, distances signed
    by the first hit.  Returns ray_o [N,3], ray_d [N,3], near [N], far [N], mask_at_box [H*W] (raster order)."""
    Kinv = np.linalg.inv(np.asarray(K, np.float64))
    Rinv = np.linalg.inv(np.asarray(R, np.float64))
    centre = -(Rinv @ np.asarray(T, np.float64).reshape(3))
    cols, rows = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    pix = np.stack([cols.ravel(), rows.ravel(), np.ones(H * W, np.float32)], 1)
    world = (pix @ Kinv.T) @ Rinv.T + centre
    origin = np.broadcast_to(centre, world.shape).astype(np.float32)
    direction = (world - centre).astype(np.float32)
    direction[np.abs(direction) < 1e-5] = 1e-5

    box = np.asarray(world_bounds, np.float32).astype(np.float64)
    lo, hi = box[0] - 0.01, box[1] + 0.01
    tol = 1e-6
    hits, on_box = [], []
    for plane in (lo, hi):                      # plane order: x/y/z of the low corner, then of the high corner
        for axis in range(3):
            t = (plane[axis] - origin[:, axis]) / direction[:, axis]
            pt = origin + t[:, None] * direction
            hits.append(pt)
            on_box.append(np.all((pt >= lo - tol) & (pt <= hi + tol), axis=1))
    hits, on_box = np.stack(hits, 1), np.stack(on_box, 1)          # [P,6,3], [P,6]
    mask_at_box = on_box.sum(1) == 2
    sel_hits, sel_on = hits[mask_at_box], on_box[mask_at_box]
    first = sel_on.argmax(1)
    second = 5 - sel_on[:, ::-1].argmax(1)
    idx = np.arange(sel_hits.shape[0])
    p_a, p_b = sel_hits[idx, first], sel_hits[idx, second]
    o, d = origin[mask_at_box], direction[mask_at_box]
    length = np.linalg.norm(d, axis=1)
    sign = np.where(((p_a - o) * d).sum(1) < 0.0, -1.0, 1.0)       # both distances take the first hit's sign
    t_a = np.linalg.norm(p_a - o, axis=1) / length * sign
    t_b = np.linalg.norm(p_b - o, axis=1) / length * sign
    near, far = np.minimum(t_a, t_b).astype(np.float32), np.maximum(t_a, t_b).astype(np.float32)
    return o.astype(np.float32), d.astype(np.float32), near, far, mask_at_box


def make_ray_camera(kind, H=512, W=512):
    """Target cameras for the ray-generation vectors, float64 as the dataset reads them from annots.npy
    (ZjumocapDataset.py:360-380), with the world box of the identity-pose scene (float32, as prepare_input makes it).
      axis    : R = I, principal point on a pixel centre: the centre row / column have a zero direction component, which
                get_near_far replaces by +1e-5 (data_utils.py:101)
      edge    : R = I, focal length chosen so that pixel column W/2 + 100 passes exactly through the front-right edge of the
                padded box (two plane hits coincide), i.e. grazing rays on both sides of the eps = 1e-6 test (:107-115)
      oblique : generic rotation / translation / intrinsics with no float32-representable entries
    Returns K [3,3], R [3,3], T [3] (float64) and bounds [2,3] (float32)."""
    bounds = np.array([[-0.5, -0.9, -0.3], [0.5, 0.9, 0.3]], np.float32)
    if kind == "axis":
        K = np.array([[1.05 * W, 0, W / 2.0], [0, 1.05 * W, H / 2.0], [0, 0, 1]], np.float64)
        R, T = np.eye(3), np.array([0.0, 0.0, 3.0])
    elif kind == "edge":
        f = 100.0 * (3.0 - (float(np.float32(0.3)) + 0.01)) / (float(np.float32(0.5)) + 0.01)
        K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float64)
        R, T = np.eye(3), np.array([0.0, 0.0, 3.0])
    elif kind == "oblique":
        rv = np.array([0.31, -0.52, 0.17])
        th = np.linalg.norm(rv)
        k = rv / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)
        T = np.array([0.137, -0.211, 2.9031])
        K = np.array([[W * 0.9137, 0, W / 2.0 + 3.37], [0, W * 0.9291, H / 2.0 - 5.11], [0, 0, 1]], np.float64)
    else:
        raise ValueError(kind)
    return K, R, T, bounds


def out_shape_dhw(bounds_smpl, voxel):
    """ZjumocapDataset.py:243-254: ceil(extent/voxel) rounded up to the next x32."""
    mn = bounds_smpl[0][[2, 1, 0]].astype(np.float64)
    mx = bounds_smpl[1][[2, 1, 0]].astype(np.float64)
    sh = np.ceil((mx - mn) / np.asarray(voxel, np.float64)).astype(np.int32)
    return ((sh | 31) + 1).astype(np.int32)


def body_vertices(n_verts=6890, aabb_half=(0.5, 0.9, 0.25)):
    """A deterministic SMPL-SHAPED point set: `n_verts` points on the surface of a capsule-limbed figure in an A-pose (ellipsoid
    torso and head, tapered capsules for neck / upper and lower arms / thighs / shanks, small ellipsoids for hands and feet), laid
    out for the nominal SMPL box of half extents (0.5, 0.9, 0.25) m and scaled per axis to `aabb_half`.  Vertex budget per part
    follows SMPL's own density (head and hands much denser than the torso), so that -- as with a real SMPL mesh at 5 mm voxels --
    torso vertices mostly sit in voxels of their own while head / hand / foot vertices share voxels.  The uniform-in-the-box
    vertices of the other scenes put a sparse-convolution site into every region of the box; these put them on a body's surface,
    which is what decides the pyramid's site counts, the occupancy volume and the progressive renderer's cull rate
    (libs/datasets/ZjumocapDataset.py:207-256 quantises the vertices the same way).  Points: Fibonacci lattices, no randomness."""
    golden = math.pi * (3.0 - math.sqrt(5.0))

    def ellipsoid(n, c, r):
        i = np.arange(n, dtype=np.float64) + 0.5
        z = 1.0 - 2.0 * i / n
        rad = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        th = golden * i
        return np.asarray(c, np.float64) + np.stack([rad * np.cos(th), z, rad * np.sin(th)], 1) * np.asarray(r, np.float64)

    def capsule(n, a, b, ra, rb):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        axis = b - a
        L = np.linalg.norm(axis)
        w = axis / L
        u = np.cross(w, [0.0, 0.0, 1.0] if abs(w[2]) < 0.9 else [1.0, 0.0, 0.0])
        u /= np.linalg.norm(u)
        v = np.cross(w, u)
        i = np.arange(n, dtype=np.float64) + 0.5
        t = i / n
        th = golden * i
        r = ra + (rb - ra) * t
        return a + t[:, None] * axis + r[:, None] * (np.cos(th)[:, None] * u + np.sin(th)[:, None] * v)

    def cap(n, c, r, half_angle):
        """n lattice points on the part of an ellipsoid that faces +z within `half_angle` (the face: SMPL spends ~half of the head's
        vertices on eyes, nose, lips and ears, 2-4 mm apart -- several per 5 mm voxel)"""
        i = np.arange(n, dtype=np.float64) + 0.5
        z = 1.0 - (1.0 - math.cos(half_angle)) * i / n
        rad = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        th = golden * i
        return np.asarray(c, np.float64) + np.stack([rad * np.cos(th), rad * np.sin(th), z], 1) * np.asarray(r, np.float64)

    parts = [ellipsoid(600, (0.0, 0.66, 0.02), (0.085, 0.115, 0.10)),                     # head
             cap(500, (0.0, 0.66, 0.02), (0.085, 0.115, 0.10), math.radians(38.0)),       # face
             capsule(90, (0.0, 0.50, 0.0), (0.0, 0.58, 0.01), 0.055, 0.05),               # neck
             ellipsoid(1700, (0.0, 0.16, 0.0), (0.175, 0.36, 0.115))]                     # torso + pelvis
    for sx in (-1.0, 1.0):
        parts += [capsule(260, (sx * 0.19, 0.43, 0.0), (sx * 0.33, 0.20, 0.0), 0.05, 0.04),          # upper arm
                  capsule(240, (sx * 0.33, 0.20, 0.0), (sx * 0.44, -0.02, 0.03), 0.04, 0.03),        # forearm
                  ellipsoid(400, (sx * 0.465, -0.085, 0.04), (0.028, 0.075, 0.04)),                  # hand
                  capsule(430, (sx * 0.095, -0.12, 0.0), (sx * 0.12, -0.50, 0.015), 0.08, 0.055),    # thigh
                  capsule(370, (sx * 0.12, -0.50, 0.015), (sx * 0.125, -0.84, 0.0), 0.055, 0.04),    # shank
                  ellipsoid(300, (sx * 0.13, -0.875, 0.06), (0.045, 0.035, 0.115))]                  # foot
    pts = np.concatenate(parts, 0)
    assert pts.shape[0] == 6890
    if n_verts != 6890:
        pts = pts[np.linspace(0, 6889, n_verts).round().astype(np.int64)]
    pts[:, 1] -= 0.5 * (pts[:, 1].max() + pts[:, 1].min())          # centred like the nominal box
    scale = np.asarray(aabb_half, np.float64) / np.array([0.5, 0.9, 0.25])
    return (pts * scale).astype(np.float32)


def pyramid_sites(coord, out_sh, n_levels=N_LEVELS):
    """Which voxels of each dense level the sparse pyramid writes for vertex voxels `coord` [M,3] (d,h,w): level k's sites are the
    level k-1 sites' images under a 3x3x3 / stride-2 / pad-1 sparse convolution (SparseConvNet.py:85-92,105-111): output o is active
    iff some active input sits at 2 o - 1 + t, t in {0,1,2}^3.  Returns n_levels bool arrays [D_k,H_k,W_k]."""
    dims = np.asarray(out_sh, np.int64)
    cur = np.zeros(tuple(dims), bool)
    c = np.asarray(coord, np.int64)
    ok = np.all((c >= 0) & (c < dims), 1)
    cur[tuple(c[ok].T)] = True
    out = []
    for _ in range(n_levels):
        dims = (dims + 2 - 3) // 2 + 1
        nxt = np.zeros(tuple(dims), bool)
        p = np.argwhere(cur)
        for t in np.ndindex(3, 3, 3):
            num = p + 1 - np.asarray(t)
            keep = np.all(num % 2 == 0, 1)
            o = num[keep] // 2
            o = o[np.all((o >= 0) & (o < dims), 1)]
            nxt[tuple(o.T)] = True
        out.append(nxt)
        cur = nxt
    return out


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
    neg_target=False,
    head_scale=1.0,
    feat_scale=1.0,
    feat_tail=0.0,
    vol_relu=False,
    body="box",
):
    """Build one synthetic frame.

    fill="full": focal length chosen so that every pixel's ray crosses the SMPL
    AABB (N = H*W, the accounting SURVEY.md §8d uses for config 2);
    fill="survey": f = 1.05*W as written in §8d (about a fifth of the pixels hit).
    pose="identity": Rh=I, Th=0 (§8d); pose="random": a non-trivial Rh/Th so the
    world->SMPL transform (BaseRender.py:52-60) is exercised.
    "Trained-like" knobs (the distributions a checkpoint has and an initialisation has not): head_scale (weights x this),
    feat_scale (feature maps x this), feat_tail t > 0 (feature maps and volumes x exp(t N'): log-normal heavy tails),
    vol_relu (the dense levels are max(., 0): the sparse conv net ends in ReLU, so about half of every level is exactly zero),
    vol_scale (volumes x this).
    body="capsules": the vertices are body_vertices() -- a capsule-limbed figure's surface instead of points uniform in the box --
    the bounds are theirs (ZjumocapDataset.py:236-240), and the dense levels are non-negative values on exactly the voxels the
    sparse pyramid would write for those vertices (pyramid_sites), zero elsewhere: site counts, occupancy and cull rates of a
    person-shaped frame.
    """
    g = _rng(seed, 7)
    hx, hy, hz = [float(a) for a in aabb_half]

    # SMPL-frame vertices ~ U(AABB) and bounds with z -/+ 0.05 (ZjumocapDataset.py:236-240)
    verts = (g.random((n_verts, 3), dtype=np.float32) * 2 - 1) * np.array([hx, hy, hz], np.float32)
    # pin the extremes so the bounds are exactly the nominal box
    verts[0] = [-hx, -hy, -hz]
    verts[1] = [hx, hy, hz]
    if body == "capsules":
        verts = body_vertices(n_verts, (hx, hy, hz))                # (the generator's draws above stay: the other arrays keep their streams)
    elif body != "box":
        raise ValueError(body)
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
    if feat_tail > 0.0:
        featmaps = featmaps * np.exp(np.float32(feat_tail) * _rng(seed, 150).standard_normal(featmaps.shape, dtype=np.float32))
    if feat_scale != 1.0:
        featmaps = (featmaps * np.float32(feat_scale)).astype(np.float32)

    volumes = []
    site_masks = None
    if make_volumes and body == "capsules":
        dhw0 = verts[:, [2, 1, 0]]
        site_masks = pyramid_sites(np.round((dhw0 - bounds[0][[2, 1, 0]]) / voxel_size).astype(np.int64), out_sh)
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
            if feat_tail > 0.0:
                v = v * np.exp(np.float32(feat_tail) * _rng(seed, 250 + k).standard_normal(v.shape, dtype=np.float32))
            if vol_relu:
                v = np.maximum(v, np.float32(0.0))
            if occ_coarse is not None:
                r = 1 << (N_LEVELS - k)
                m = np.repeat(np.repeat(np.repeat(occ_coarse, r, 0), r, 1), r, 2)
                v = np.abs(v) * m[None, None].astype(np.float32)
            if site_masks is not None:
                v = np.abs(v) * site_masks[k - 1][None, None].astype(np.float32)
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
        # neg_target: the target camera in the negated (THuman-style) convention as well; only the progressive renderer
        # derives rays from target_pose (demo_render.py:201-239), ray_o/ray_d above stay those of the un-negated camera
        "target_pose": (np.concatenate([R, T], 1)[None] * (-1.0 if neg_target else 1.0)).astype(np.float32),
        "target_K_inv": np.linalg.inv(K)[None],          # float32, as ZjumocapDataset.py:480 makes it
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
    scene["head"] = make_head_weights(seed, bias_std=bias_std, sigma_bias=sigma_bias, head_scale=head_scale)
    return scene
