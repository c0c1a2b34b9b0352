#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference); the .npz files it
writes are committed, this script is committed, nothing of the reference is.

What executes here is the reference's own code:
  * libs/renders/BaseRender.py  Renderer.render / batchify_rays / render_rays /
    get_sampling_points / pts_to_can_pts / get_grid_coords / raw2outputs, Projector.*
  * libs/nerfheads/trainhead.py NeRFHead.forward / NeRFSigmaHead.forward /
    NeRFRGBHead.forward / fused_mean_variance
  * libs/nerfheads/networks/SparseConvNet.py SparseConvNet.forward (the
    F.grid_sample + cat + view lines :105-124)
  * libs/datasets/data_utils.py get_rays / get_near_far
  * libs/renders/demo_render.py Renderer.render / batchify_rays / render_rays / get_sampling_points /
    pts_to_can_pts / get_grid_coords, Projector.* ; libs/nerfheads/trainhead.py NeRFSigmaHead.test_forward ;
    libs/nerfheads/networks/SparseConvNet.py SparseConvNet.encode  (demo_* vectors: the progressive renderer)
  * libs/encoders/UNet.py ResUNet.forward (encoder_* vectors only)
  * libs/nerfheads/networks/MultiHeadAttention.py MultiHeadAttention.forward (attention_* vectors only)

What is NOT the reference: `spconv` (v1.2.1, not in the tree, not installed) is
replaced by inert stand-ins so the modules import and construct; the sparse
convolution stages are replaced by objects whose .dense() returns the synthetic
dense volumes, i.e. the 4 feature levels are INPUTS of every vector here.  The
encoder is likewise replaced by a module returning the synthetic featmaps.
cv2/mcubes/trimesh are empty stubs (import-time only); np.int is aliased for
numpy>=1.24.  Parity of the volume *builder* stays unpinned.

demo_render.py names its device by the literal "cuda" (`.to("cuda")`, `torch.cuda.synchronize()`); this container has
no GPU, so `_device_shim()` maps that name to "cpu" and makes the synchronize a no-op for the duration of a demo_* case.
Nothing else of that file is touched: its arithmetic runs as written, on CPU tensors.

Inputs are never stored: they are regenerated from (seed, config) by
gp-nerf_amd/synthetic.py; each .npz carries a SHA-256 over the input bytes.
"""
import hashlib
import importlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _install_stubs():
    for name in ("mcubes", "trimesh", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sp = types.ModuleType("spconv")

    class SparseSequential(nn.Sequential):
        pass

    class _InertConv(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    class SparseConvTensor:
        def __init__(self, features, indices, spatial_shape, batch_size):
            self.features, self.indices = features, indices
            self.spatial_shape, self.batch_size = spatial_shape, batch_size

    sp.SparseSequential = SparseSequential
    sp.SubMConv3d = _InertConv
    sp.SparseConv3d = _InertConv
    sp.SparseConvTensor = SparseConvTensor
    sys.modules["spconv"] = sp
    if not hasattr(np, "int"):
        np.int = int  # data_utils.py:123,126 predate numpy 1.24


def _paths():
    for sub in ("", "libs/datasets", "libs/encoders", "libs/nerfheads", "libs/nerfheads/networks", "libs/renders"):
        p = os.path.join(REF, sub)
        if p not in sys.path:
            sys.path.insert(0, p)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)


class _Dense:
    def __init__(self, vol):
        self.vol = vol

    def dense(self):
        return self.vol


class _Level(nn.Module):
    """Stands where a double_conv stage stood; hands back the synthetic level."""

    def __init__(self, vol):
        super().__init__()
        self.vol = vol

    def forward(self, x):
        return _Dense(self.vol)


class _Pass(nn.Module):
    def forward(self, x):
        return x


class _FixedEncoder(nn.Module):
    def __init__(self, featmaps):
        super().__init__()
        self.featmaps = featmaps

    def forward(self, x):
        return self.featmaps


def sha_inputs(scene):
    h = hashlib.sha256()
    for k in ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "bounds",
              "out_sh", "Rh", "Th", "featmaps"):
        h.update(np.ascontiguousarray(scene[k]).tobytes())
    for v in scene["volumes"]:
        h.update(np.ascontiguousarray(v).tobytes())
    for k, v in scene["head"].items():
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def build_reference_renderer(scene, n_samples, neg_ray):
    BaseRender = importlib.import_module("BaseRender")
    trainhead = importlib.import_module("trainhead")
    head = trainhead.NeRFHead(in_feat_ch=32, n_smpl=6890, code_dim=32, attn_n_heads=4,
                              spconv_n_layers=4, spconv_out_dim=[32, 32, 32, 32], use_rgbhead=True)
    sd = head.state_dict()
    for k, v in scene["head"].items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v.copy())
    head.load_state_dict(sd, strict=True)
    vols = [torch.from_numpy(v) for v in scene["volumes"]]
    net = [_Pass()]
    for v in vols:
        net += [_Pass(), _Level(v)]
    head.sigmahead.xyzc_net.net = nn.ModuleList(net)
    enc = _FixedEncoder(torch.from_numpy(scene["featmaps"]))
    r = BaseRender.Renderer(enc, head, is_train=False, neg_ray_train=neg_ray, neg_ray_val=neg_ray,
                            n_rays=1024, n_samples=n_samples,
                            voxel_size=[float(x) for x in scene["voxel_size"]], chunk=400)
    r.eval()
    return r, BaseRender, trainhead


class _float_is_double:
    """`Tensor.float()` -> `.double()` while the reference's head runs in float64 (trainhead.py:55 casts the grid with .float())."""

    def __enter__(self):
        self._f = torch.Tensor.float
        torch.Tensor.float = lambda t, *a, **k: t.double()

    def __exit__(self, *exc):
        torch.Tensor.float = self._f


class _Head64(nn.Module):
    """The reference's NeRFHead evaluated in float64 on the float32 renderer's own inputs: sample positions, projections, masks
    and gathered view features stay the float32 run's bit for bit (so no in-bounds test flips), the trilinear volume
    interpolation, every dense layer and -- behind it -- raw2outputs run in double.  The distance between this run and the plain
    float32 one is what the reference's OWN float32 rounding in the head does to each map on these parameters."""

    def __init__(self, head):
        super().__init__()
        self.head = head.double()
        for m in self.head.sigmahead.xyzc_net.net:
            if hasattr(m, "vol"):
                m.vol = m.vol.double()
        self.use_rgbhead = head.use_rgbhead

    def forward(self, sp_input, grid_coords, smpl_feat, rgb_feat, mask):
        with _float_is_double():
            return self.head(sp_input, grid_coords.double(), smpl_feat.double(), rgb_feat.double(), mask.double())


def head_f64_spread(scene, n_samples, neg_ray, chunk, ret32):
    """max |float32 run - float64-head run| per map over ALL rays (see _Head64), and the float64-head maps themselves"""
    r, _, _ = build_reference_renderer(scene, n_samples, neg_ray)
    r.chunk = chunk
    r.nerfhead = _Head64(r.nerfhead)
    with torch.no_grad():
        ret64 = r.render(to_batch(scene))
    assert ret64["rgb_map"].dtype == torch.float64
    keys = ("rgb_map", "depth_map", "acc_map", "rgb_in_map")
    return {k: float((ret32[k].double() - ret64[k]).abs().max()) for k in keys}, ret64


def to_batch(scene):
    keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "target_K", "target_pose",
            "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk")
    return {k: torch.from_numpy(np.ascontiguousarray(scene[k])) for k in keys}


def run_case(name, scene_kw, n_samples, neg_ray=False, stretch=None, stages_rays=32, outputs_only=False, ray_stride=None, chunk=400, spread=False):
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    scene = syn.make_scene(**scene_kw)
    if stretch is not None:
        # push samples outside the volume / the source images: zero-padding paths
        mid = 0.5 * (scene["near"] + scene["far"])
        half = 0.5 * (scene["far"] - scene["near"])
        scene["near"] = (mid - stretch * half).astype(np.float32)
        scene["far"] = (mid + stretch * half).astype(np.float32)
    r, BaseRender, trainhead = build_reference_renderer(scene, n_samples, neg_ray)
    r.chunk = chunk
    batch = to_batch(scene)
    with torch.no_grad():
        ret = r.render(batch)
    out = {
        "rgb_map": ret["rgb_map"][0].numpy(),
        "depth_map": ret["depth_map"][0, :, 0].numpy(),
        "acc_map": ret["acc_map"][0, :, 0].numpy(),
        "disp_map": ret["disp_map"][0, :, 0].numpy(),
        "rgb_in_map": ret["rgb_in_map"][0].numpy(),
    }
    if spread:
        sp, ret64 = head_f64_spread(scene, n_samples, neg_ray, chunk, ret)
        for k, v in sp.items():
            out["spread_" + k] = np.float64(v)
        for k in ("rgb_map", "depth_map", "acc_map"):
            out[k + "_head64"] = ret64[k][0].numpy().reshape(out[k].shape)          # float64
        print("   float32-vs-float64 head spread of the reference:", sp)
    if not outputs_only:
        out["weights"] = ret["alpha"][0].numpy()
        out["z_vals"] = ret["z_vals"][0].numpy()
        # stage-level vectors through the reference's own functions, for `stages_rays` rays spread evenly over the ray list
        # (`st_rays`); the two wide per-sample arrays (128 volume features, 3 x 35 view features) for every 4th of those (`st_heavy`)
        with torch.no_grad():
            n_all = batch["ray_o"].shape[1]
            idx = np.unique(np.linspace(0, n_all - 1, min(stages_rays, n_all)).round().astype(np.int64))
            heavy = np.arange(0, idx.size, 4)
            ti = torch.from_numpy(idx)
            rays_o, rays_d = batch["ray_o"][:, ti], batch["ray_d"][:, ti]
            pts, z = r.get_sampling_points(rays_o, rays_d, batch["near"][:, ti], batch["far"][:, ti])
            pts_smpl = r.pts_to_can_pts(pts.float(), batch)
            sp_input = r.prepare_sp_input(batch)
            grid = r.get_grid_coords(pts_smpl, sp_input, batch).view(1, -1, 3)
            src_imgs = batch["src_imgs"] * 0.5 + 0.5
            V = 3
            H, W = src_imgs.shape[-2:]
            cams = torch.ones((1, V, 34))
            cams[:, :, 0], cams[:, :, 1] = H, W
            Kh = torch.eye(4)[None, None].repeat(1, V, 1, 1)
            Kh[:, :, :3, :3] = batch["src_Ks"]
            Ph = torch.eye(4)[None, None].repeat(1, V, 1, 1)
            Ph[:, :, :3, :4] = batch["src_poses"]
            cams[:, :, 2:18] = Kh.reshape(1, V, -1)
            cams[:, :, -16:] = Ph.reshape(1, V, -1)
            xyz = batch["feature"][..., :3]
            smpl_xyz = torch.bmm(xyz, batch["Rh"].transpose(1, 2)) + batch["Th"]
            proj = BaseRender.Projector("cpu", neg_ray=neg_ray)
            rgb_feat, smpl_feat, mask = proj.compute(pts.squeeze(0), smpl_xyz, src_imgs, cams,
                                                     featmaps=torch.from_numpy(scene["featmaps"]))
            raw, rgb_in = r.nerfhead(sp_input, grid, smpl_feat, rgb_feat, mask)
            pixel_mask = mask[..., 0].sum(dim=2) > 1
            _, _, _, _, _, ray_mask, alpha = r.raw2outputs(raw, z.squeeze(0), pixel_mask, neg=neg_ray)
            vol_feat = r.nerfhead.sigmahead.xyzc_net(None, grid[:, None, None].float())  # [1,128,P]
        out.update({
            "st_pts": pts[0].numpy(), "st_z": z[0].numpy(), "st_pts_smpl": pts_smpl[0].numpy(),
            "st_grid": grid[0].numpy(), "st_rgb_feat": rgb_feat.numpy()[heavy], "st_mask": mask[..., 0].numpy(),
            "st_raw": raw.numpy(), "st_rgb_in": rgb_in.numpy(),
            "st_vol_feat": vol_feat[0].t().contiguous().numpy().reshape(idx.size, n_samples, 128)[heavy].reshape(-1, 128),
            "st_ray_mask": ray_mask.numpy(), "st_alpha": alpha.numpy(), "st_rays": idx, "st_heavy": idx[heavy],
        })
    if ray_stride is not None:
        # BASELINE.json's full-size configurations: every `ray_stride`-th ray of the reference's maps + a SHA-256 over all of them
        assert outputs_only
        full = hashlib.sha256()
        for k in ("rgb_map", "depth_map", "acc_map", "disp_map", "rgb_in_map"):
            full.update(np.ascontiguousarray(out[k]).tobytes())
        out = {k: np.ascontiguousarray(v[::ray_stride]) for k, v in out.items()}
        out["ray_stride"] = np.int64(ray_stride)
        out["outputs_sha256"] = np.frombuffer(full.digest(), np.uint8)
    meta = {"scene_kw": scene_kw, "n_samples": n_samples, "neg_ray": bool(neg_ray), "stretch": stretch,
            "n_rays": int(scene["ray_o"].shape[1]), "sha256_inputs": sha_inputs(scene),
            "torch": torch.__version__, "numpy": np.__version__}
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={meta['n_rays']} S={n_samples} rgb mean={out['rgb_map'].mean():.4f} "
          f"acc mean={out['acc_map'].mean():.4f} depth mean={out['depth_map'].mean():.4f} -> {os.path.getsize(path)} B")


def run_rays_case(name, H, W, kind):
    """get_rays + get_near_far (data_utils.py:47-63,96-130) as sample_ray's test branch chains them (:294-300): float64
    camera (the dataset's dtype) -> float64 rays rounded to float32 -> get_near_far on the float32 rays."""
    du = importlib.import_module("data_utils")
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    K, R, T, bounds = syn.make_ray_camera(kind, H, W)
    ray_o, ray_d = du.get_rays(H, W, K, R, T)
    assert ray_d.dtype == np.float64
    ray_o = ray_o.reshape(-1, 3).astype(np.float32)
    ray_d = ray_d.reshape(-1, 3).astype(np.float32)
    n_clamped = int((np.abs(ray_d) < 1e-5).sum())
    near, far, mask_at_box = du.get_near_far(bounds, ray_o, ray_d)            # clamps ray_d in place (:101)
    assert near.dtype == np.float64
    ray_o, ray_d = ray_o[mask_at_box], ray_d[mask_at_box]
    near, far = near.astype(np.float32), far.astype(np.float32)
    step = max(1, ray_d.shape[0] // 4096)
    out = {"K": K, "R": R, "T": T, "bounds": bounds, "H": np.int32(H), "W": np.int32(W), "kind": np.frombuffer(kind.encode(), np.uint8),
           "mask_at_box_bits": np.packbits(mask_at_box), "n_rays": np.int64(mask_at_box.sum()),
           "near": near, "far": far, "ray_o": ray_o[0], "ray_d_stride": np.int64(step), "ray_d_sub": ray_d[::step],
           "ray_d_sha256": np.frombuffer(hashlib.sha256(np.ascontiguousarray(ray_d).tobytes()).digest(), np.uint8),
           "n_clamped": np.int64(n_clamped), "n_degenerate": np.int64((near == far).sum())}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {int(mask_at_box.sum())}/{H*W} rays hit, {n_clamped} clamped components, {int((near == far).sum())} rays with near == far "
          f"-> {os.path.getsize(path)} B")

class _device_shim:
    """`.to("cuda")` -> `.to("cpu")` and a no-op torch.cuda.synchronize while the reference's demo renderer runs."""

    def __enter__(self):
        self._to, self._sync = torch.Tensor.to, torch.cuda.synchronize
        orig = self._to

        def to(t, *args, **kw):
            args = tuple("cpu" if isinstance(a, str) and a.startswith("cuda") else a for a in args)
            if isinstance(kw.get("device"), str) and kw["device"].startswith("cuda"):
                kw["device"] = "cpu"
            return orig(t, *args, **kw)

        torch.Tensor.to = to
        torch.cuda.synchronize = lambda *a, **k: None
        return self

    def __exit__(self, *exc):
        torch.Tensor.to, torch.cuda.synchronize = self._to, self._sync


def run_demo_case(name, scene_kw, n_samples, neg_ray=False, probe=96):
    """The progressive renderer, libs/renders/demo_render.py Renderer.render (:429-498 -> batchify_rays :378-392 ->
    render_rays :96-376), on a synthetic frame whose 4 dense levels are sparse and non-negative (what the ReLU-terminated
    sparse conv net produces).  Stored: what render() returns (rgb_map, mask_at_box; pred_img is checked to be their
    scatter and not stored), plus values the run passes between the reference's own functions, captured by wrapping the
    callee: masks3d / mask_xyz of SparseConvNet.encode, the rays handed to get_sampling_points, and the first `probe`
    points through NeRFSigmaHead.test_forward."""
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    demo = importlib.import_module("demo_render")
    trainhead = importlib.import_module("trainhead")
    scene = syn.make_scene(**scene_kw)
    head = trainhead.NeRFHead(in_feat_ch=32, n_smpl=6890, code_dim=32, attn_n_heads=4,
                              spconv_n_layers=4, spconv_out_dim=[32, 32, 32, 32], use_rgbhead=True)
    sd = head.state_dict()
    for k, v in scene["head"].items():
        sd[k] = torch.from_numpy(v.copy())
    head.load_state_dict(sd, strict=True)
    net = [_Pass()]
    for v in scene["volumes"]:
        net += [_Pass(), _Level(torch.from_numpy(v))]
    head.sigmahead.xyzc_net.net = nn.ModuleList(net)
    enc = _FixedEncoder(torch.from_numpy(scene["featmaps"]))
    # neg_ray_val is the one batchify_rays picks here: body_msk is wider than n_rays (demo_render.py:380-384)
    r = demo.Renderer(enc, head, is_train=False, neg_ray_train=not neg_ray, neg_ray_val=neg_ray, n_rays=1024,
                      n_samples=n_samples, voxel_size=[float(x) for x in scene["voxel_size"]], chunk=400)
    r.eval()
    batch = to_batch(scene)
    batch["target_K_inv"] = torch.from_numpy(scene["target_K_inv"].copy())
    batch["body_msk"] = torch.ones((1, 2048))
    cap = {}
    gsp = r.get_sampling_points

    def get_sampling_points(ray_o, ray_d, near, far, perturb=1):
        cap.update(ray_o=ray_o[0].numpy().copy(), ray_d=ray_d[0].numpy().copy(), near=near[0].numpy().copy(), far=far[0].numpy().copy())
        return gsp(ray_o, ray_d, near, far, perturb)

    r.get_sampling_points = get_sampling_points
    tf = head.sigmahead.test_forward

    def test_forward(sp_input, grid_coords, rgb_feat, mask):
        out = tf(sp_input, grid_coords, rgb_feat, mask)
        k = min(probe, rgb_feat.shape[0])
        cap.update(tf_grid=grid_coords[0, :k].numpy().copy(), tf_rgb_feat=rgb_feat[:k, 0].numpy().copy(),
                   tf_mask=mask[:k, 0, :, 0].numpy().copy(), tf_sigma_feat=out[0][:k, 0].numpy().copy(),
                   tf_globalfeat=out[1][:k, 0, 0].numpy().copy(), n_kept=np.int64(rgb_feat.shape[0]))
        return out

    head.sigmahead.test_forward = test_forward
    with torch.no_grad(), _device_shim():
        ret = r.render(batch)
    mask = np.asarray(ret["mask_at_box"]).astype(bool)
    pred = np.zeros((512, 512, 3))
    pred[mask.reshape(512, 512)] = ret["rgb_map"]
    assert ret["pred_img"].dtype == np.float64 and np.array_equal(pred, ret["pred_img"])
    assert set(ret) == {"rgb_map", "pred_img", "mask_at_box", "time_slots", "etime", "rtime"}, sorted(ret)
    xn = head.sigmahead.xyzc_net
    out = {"rgb_map": ret["rgb_map"].astype(np.float32), "mask_at_box_bits": np.packbits(mask),
           "masks3d": xn.masks3d.numpy().astype(np.float32), "n_mask_xyz": np.int64(xn.mask_xyz.shape[0]),
           "mask_xyz_head": xn.mask_xyz[:64].numpy().astype(np.float32), "target_K_inv": scene["target_K_inv"],
           "time_slot_keys": np.frombuffer(json.dumps(sorted(ret["time_slots"])).encode(), dtype=np.uint8)}
    out.update(cap)
    meta = {"scene_kw": scene_kw, "n_samples": n_samples, "neg_ray": bool(neg_ray), "n_rays": int(mask.sum()),
            "sha256_inputs": sha_inputs(scene), "torch": torch.__version__, "numpy": np.__version__,
            "reference": "libs/renders/demo_render.py Renderer.render, eval, CPU fp32 via the device-name shim"}
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {meta['n_rays']} rays, {int(cap['n_kept'])}/{meta['n_rays'] * n_samples} samples kept, "
          f"rgb mean={out['rgb_map'].mean():.4f} max={out['rgb_map'].max():.4f} -> {os.path.getsize(path)} B")


def run_e2e_case(name, scene_kw, n_samples, seed, ray_stride=None, chunk=400, spread=False, weights_kw=None):
    """The evaluation loop's per-frame chain with the reference's REAL image encoder (BASELINE.json configs[4] in miniature;
    the ZJU-MoCap data itself is not in the tree): libs/encoders/UNet.py ResUNet.forward -> libs/renders/BaseRender.py
    Renderer.render, then libs/evaluators/if_nerf.py Evaluator.psnr_metric on the result against a seeded ground truth.
    Source images are the structured encoder images; the 4 dense levels stay inputs (spconv is not available).

    `ray_stride` (the config-5-sized case: 512x512 sources, 128x128 feature maps, ~75 k rays x 64 samples): the maps are
    stored for every `ray_stride`-th ray plus a SHA-256 over the full arrays (so that a regeneration is checked bit for bit),
    the feature maps for every 4th texel plus per-(view, channel) float64 means, and the ground truth of the PSNR as the
    8-bit image a dataset would hold (clip(render + noise) rounded to 1/255) for ALL rays, so the evaluator's PSNR over the
    whole frame is pinned to the reference evaluator's value.  The same case is also run with the reference's encoder switched
    to float64 (`ResUNet.double()`, its output rounded to float32 before the renderer, everything else unchanged): the distance
    between the two runs (`spread_*`, max over ALL rays, and `*_enc64` for the stored rays) is what the reference's own float32
    rounding inside the encoder does to each map at this size -- the yardstick for any other float32 encoder."""
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    UNet = importlib.import_module("UNet")
    sk = types.ModuleType("skimage"); skm = types.ModuleType("skimage.measure"); skm.compare_ssim = None
    sys.modules.setdefault("skimage", sk); sys.modules.setdefault("skimage.measure", skm)
    if_nerf = importlib.import_module("libs.evaluators.if_nerf")
    scene = syn.make_scene(**scene_kw)
    scene["src_imgs"] = syn.make_encoder_images(scene_kw["H"], scene_kw["W"], seed)[None]
    r, BaseRender, trainhead = build_reference_renderer(scene, n_samples, False)
    r.chunk = chunk
    enc = UNet.ResUNet(encoder="resnet34", out_ch=32)
    enc_state = syn.make_encoder_weights(seed, **(weights_kw or {}))
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in enc_state.items()}, strict=True)
    r.encoder = enc.eval()
    batch = to_batch(scene)
    with torch.no_grad():
        ret = r.render(batch)
        featmaps = enc(batch["src_imgs"][0]).numpy()
    rgb = ret["rgb_map"][0].numpy()
    g = np.random.Generator(np.random.PCG64([seed, 909]))
    rgb_gt = np.clip(rgb + 0.05 * g.standard_normal(rgb.shape, dtype=np.float32), 0, 1).astype(np.float32)
    if ray_stride is not None:
        gt_u8 = np.round(rgb_gt * np.float32(255.0)).astype(np.uint8)
        rgb_gt = gt_u8.astype(np.float32) / np.float32(255.0)
    ev = if_nerf.Evaluator(None, "seq")
    psnr = float(ev.psnr_metric(rgb, rgb_gt))
    h = hashlib.sha256()
    h.update(sha_inputs(scene).encode())
    for k in sorted(enc_state):
        h.update(np.ascontiguousarray(enc_state[k]).tobytes())
    out = {"rgb_map": rgb, "depth_map": ret["depth_map"][0, :, 0].numpy(), "acc_map": ret["acc_map"][0, :, 0].numpy(),
           "rgb_in_map": ret["rgb_in_map"][0].numpy(), "featmaps": featmaps.astype(np.float32), "rgb_gt": rgb_gt,
           "psnr": np.float64(psnr), "mse": np.float64(np.mean((rgb - rgb_gt) ** 2))}
    if weights_kw:
        # "trained-like" case: the head's own float32 noise on the reference's feature maps (the yardstick of the identical-inputs leg)
        sc_h = dict(scene, featmaps=featmaps.astype(np.float32))
        with torch.no_grad():
            r.encoder = _FixedEncoder(torch.from_numpy(sc_h["featmaps"]))
            ret_fixed = r.render(batch)
        r.encoder = enc
        assert torch.equal(ret_fixed["rgb_map"], ret["rgb_map"])
        sp, _ = head_f64_spread(sc_h, n_samples, False, chunk, ret)
        for k, v in sp.items():
            out["spread_head_" + k] = np.float64(v)
        print("   float32-vs-float64 head spread of the reference:", sp)
    if ray_stride is not None or spread:
        import copy
        rs = ray_stride or 1
        with torch.no_grad():
            fm64 = copy.deepcopy(enc).double()(batch["src_imgs"][0].double()).float()
            r.encoder = _FixedEncoder(fm64)
            ret64 = r.render(batch)
        r.encoder = enc
        pairs = (("rgb_map", ret64["rgb_map"][0].numpy()), ("depth_map", ret64["depth_map"][0, :, 0].numpy()),
                 ("acc_map", ret64["acc_map"][0, :, 0].numpy()))
        for k, v in pairs:
            out["spread_" + k] = np.float64(np.abs(v.astype(np.float64) - out[k]).max())
            out[k + "_enc64"] = v[::rs]
        out["spread_featmaps"] = np.float64(np.abs(fm64.numpy().astype(np.float64) - featmaps).max())
        print("   float32-vs-float64 encoder spread of the reference:", {k: float(v) for k, v in out.items() if k.startswith("spread_")})
    if ray_stride is not None:
        full = hashlib.sha256()
        for k in ("rgb_map", "depth_map", "acc_map", "rgb_in_map", "featmaps"):
            full.update(np.ascontiguousarray(out[k]).tobytes())
        fm = out.pop("featmaps")
        out = {k: (v[::ray_stride] if k in ("rgb_map", "depth_map", "acc_map", "rgb_in_map") else v) for k, v in out.items() if k != "rgb_gt"}
        print("   float32-vs-float64 encoder spread of the reference:", {k: float(v) for k, v in out.items() if k.startswith("spread_")})
        out.update(rgb_gt_u8=gt_u8, ray_stride=np.int64(ray_stride), featmaps_sub=np.ascontiguousarray(fm[:, :, ::4, ::4]),
                   featmaps_stride=np.int64(4), featmaps_chan_mean=fm.astype(np.float64).mean(axis=(2, 3)),
                   featmaps_absmax=np.float64(np.abs(fm).max()), outputs_sha256=np.frombuffer(full.digest(), np.uint8))
    meta = {"scene_kw": scene_kw, "n_samples": n_samples, "seed": seed, "n_rays": int(rgb.shape[0]), "sha256_inputs": h.hexdigest(),
            "torch": torch.__version__, "reference": "UNet.ResUNet.forward -> BaseRender.Renderer.render -> if_nerf.Evaluator.psnr_metric, eval, CPU fp32"}
    if weights_kw:
        meta["weights_kw"] = weights_kw
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={meta['n_rays']} rgb mean={rgb.mean():.4f} acc mean={out['acc_map'].mean():.4f} psnr={psnr:.4f} -> {os.path.getsize(path)} B")


def run_loop_case(name, frames, n_samples):
    """The evaluation LOOP of BASELINE.json configs[4] (libs/trainers/BaseTrainer.py:255-280 Trainer.evaluate -> per frame
    render.module.render(batch) -> libs/evaluators/if_nerf.py:49-66 Evaluator.evaluate, `total_time += ret["rtime"]`, then
    Evaluator.summarize :68-83), run as the README's inference command runs it: with the progressive renderer
    (`render.file demo_render` -- the dense BaseRender.Renderer returns no "rtime" and the loop would stop at its first frame),
    over `frames` synthetic frames that share one head.  Two passes: the first renders each frame once to derive an 8-bit ground
    truth (clip(pred + noise) rounded to 1/255, as a dataset would hold it) so that the PSNRs are informative; the second IS the
    reference's loop.  Stored per frame: ground truth, dataset mask, MSE and PSNR as the reference's evaluator computed them
    (captured from its lists inside summarize()), every 8th predicted pixel; plus the summary means, the frame count and the keys
    of what render() returned.  Import-time stand-ins: tensorboardX / torchvision / termcolor (never called), cv2.boundingRect and
    skimage's compare_ssim (inert: SSIM stays UNPINNED, its value is not stored)."""
    import tempfile
    from types import SimpleNamespace as NS
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    for mod, attrs in (("tensorboardX", dict(SummaryWriter=object)), ("torchvision", dict(__version__="0.15.0")),
                       ("termcolor", dict(colored=lambda s, *a, **k: s))):
        m = sys.modules.setdefault(mod, types.ModuleType(mod))
        for k, v in attrs.items():
            setattr(m, k, v)
    sk = sys.modules.setdefault("skimage", types.ModuleType("skimage"))
    skm = sys.modules.setdefault("skimage.measure", types.ModuleType("skimage.measure"))
    skm.compare_ssim = lambda *a, **k: float("nan")
    sk.measure = skm
    cv2 = sys.modules["cv2"]
    cv2.boundingRect = lambda m: (0, 0, int(m.shape[1]), int(m.shape[0]))
    if_nerf = importlib.import_module("libs.evaluators.if_nerf")
    if_nerf.compare_ssim = skm.compare_ssim
    BaseTrainer = importlib.import_module("libs.trainers.BaseTrainer")
    demo = importlib.import_module("demo_render")
    trainhead = importlib.import_module("trainhead")
    scenes = [syn.make_scene(**kw) for kw in frames]
    head = trainhead.NeRFHead(in_feat_ch=32, n_smpl=6890, code_dim=32, attn_n_heads=4, spconv_n_layers=4, spconv_out_dim=[32, 32, 32, 32],
                              use_rgbhead=True)
    sd = head.state_dict()
    for k, v in scenes[0]["head"].items():                      # ONE model for the whole loop: the first frame's head
        sd[k] = torch.from_numpy(v.copy())
    head.load_state_dict(sd, strict=True)
    levels = [_Level(None) for _ in range(4)]
    net = [_Pass()]
    for lv in levels:
        net += [_Pass(), lv]
    head.sigmahead.xyzc_net.net = nn.ModuleList(net)
    enc = _FixedEncoder(None)
    r = demo.Renderer(enc, head, is_train=False, neg_ray_train=True, neg_ray_val=False, n_rays=1024, n_samples=n_samples,
                      voxel_size=[float(x) for x in scenes[0]["voxel_size"]], chunk=400)
    rets = []

    class _Module:
        """what `self.render.module` is to the loop; switches the per-frame products (feature maps, dense levels) by frame_index"""

        def render(self, batch):
            i = int(batch["frame_index"])
            enc.featmaps = torch.from_numpy(scenes[i]["featmaps"])
            for lv, v in zip(levels, scenes[i]["volumes"]):
                lv.vol = torch.from_numpy(v)
            with _device_shim():
                ret = r.render(batch)
            rets.append(ret)
            return ret

    model = NS(module=_Module(), eval=lambda: r.eval(), train=lambda: None)

    def batch_of(i):
        b = to_batch(scenes[i])
        b["target_K_inv"] = torch.from_numpy(scenes[i]["target_K_inv"].copy())
        b["body_msk"] = torch.ones((1, 2048))
        b["mask_at_box"] = torch.from_numpy(scenes[i]["mask_at_box"].copy())
        b["frame_index"] = torch.tensor([i])
        return b

    # pass 1: a ground truth per frame from the reference's own render
    gts, masks = [], []
    with torch.no_grad():
        for i in range(len(scenes)):
            b = batch_of(i)
            ret = model.module.render(b)
            m = scenes[i]["mask_at_box"][0].reshape(512, 512)
            pred = ret["pred_img"][m]
            g = np.random.Generator(np.random.PCG64([int(frames[i]["seed"]), 911]))
            gts.append(np.round(np.clip(pred + 0.05 * g.standard_normal(pred.shape), 0, 1) * 255.0).astype(np.uint8))
            masks.append(m)
    first = rets[:]
    del rets[:]
    # pass 2: the reference's loop
    cap = {}
    summarize = if_nerf.Evaluator.summarize

    def summarize_and_capture(self):
        cap.update(mse=[float(v) for v in self.mse], psnr=[float(v) for v in self.psnr])
        cap["summary"] = summarize(self)
        return cap["summary"]

    if_nerf.Evaluator.summarize = summarize_and_capture
    BaseTrainer.Evaluator = if_nerf.Evaluator
    tmp = tempfile.mkdtemp(prefix="gpnerf_loop_")
    cfg = NS(dataset=NS(H=1024, W=1024, ratio=0.5), test=NS(test_seq="loop", save_imgs=False), head=NS(rgb=NS(use_rgbhead=True)),
             result_dir=tmp, train=NS(max_epoch=1), render=NS(file="demo_render"), output_dir="")
    loader = []
    for i in range(len(scenes)):
        b = batch_of(i)
        b["rgb"] = torch.from_numpy(gts[i].astype(np.float32) / np.float32(255.0))[None]
        loader.append(b)
    try:
        t = BaseTrainer.Trainer(cfg, model, criterion=None, optimizer=None, lr_scheduler=None, logger=None, log_dir=None,
                                performance_indicator="psnr", last_iter=None, rank=0, device="cpu")
        t.evaluate(loader, os.path.join(tmp, "loop"), is_vis=False)
    finally:
        if_nerf.Evaluator.summarize = summarize
    assert len(rets) == len(scenes) and len(cap["psnr"]) == len(scenes)
    saved = np.load(os.path.join(tmp, "loop", "metrics.npy"))             # if_nerf.py:72-80 keeps the per-frame MSE list
    assert np.allclose(saved, cap["mse"])
    out = {"count": np.int64(len(rets)), "mse": np.array(cap["mse"], np.float64), "psnr": np.array(cap["psnr"], np.float64),
           "summary_mse": np.float64(cap["summary"]["mse"]), "summary_psnr": np.float64(cap["summary"]["psnr"]),
           "ret_keys": np.frombuffer(json.dumps(sorted(rets[0])).encode(), dtype=np.uint8)}
    h = hashlib.sha256()
    for i, (ret, ret1) in enumerate(zip(rets, first)):
        assert np.array_equal(ret["pred_img"], ret1["pred_img"]), "the reference's two passes differ"
        assert isinstance(ret["rtime"], float) and isinstance(ret["etime"], float)
        pred = ret["pred_img"][masks[i]].astype(np.float32)
        out[f"gt_u8_{i}"] = gts[i]
        out[f"mask_at_box_bits_{i}"] = np.packbits(masks[i])
        out[f"pred_sub_{i}"] = np.ascontiguousarray(pred[::8])
        out[f"sel_mask_bits_{i}"] = np.packbits(np.asarray(ret["mask_at_box"]).astype(bool))
        h.update(sha_inputs(scenes[i]).encode())
    meta = {"frames": frames, "n_samples": n_samples, "sha256_inputs": h.hexdigest(), "torch": torch.__version__,
            "reference": "BaseTrainer.Trainer.evaluate -> demo_render.Renderer.render -> if_nerf.Evaluator.evaluate / summarize, eval, CPU fp32 "
                         "via the device-name shim; SSIM not computed (scikit-image absent): unpinned"}
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(rets)} frames, psnr {cap['psnr']}, summary {cap['summary']['psnr']:.5f} -> {os.path.getsize(path)} B")


DEMO_SMALL = dict(H=512, W=512, aabb_half=(0.12, 0.16, 0.05), voxel=0.005, bias_std=0.1, pose="random")
DEMO_CASES = [
    ("demo_zju_s32", dict(seed=21, focal_mul=1.6, vol_occupancy=0.3, sigma_bias=0.5, **DEMO_SMALL), 32, {}),
    ("demo_neg_s96", dict(seed=22, focal_mul=1.3, vol_occupancy=0.6, sigma_bias=0.5, neg_cams=True, neg_target=True, **DEMO_SMALL), 96,
     dict(neg_ray=True)),
    ("demo_dense_s16", dict(seed=23, focal_mul=1.0, vol_occupancy=0.9, sigma_bias=2.0, **DEMO_SMALL), 16, {}),
    # the progressive renderer on "trained-like" parameters (round 4): head x 2 with biases, sparse non-negative levels and feature
    # maps x 4 with log-normal tails -- the distribution of trained_h2_s64, whose float32-vs-float64 yardstick the tests borrow
    ("demo_trained_s32", dict(seed=24, focal_mul=1.5, vol_occupancy=0.4, sigma_bias=-4.0, head_scale=2.0, feat_scale=4.0, feat_tail=0.5,
                              vol_scale=4.0, **dict(DEMO_SMALL, bias_std=0.3)), 32, {}),
    # a PERSON-SHAPED frame at full size (round 5): vertices on a capsule-limbed figure's surface
    # (synthetic.body_vertices), the dense levels non-negative on exactly the voxels the sparse pyramid writes for them, the
    # full-size SMPL box, SURVEY.md 8d's f = 1.05 W camera, 64 samples -- occupancy and cull rate of a body, not of random blocks
    ("demo_body_s64", dict(H=512, W=512, seed=25, focal_mul=1.05, body="capsules", sigma_bias=0.5, bias_std=0.1, pose="random"), 64, {}),
]


SMALL = dict(aabb_half=(0.12, 0.16, 0.05), voxel=0.005, bias_std=0.1, sigma_bias=0.0)

CASES = [
    # name, scene kwargs, S, extras
    ("base_s32", dict(H=16, W=16, seed=1, fill="full", pose="random", **SMALL), 32, {}),
    ("base_s64", dict(H=16, W=16, seed=2, fill="full", pose="random", **SMALL), 64, {}),
    ("base_s8", dict(H=16, W=16, seed=3, fill="full", pose="identity", **SMALL), 8, {}),
    ("neg_s32", dict(H=16, W=16, seed=4, fill="full", pose="random", neg_cams=True, **SMALL), 32,
     dict(neg_ray=True)),
    ("allmasked_s32", dict(H=8, W=8, seed=8, fill="full", pose="random", **SMALL), 32, dict(neg_ray=True)),
    ("partial_s32", dict(H=32, W=32, seed=5, focal_mul=8.0, pose="random", **SMALL), 32, {}),
    ("stretch_s32", dict(H=16, W=16, seed=6, fill="full", pose="random", **SMALL), 32, dict(stretch=2.5)),
    ("nonsquare_s16", dict(H=24, W=40, seed=9, focal_mul=7.0, pose="random", **SMALL), 16, {}),
    ("wide_s16", dict(H=128, W=128, seed=7, focal_mul=0.6, pose="random", **SMALL), 16, {}),
    ("config1_64x64_s32", dict(H=64, W=64, seed=0, fill="full", pose="identity",
                               aabb_half=(0.25, 0.45, 0.125), voxel=0.005), 32, dict(outputs_only=True)),
]

# "Trained-like" parameters and features (VERDICT r3 next #1a): every other fixture runs `weights_init` heads (kaiming-normal, zero
# bias), N(0,1) features and InstanceNorm scales near 1 -- none of which a trained checkpoint has.  Here: head weights x 1.5 / 2 / 3
# with non-zero biases, feature maps and volumes x 4 with log-normal tails, ReLU-sparse dense levels; >= 4 096 rays x 64 samples.
TRAINED = dict(fill="full", pose="random", aabb_half=(0.12, 0.16, 0.05), voxel=0.005, feat_scale=4.0, feat_tail=0.5, vol_scale=4.0, vol_relu=True)
TRAINED_CASES = [
    ("trained_h1_s64", dict(H=64, W=64, seed=46, bias_std=0.1, sigma_bias=-10.0, head_scale=1.0, **TRAINED), 64, dict(outputs_only=True, spread=True)),
    ("trained_h1p5_s64", dict(H=64, W=64, seed=50, bias_std=0.2, sigma_bias=-2.0, head_scale=1.5, **TRAINED), 64, dict(outputs_only=True, spread=True)),
    ("trained_h2_s64", dict(H=64, W=64, seed=51, bias_std=0.3, sigma_bias=-4.0, head_scale=2.0, **TRAINED), 64, dict(spread=True)),
    ("trained_h3_s64", dict(H=72, W=72, seed=52, bias_std=0.5, sigma_bias=-16.0, head_scale=3.0, **TRAINED), 64, dict(outputs_only=True, spread=True)),
]
TRAINED_ENC = dict(gamma_range=(0.5, 6.0), beta_std=0.5)

# BASELINE.json configs[1..3] at FULL size, on the very scenes bench.py and tests/test_gpu_configs.py render: the reference's
# Renderer.render over every ray (test chunk 2000), every 64th / 256th ray stored.  Slow (1 - 5 minutes each on 8 CPU threads).
FULL_CASES = [
    ("config2_512x512_s64", dict(H=512, W=512, seed=0, fill="full", pose="identity"), 64, dict(outputs_only=True, ray_stride=64, chunk=2000)),
    ("config3_512x512_s128", dict(H=512, W=512, seed=0, fill="full", pose="identity", sigma_bias=1.0), 128,
     dict(outputs_only=True, ray_stride=64, chunk=2000)),
    ("config4_1024x1024_s64", dict(H=1024, W=1024, seed=0, fill="full", pose="identity"), 64, dict(outputs_only=True, ray_stride=256, chunk=2000)),
]


def run_encoder_case(name, H, W, seed, stride=None, weights_kw=None):
    """libs/encoders/UNet.py ResUNet.forward on seeded images with seeded parameters.  The parameters
    come from gp-nerf_amd/synthetic.py by state_dict key, and are loaded strict=True into the reference's module, so the
    vector also pins the key/shape map.  `stride` (the 512x512 case, the size BASELINE configs[4] encodes at): every
    `stride`-th texel of the [3,32,H/4,W/4] result is stored, with a SHA-256 over the whole result, per-(view, channel)
    float64 means and mean squares, and the per-channel max-abs."""
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    UNet = importlib.import_module("UNet")
    state = syn.make_encoder_weights(seed, **(weights_kw or {}))
    net = UNet.ResUNet(encoder="resnet34", out_ch=32)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    net.eval()
    imgs = syn.make_encoder_images(H, W, seed)
    with torch.no_grad():
        out = net(torch.from_numpy(imgs)).numpy()
        out64 = None
        if weights_kw:           # the reference's own float32-vs-float64 distance on these parameters: the yardstick for a "trained-like" case
            import copy
            out64 = copy.deepcopy(net).double()(torch.from_numpy(imgs).double()).numpy()
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(imgs).tobytes())
    for k in sorted(state):
        h.update(np.ascontiguousarray(state[k]).tobytes())
    meta = dict(name=name, H=H, W=W, seed=seed, inputs_sha256=h.hexdigest(), torch=torch.__version__,
                reference="libs/encoders/UNet.py ResUNet(resnet34, out_ch=32).forward, eval, CPU fp32")
    if weights_kw:
        meta["weights_kw"] = weights_kw
    out = out.astype(np.float32)
    if stride is None:
        arrs = dict(featmaps=out)
        if out64 is not None:
            arrs["spread_f32_f64"] = np.float64(np.abs(out64 - out.astype(np.float64)).max())
            arrs["featmaps_absmax"] = np.float64(np.abs(out).max())
    else:
        o64 = out.astype(np.float64)
        arrs = dict(featmaps_sub=np.ascontiguousarray(out[:, :, ::stride, ::stride]), featmaps_stride=np.int64(stride),
                    featmaps_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest(), np.uint8),
                    featmaps_chan_mean=o64.mean(axis=(2, 3)), featmaps_chan_meansq=(o64 * o64).mean(axis=(2, 3)),
                    featmaps_chan_absmax=np.abs(out).max(axis=(2, 3)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrs)
    print(name, out.shape, float(np.abs(out).max()))


def run_attention_case(name, n, code_dim, seed):
    """libs/nerfheads/networks/MultiHeadAttention.py as trainhead.py:35-37,48-52 uses it: query = vertex code (length 1),
    keys/values = the vertex's features in the 3 source views, sum=False."""
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    MHA = importlib.import_module("MultiHeadAttention")
    state, code, feat = syn.make_attention_case(n, code_dim, seed)
    m = MHA.MultiHeadAttention(4, code_dim, code_dim // 4, code_dim // 4, kv_dim=32, sum=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(code).unsqueeze(1), torch.from_numpy(feat), torch.from_numpy(feat))[0].squeeze(1).numpy()
    h = hashlib.sha256()
    for a in [code, feat] + [state[k] for k in sorted(state)]:
        h.update(np.ascontiguousarray(a).tobytes())
    meta = dict(name=name, n=n, code_dim=code_dim, seed=seed, inputs_sha256=h.hexdigest(), torch=torch.__version__,
                reference="libs/nerfheads/networks/MultiHeadAttention.py MultiHeadAttention(4, d, d/4, d/4, kv_dim=32, sum=False), eval, CPU fp32")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), out=out.astype(np.float32),
                        meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(name, out.shape, float(np.abs(out).max()))


def main():
    _install_stubs()
    _paths()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    for name, kw, S, extra in CASES + TRAINED_CASES + FULL_CASES:
        if only and name not in only:
            continue
        run_case(name, kw, S, **extra)
    if not only or "e2e_64x64_s32" in only:
        run_e2e_case("e2e_64x64_s32", dict(H=64, W=64, seed=31, fill="full", pose="random", **dict(SMALL, sigma_bias=0.3)), 32, 31, spread=True)
    for name, kw, S, extra in DEMO_CASES:
        if not only or name in only:
            run_demo_case(name, kw, S, **extra)
    for name, H, W, kind in (("rays_48", 48, 48, "oblique"), ("rays_512_axis", 512, 512, "axis"), ("rays_512_edge", 512, 512, "edge"),
                             ("rays_512_oblique", 512, 512, "oblique")):
        if not only or name in only:
            run_rays_case(name, H, W, kind)
    for name, n, d, seed in (("attention_d16", 257, 16, 5), ("attention_d32", 300, 32, 6)):
        if not only or name in only:
            run_attention_case(name, n, d, seed)
    for name, H, W, seed, stride in (("encoder_64x64", 64, 64, 3, None), ("encoder_72x88", 72, 88, 4, None),
                                     ("encoder_512x512", 512, 512, 11, 4)):
        if not only or name in only:
            run_encoder_case(name, H, W, seed, stride)
    if not only or "encoder_trained_96x128" in only:
        run_encoder_case("encoder_trained_96x128", 96, 128, 12, None, weights_kw=TRAINED_ENC)
    if not only or "e2e_trained_64x64_s32" in only:
        run_e2e_case("e2e_trained_64x64_s32", dict(H=64, W=64, seed=35, bias_std=0.3, sigma_bias=-60.0, head_scale=2.0, **TRAINED), 32, 35,
                     spread=True, weights_kw=TRAINED_ENC)
    if not only or "loop_demo_3frames" in only:
        run_loop_case("loop_demo_3frames", [dict(seed=61, focal_mul=1.6, vol_occupancy=0.3, sigma_bias=0.5, **DEMO_SMALL),
                                            dict(seed=62, focal_mul=1.4, vol_occupancy=0.5, sigma_bias=0.5, **DEMO_SMALL),
                                            dict(seed=63, focal_mul=1.8, vol_occupancy=0.4, sigma_bias=1.0, **DEMO_SMALL)], 32)
    if not only or "e2e_512_survey" in only:
        # the size BASELINE.json configs[4] runs at: 512x512 sources (1024 x ratio 0.5), full-size SMPL box, literal f = 1.05 W
        # camera of SURVEY.md 8d, 64 samples per ray (configs/trainzju_valzju.yaml train.n_samples), test chunk 2000
        run_e2e_case("e2e_512_survey", dict(H=512, W=512, seed=33, fill="survey", pose="random", bias_std=0.1, sigma_bias=0.3),
                     64, 33, ray_stride=16, chunk=2000)


if __name__ == "__main__":
    main()
