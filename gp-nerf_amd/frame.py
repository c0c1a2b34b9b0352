"""Per-frame constants and the Python face of the fused render entry point.

torch is used for device memory and streams only; all arithmetic of the path runs in
the HIP library (include/gpnerf_hip.h).  Nothing here falls back to PyTorch ops.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_WORKSPACES = {}


def _workspace(dev, nbytes):
    """The scratch a gpnerf_render_fused call borrows (tile queues, chained lists, the colour list: 0.5 GB for 512 x 512 x 64, 2 GB for
    1024 x 1024 x 64).  One tensor per (device, stream), kept and grown rather than allocated per call: the call owns it only until the
    stream reaches its end, and calls on one stream are ordered -- while a fresh torch.empty of that size per call can fall out of the
    caching allocator's pool and cost a device allocation (milliseconds of idle device) every frame."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(dev).cuda_stream))
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        _WORKSPACES.pop(key, None)
        ws = None               # (release the smaller one first)
        while len(_WORKSPACES) >= 4:        # (streams come and go: at most four are remembered)
            _WORKSPACES.pop(next(iter(_WORKSPACES)))
        ws = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
        _WORKSPACES[key] = ws
    return ws


def _require_gpu(t, what):
    if not t.is_cuda:
        raise L.GpnerfError(f"{what} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")


def fetch_host(*items):
    """Small per-frame constants (camera matrices, Rh, Th, bounds, out_sh) as float64 numpy arrays with ONE device-to-host copy
    for all the device tensors among them (each `.cpu()` of its own is a synchronisation: six of them cost ~0.4 ms per frame)."""
    dev = [i for i, t in enumerate(items) if isinstance(t, torch.Tensor) and t.is_cuda]
    out = [None] * len(items)
    if dev:
        # float32 tensors travel as they are (the widening to float64 is exact and happens on the host: a cast per item on the
        # device was six launches per frame); anything else (int64 out_sh, float64 cameras) is widened on the device first
        f32 = [i for i in dev if items[i].dtype == torch.float32]
        rest = [i for i in dev if items[i].dtype != torch.float32]
        for group, widen in ((f32, False), (rest, True)):
            if not group:
                continue
            parts = [items[i].detach().reshape(-1) for i in group]
            if widen:
                parts = [p.to(torch.float64) for p in parts]
            cat = parts[0] if len(parts) == 1 else torch.cat(parts)
            flat = cat.cpu().numpy().astype(np.float64)
            pos = 0
            for i in group:
                n = items[i].numel()
                out[i] = flat[pos:pos + n].reshape(tuple(items[i].shape))
                pos += n
    for i, t in enumerate(items):
        if out[i] is None:
            out[i] = (t.detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)).astype(np.float64)
    return out


def pack_head(state, device):
    """Pack the per-ray MLP parameters into the kernel's LDS image (gpnerf_pack_head).

    ``state`` maps reference parameter names (relative to ``nerfhead.``, e.g.
    ``rgbhead.base_fc.0.weight``) to tensors / arrays.  Returns a float32 device tensor.
    """
    lib = L.lib()
    params = L.GpnerfHeadParams()
    keep = []
    for short, name in L.HEAD_FIELDS:
        for suffix, field in (("weight", "_w"), ("bias", "_b")):
            v = state[f"{name}.{suffix}"]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            want = L.HEAD_SHAPES[short] if suffix == "weight" else (L.HEAD_SHAPES[short][0],)
            if tuple(a.shape) != want:
                raise L.GpnerfError(f"{name}.{suffix}: shape {a.shape}, expected {want}")
            keep.append(a)
            setattr(params, short + field, a.ctypes.data_as(L.FP))
    n = int(lib.gpnerf_head_blob_floats())
    blob = np.zeros(n, np.float32)
    L.check(lib.gpnerf_pack_head(C.byref(params), blob.ctypes.data_as(L.FP)), "gpnerf_pack_head")
    out = torch.from_numpy(blob).to(device)
    # the f16 hi/lo image for GPNERF_FLAG_SPLIT_F16 rides along as an attribute of the fp32 one
    ns = int(lib.gpnerf_head_blob_split_floats())
    sblob = np.zeros(ns, np.float32)
    L.check(lib.gpnerf_pack_head_split(C.byref(params), sblob.ctypes.data_as(L.FP)), "gpnerf_pack_head_split")
    out._gpnerf_split = torch.from_numpy(sblob).to(device)
    # ... and the reference-order image (GPNERF_FLAG_REF_ORDER; the form every render uses unless it asks for folded levels)
    rblob = np.zeros(n, np.float32)
    L.check(lib.gpnerf_pack_head_ref(C.byref(params), rblob.ctypes.data_as(L.FP)), "gpnerf_pack_head_ref")
    out._gpnerf_ref = torch.from_numpy(rblob).to(device)
    return out


class Frame:
    """Everything render_rays reads that does not depend on the ray (BaseRender.py:110-157),
    re-laid out channels-last on the device.  Built once per target view."""

    def __init__(self, src_imgs, featmaps, volumes, src_Ks, src_poses, Rh, Th, bounds_min, voxel_size, out_sh,
                 head_blob, consts=None, imgs4=None):
        """
        src_imgs   [V,3,H,W] in [-1,1] (batch['src_imgs'][0]); de-normalised here (BaseRender.py:231)
        featmaps   [V,32,h,w]           encoder output (BaseRender.py:222)
        volumes    4 x [1,32,D,H,W] or [32,D,H,W]   dense levels (SparseConvNet.py:111)
        src_Ks [V,3,3], src_poses [V,3,4], Rh [3,3], Th [1,3] or [3], bounds_min [3] (xyz), voxel_size [3], out_sh [3] (dhw)
        head_blob  device tensor from pack_head()
        consts     optional: what fetch_host(src_Ks, src_poses, Rh, Th, bounds_min, voxel_size, out_sh) returned earlier, so that
                   building the frame does not synchronise with the device (Renderer.render fetches them before it enqueues
                   the encoder, while the queue is still empty)
        imgs4      optional: relayout_images(src_imgs) done earlier (Renderer.render does it beside the encoder)
        """
        lib = L.lib()
        dev = src_imgs.device
        for t, w in ((src_imgs, "src_imgs"), (featmaps, "featmaps"), (head_blob, "head_blob")):
            _require_gpu(t, w)
        if src_imgs.shape[0] != L.VIEWS or featmaps.shape[0] != L.VIEWS or featmaps.shape[1] != L.CH:
            raise L.GpnerfError(f"the kernels are built for V={L.VIEWS} views x {L.CH} channels "
                                f"(rgb_fc's 96 inputs hard-wire it: trainhead.py:96,143); got {tuple(src_imgs.shape)}, {tuple(featmaps.shape)}")
        if len(volumes) != L.LEVELS:
            raise L.GpnerfError(f"expected {L.LEVELS} volume levels, got {len(volumes)}")
        st = _stream_ptr(dev)
        self.device = dev
        f = L.GpnerfFrame()
        V, _, H, W = src_imgs.shape
        src = src_imgs.contiguous().float()
        if imgs4 is not None and (tuple(imgs4.shape) != (V, H, W, 4) or imgs4.dtype != torch.float32 or imgs4.device != dev or not imgs4.is_contiguous()):
            raise L.GpnerfError(f"imgs4 must be relayout_images(src_imgs): float32 [{V},{H},{W},4] on {dev}")
        self.imgs = imgs4 if imgs4 is not None else relayout_images(src)
        fh, fw = featmaps.shape[-2:]
        if (featmaps.dtype == torch.float32 and not featmaps.is_contiguous()
                and featmaps.is_contiguous(memory_format=torch.channels_last)):
            fm = featmaps                       # physical [V,h,w,32] already (encoder.ResUNet on the GPU): no copy
            self.featmaps = featmaps.permute(0, 2, 3, 1)
        else:
            fm = featmaps.contiguous().float()
            self.featmaps = torch.empty((V, fh, fw, L.CH), device=dev, dtype=torch.float32)
            L.check(lib.gpnerf_relayout_featmaps(fm.data_ptr(), self.featmaps.data_ptr(), V, fh, fw, st), "gpnerf_relayout_featmaps")
        keep = [src, fm]
        self._set_volumes(f, volumes, keep)
        self._keep = keep  # sources stay alive until the re-layout kernels have run (stream order)
        f.featmaps, f.feat_h, f.feat_w = self.featmaps.data_ptr(), fh, fw
        f.imgs, f.img_h, f.img_w = self.imgs.data_ptr(), H, W
        # K4 @ P4 in fp32, as train_intrinsics.bmm(train_poses) does (BaseRender.py:233-247,314)
        Ks_h, poses_h, Rh_h, Th_h, bmin_h, vox_h, osh_h = consts if consts is not None else fetch_host(src_Ks, src_poses, Rh, Th, bounds_min,
                                                                                                          voxel_size, out_sh)
        KP = np.zeros((2, V, 4, 4), np.float32)
        KP[:, :, 3, 3] = 1.0
        KP[0, :, :3, :3] = Ks_h.astype(np.float32).reshape(V, 3, 3)
        KP[1, :, :3, :4] = poses_h.astype(np.float32).reshape(V, 3, 4)
        KPt = torch.from_numpy(KP)
        M = torch.bmm(KPt[0], KPt[1]).numpy()              # (torch's own CPU product: its multiply-add order is the reference's)
        for v in range(V):
            f.proj[v][:] = M[v].ravel()[:12].tolist()

        def flat(a, n):
            a = a.astype(np.float32).ravel()
            assert a.size == n, (a.shape, n)
            return a.tolist()

        f.Rh[:] = flat(Rh_h, 9)
        f.Th[:] = flat(Th_h, 3)
        f.bounds_min[:] = flat(bmin_h, 3)
        f.voxel[:] = flat(vox_h, 3)
        f.out_sh[:] = [int(v) for v in osh_h.ravel()[:3]]
        self.head_blob = head_blob
        f.head_blob = head_blob.data_ptr()
        self.head_blob_split = getattr(head_blob, "_gpnerf_split", None)
        f.head_blob_split = self.head_blob_split.data_ptr() if self.head_blob_split is not None else None
        self.head_blob_ref = getattr(head_blob, "_gpnerf_ref", None)
        f.head_blob_ref = self.head_blob_ref.data_ptr() if self.head_blob_ref is not None else None
        self.c = f

    def _set_volumes(self, f, volumes, keep):
        lib = L.lib()
        self.vols = []
        # whatever was derived from the previous levels goes with them: the folded coarse levels and the occupancy volume
        # would otherwise be read with the NEW levels' dimensions
        self._folded_valid, self.vols_folded, self.occ = False, None, None
        f.occ = None
        for l in range(L.LEVELS):
            f.vol_folded[l] = None
        for l, v in enumerate(volumes):
            _require_gpu(v, f"volumes[{l}]")
            if getattr(v, "_gpnerf_ndhwc", False):            # already channels-last (gpnerf_sparse_to_dense): no copy
                if v.shape[-1] != L.CH or v.dtype != torch.float32 or not v.is_contiguous():
                    raise L.GpnerfError(f"volume level {l}: expected contiguous fp32 [D,H,W,{L.CH}]")
                self.vols.append(v)
                f.vol[l] = v.data_ptr()
                f.vol_dhw[l][0], f.vol_dhw[l][1], f.vol_dhw[l][2] = v.shape[0], v.shape[1], v.shape[2]
                continue
            v = v.reshape(v.shape[-4:]).contiguous().float()
            if v.shape[0] != L.CH:
                raise L.GpnerfError(f"volume level {l}: {v.shape[0]} channels, expected {L.CH}")
            D, Hh, Ww = v.shape[1:]
            o = torch.empty((D, Hh, Ww, L.CH), device=v.device, dtype=torch.float32)
            L.check(lib.gpnerf_relayout_volume(v.data_ptr(), o.data_ptr(), D, Hh, Ww, _stream_ptr(v.device)),
                    "gpnerf_relayout_volume")
            self.vols.append(o)
            keep.append(v)
            f.vol[l] = o.data_ptr()
            f.vol_dhw[l][0], f.vol_dhw[l][1], f.vol_dhw[l][2] = D, Hh, Ww

    def build_occupancy(self):
        """SparseConvNet.encode's `masks3d` (SparseConvNet.py:135-139) from the 4 levels; enables occ_cull renders."""
        D, H, W = self.vols[0].shape[:3]
        self.occ = torch.empty((D, H, W), device=self.vols[0].device, dtype=torch.float32)
        L.check(L.lib().gpnerf_build_occupancy(C.byref(self.c), self.occ.data_ptr(), _stream_ptr(self.occ.device)),
                "gpnerf_build_occupancy")
        self.c.occ = self.occ.data_ptr()
        return self.occ

    def fold_volumes(self):
        """gpnerf_fold_volumes: the sigma feature layer applied to every voxel of the two coarse levels (64 values per voxel), so
        that the fp32 form interpolates their share of the layer's pre-activation instead of running it per sample.  Per-frame
        work (~10 us): the buffers are allocated once per Frame, every call recomputes them on the current stream."""
        if getattr(self, "vols_folded", None) is None:
            self.vols_folded = [torch.empty(tuple(v.shape[:3]) + (2 * L.CH,), device=v.device, dtype=torch.float32) if l >= L.FOLD_FIRST_LEVEL
                                else None for l, v in enumerate(self.vols)]
        ptrs = (C.c_void_p * L.LEVELS)(*[v.data_ptr() if v is not None else None for v in self.vols_folded])
        L.check(L.lib().gpnerf_fold_volumes(C.byref(self.c), ptrs, _stream_ptr(self.vols[0].device)), "gpnerf_fold_volumes")
        self._folded_valid = True
        return self.vols_folded

    @classmethod
    def for_volumes(cls, volumes, head_blob):
        """A frame that carries only the 4 dense levels (enough for gpnerf_sample_volume)."""
        if len(volumes) != L.LEVELS:
            raise L.GpnerfError(f"expected {L.LEVELS} volume levels, got {len(volumes)}")
        self = cls.__new__(cls)
        f = L.GpnerfFrame()
        self._keep = []
        self._set_volumes(f, volumes, self._keep)
        self.device = self.vols[0].device
        self.head_blob = head_blob
        f.head_blob = head_blob.data_ptr() if head_blob is not None else None
        self.c = f
        return self

    @staticmethod
    def consts_of_batch(batch, voxel_size):
        """The frame's small constants on the host, ONE device-to-host copy: Frame(..., consts=this)."""
        return fetch_host(batch["src_Ks"][0], batch["src_poses"][0], batch["Rh"][0], batch["Th"][0], batch["bounds"][0, 0], voxel_size,
                          batch["out_sh"][0])

    @classmethod
    def from_batch(cls, batch, featmaps, volumes, voxel_size, head_blob, consts=None, imgs4=None):
        """batch: the reference's batch dict (leading dim 1) with device tensors."""
        return cls(batch["src_imgs"][0], featmaps, volumes, batch["src_Ks"][0], batch["src_poses"][0], batch["Rh"][0],
                   batch["Th"][0], batch["bounds"][0, 0], voxel_size, batch["out_sh"][0], head_blob, consts=consts, imgs4=imgs4)


def relayout_images(src_imgs):
    """[V,3,H,W] in [-1,1] -> the frame's [V,H,W,4] image (de-normalised, channels-last, one padding channel: a pixel is one
    16-byte load), on the current stream (gpnerf_relayout_images)."""
    _require_gpu(src_imgs, "src_imgs")
    src = src_imgs.contiguous().float()
    V, _, H, W = src.shape
    out = torch.empty((V, H, W, 4), device=src.device, dtype=torch.float32)
    L.check(L.lib().gpnerf_relayout_images(src.data_ptr(), out.data_ptr(), V, H, W, _stream_ptr(src.device)), "gpnerf_relayout_images")
    return out


def patch_order(mask_at_box, H, W, patch_w=32, patch_h=8):
    """Permutation of the hit-ray list (raster order of `mask_at_box`) into image patches: each run of
    patch_w*patch_h consecutive slots covers one patch, one patch row (patch_w pixels) per wavefront.
    Returns an int32 numpy array for render_fused(ray_order=...)."""
    m = np.asarray(mask_at_box).reshape(H, W).astype(bool)
    idx = np.full((H, W), -1, np.int64)
    idx[m] = np.arange(int(m.sum()))
    Hp, Wp = -(-H // patch_h) * patch_h, -(-W // patch_w) * patch_w
    pad = np.full((Hp, Wp), -1, np.int64)
    pad[:H, :W] = idx
    t = pad.reshape(Hp // patch_h, patch_h, Wp // patch_w, patch_w).transpose(0, 2, 1, 3).reshape(-1)
    return t[t >= 0].astype(np.int32)


def patch_order_device(mask, H, W, patch_w=4, patch_h=8, n_kept=None):
    """patch_order() on the device for a bool mask [H*W]: int32 permutation that lays the kept pixels out patch by patch,
    so that a wavefront's 32 rays cover a compact patch_w x patch_h block instead of a 32-pixel row.  With sample culling a
    compact block is empty or full together far more often (measured: -10 % frame time at 10-30 % occupancy)."""
    m = mask.view(H, W)
    idx = (torch.cumsum(m.reshape(-1).to(torch.int32), 0, dtype=torch.int32) - 1).view(H, W)
    idx = torch.where(m, idx, torch.full_like(idx, -1))
    Hp, Wp = -(-H // patch_h) * patch_h, -(-W // patch_w) * patch_w
    if (Hp, Wp) != (H, W):
        idx = torch.nn.functional.pad(idx, (0, Wp - W, 0, Hp - H), value=-1)
    t = idx.view(Hp // patch_h, patch_h, Wp // patch_w, patch_w).permute(0, 2, 1, 3).reshape(-1)
    if n_kept is not None:
        # the caller knows how many pixels the mask keeps (it holds their rays): compaction with a static size, no host round trip
        # (boolean indexing synchronises to learn the count); a wrong n_kept shows up as -1 entries / a short list and is rejected
        pos = torch.nonzero_static(t >= 0, size=int(n_kept), fill_value=-1).squeeze(1)
        return t.index_select(0, pos.clamp_min(0)).masked_fill_(pos < 0, -1).contiguous()
    return t[t >= 0].contiguous()


def patch_order_rays(mask, H, W, n, patch_w=32, patch_h=8):
    """patch_order() for the n rays of a frame whose kept pixels are `mask` (bool / uint8 device tensor [H*W]): three launches, no
    host round trip (gpnerf_patch_order); the identity when the mask does not keep exactly n pixels."""
    _require_gpu(mask, "mask_at_box")
    m = mask.reshape(-1)
    m = (m if m.dtype in (torch.uint8, torch.bool) else (m != 0)).contiguous()
    if m.numel() != H * W:
        raise L.GpnerfError(f"mask has {m.numel()} entries for a {H}x{W} image")
    lib = L.lib()
    scratch = torch.empty((int(lib.gpnerf_patch_order_scratch_bytes(H, W, patch_w, patch_h)) // 4,), device=m.device, dtype=torch.int32)
    order = torch.empty((int(n),), device=m.device, dtype=torch.int32)
    L.check(lib.gpnerf_patch_order(m.data_ptr(), H, W, patch_w, patch_h, int(n), scratch.data_ptr(), order.data_ptr(), _stream_ptr(m.device)),
            "gpnerf_patch_order")
    return order


def render_fused(frame, rays, n_samples, neg_ray=False, early_term=False, term_eps=1e-5,
                 want=("weights", "z_vals", "rgb_in", "ray_mask"), ray_order=None, occ_cull=False, load_balance=True,
                 split_f16=False, flip=None, subset=False, guard=None, fold=None, reserve_cus=0, exits=True, workspace_cap=None, shared_device=False):
    """gpnerf_render_fused over rays [N,8] (device).  Returns a dict of device tensors [N,...].
    neg_ray: the Projector's front test (h_z < 0).  flip: raw2outputs(neg=True); defaults to neg_ray for the dense renderer
    (BaseRender.py:86-88) and to False with occ_cull, because the progressive renderer's integral never flips
    (demo_render.py:329-344).
    ray_order: optional int32 device tensor [N], a permutation that groups rays into cache-friendly tiles.
    load_balance: lend the kernel a workspace: large frames run persistent workgroups on a tile queue, small frames split a
    tile's samples over several wavefronts.
    split_f16: dense layers on f16 MFMA with fp32 operands split into hi + lo (GPNERF_FLAG_SPLIT_F16).
    guard: with split_f16, check every MFMA operand against the f16 range and render the tiles that reach it again in the fp32
    form (GPNERF_FLAG_SPLIT_GUARD; default on whenever the kernel has a workspace).  want=("guard_tiles",) returns how many
    32-ray tiles that were (int32 tensor [1]).
    subset: ray_order lists the rows of `rays` to render (any number of distinct rows); outputs keep rays' row count, rows
    that are not listed come back zero.
    fold: which fp32 form.  False / None (default): the REFERENCE-ORDER form -- every dense layer accumulates as the reference's
    sgemm does (k ascending from zero, bias last, unscaled), x / 3 and the trilinear taps round as the reference's do: on trained
    parameters it sits at the op-for-op CPU oracle's distance from the reference.  True: the round-4 fast
    form -- coarse levels folded into the sigma feature layer per frame (Frame.fold_volumes), log2(e)-scaled layers: ~8 % faster layer for layer (the same time once both defer the colour branch),
    the same 1e-5 at initialisation scale, 5-10 x further from the reference on trained-like parameters.  "keep": True without
    re-folding a Frame that is already folded.
    shared_device=True: other processes' kernels share the device (GPNERF_FLAG_SHARED_DEVICE): no launch waits for its own workgroups.
    exits=False: every layer evaluated for every sample (GPNERF_FLAG_NO_EXITS).  By default the fp32 forms leave out what cannot
    change an output, bit for bit: the sigma feature layer of levels whose features are zero in all 32 samples of a step, and the
    colour branch of samples whose weight alpha * T is zero (the rest are listed and evaluated 32 at a time -- by a second launch
    over the whole frame's list where the workspace has room for it, out of a per-wavefront queue otherwise: the same bits),
    and everything behind the sample at which all 32 rays of a tile have a transmittance of exactly 0;
    want=("step_stats",) returns the 8 counters of GpnerfOutputs.step_stats (steps, empty-space steps, steps minus colour
    evaluations, opaque-tail steps, volume levels left out, colour evaluations, 0, 0).  A launch that returns `raw` keeps the
    colour branch in the step.
    reserve_cus: plan the launch for that many fewer compute units (multiple of 8), leaving them to kernels of other streams
    (GPNERF_FLAG_RESERVE_CUS).  The maps are those of a chip with that many fewer CUs."""
    lib = L.lib()
    _require_gpu(rays, "rays")
    rays = rays.contiguous().float()
    N, S = rays.shape[0], int(n_samples)
    dev = rays.device
    if subset:
        if ray_order is None:
            raise L.GpnerfError("subset=True needs ray_order (the rows to render)")
        alloc = torch.zeros
    else:
        alloc = torch.empty
    res = {
        "rgb_map": alloc((N, 3), device=dev), "depth_map": alloc((N,), device=dev),
        "acc_map": alloc((N,), device=dev), "disp_map": alloc((N,), device=dev),
    }
    o = L.GpnerfOutputs()
    o.rgb, o.depth, o.acc, o.disp = (res[k].data_ptr() for k in ("rgb_map", "depth_map", "acc_map", "disp_map"))
    if "weights" in want:
        res["weights"] = torch.empty((N, S), device=dev)
        o.weights = res["weights"].data_ptr()
    if "z_vals" in want:
        res["z_vals"] = torch.empty((N, S), device=dev)
        o.z_vals = res["z_vals"].data_ptr()
    if "rgb_in" in want:
        res["rgb_in_map"] = torch.empty((N, 9), device=dev)
        o.rgb_in = res["rgb_in_map"].data_ptr()
    if "ray_mask" in want:
        res["ray_mask"] = torch.empty((N,), device=dev, dtype=torch.uint8)
        o.ray_mask = res["ray_mask"].data_ptr()
    if "raw" in want:
        res["raw"] = torch.empty((N, S, 4), device=dev)
        o.raw = res["raw"].data_ptr()
    if "step_stats" in want:                               # include/gpnerf_hip.h GpnerfOutputs.step_stats: 8 counters
        res["step_stats"] = torch.zeros((8,), device=dev, dtype=torch.int32)
        o.step_stats = res["step_stats"].data_ptr()
    if "samples_done" in want:
        res["samples_done"] = torch.empty((N,), device=dev, dtype=torch.int32)
        o.samples_done = res["samples_done"].data_ptr()
    if flip is None:
        flip = bool(neg_ray) and not occ_cull
    flags = (L.FLAG_NEG_RAY if neg_ray else 0) | (L.FLAG_FLIP_SAMPLES if flip else 0) | (L.FLAG_EARLY_TERM if early_term else 0)
    flags |= (int(reserve_cus) & 0xff) << 24
    if not exits:
        flags |= L.FLAG_NO_EXITS
    if shared_device:
        flags |= L.FLAG_SHARED_DEVICE
    if split_f16:
        if not frame.c.head_blob_split:
            raise L.GpnerfError("split_f16 needs the f16 hi/lo head image (build the frame from pack_head()'s tensor)")
        flags |= L.FLAG_SPLIT_F16
        if guard is None:
            guard = bool(load_balance)
        if guard:
            if not load_balance:
                raise L.GpnerfError("the split form's range guard keeps its flags in the workspace (load_balance=True)")
            flags |= L.FLAG_SPLIT_GUARD
    if occ_cull:
        if not frame.c.occ:
            frame.build_occupancy()
        flags |= L.FLAG_OCC_CULL
    if ray_order is not None:
        _require_gpu(ray_order, "ray_order")
        if ray_order.dtype != torch.int32 or not ray_order.is_contiguous() or (ray_order.numel() != N and not subset):
            raise L.GpnerfError("ray_order must be a contiguous int32 tensor with one entry per ray")
    n_launch = int(ray_order.numel()) if subset else N
    refold = fold is True
    fold = bool(fold)
    if fold and not split_f16:
        if refold or not getattr(frame, "_folded_valid", False):
            frame.fold_volumes()
        for l in range(L.FOLD_FIRST_LEVEL, L.LEVELS):
            frame.c.vol_folded[l] = frame.vols_folded[l].data_ptr()
        if L._DEBUG and os.environ.get("GPNERF_X_VIEWTAB") == "1":       # tools/probes/view_fold_proxy.sh: a [V][h][w][64] table for the diagnostic builds
            if getattr(frame, "_viewtab", None) is None:
                frame._viewtab = torch.randn((L.VIEWS, frame.c.feat_h, frame.c.feat_w, 64), device=rays.device)
            frame.c.vol_folded[0] = frame._viewtab.data_ptr()
    else:
        for l in range(L.LEVELS):
            frame.c.vol_folded[l] = None
    if subset and any(k in want for k in ("weights", "z_vals", "raw")):
        raise L.GpnerfError("subset launches return the per-ray maps only")
    ws_bytes = int(lib.gpnerf_render_workspace_bytes(n_launch, S)) if load_balance else 0
    if workspace_cap is not None:       # lend less than the launch could use (it then keeps to the forms that fit: include/gpnerf_hip.h `workspace`)
        ws_bytes = min(ws_bytes, int(workspace_cap))
    ws = _workspace(dev, ws_bytes) if ws_bytes else None
    L.check(lib.gpnerf_render_fused(C.byref(frame.c), rays.data_ptr(), n_launch, S, flags, float(term_eps),
                                    ray_order.data_ptr() if ray_order is not None else None, C.byref(o),
                                    ws.data_ptr() if ws is not None else None, ws_bytes, _stream_ptr(dev)), "gpnerf_render_fused")
    if "guard_tiles" in want:
        # the flag count is the first word of the guard block, which is the tail of the workspace (include/gpnerf_hip.h)
        if flags & L.FLAG_SPLIT_GUARD:
            off = ((ws_bytes - int(lib.gpnerf_render_guard_bytes(n_launch))) // 256) * 256
            res["guard_tiles"] = ws[off:off + 4].view(torch.int32).clone()
        else:
            res["guard_tiles"] = torch.zeros((1,), device=dev, dtype=torch.int32)
    return res


def _ref_image(head_blob):
    """the reference-order image riding on pack_head()'s tensor (what the stage entry points stage into LDS)"""
    ref = getattr(head_blob, "_gpnerf_ref", None)
    if ref is None:
        raise L.GpnerfError("head_blob must be pack_head()'s tensor (it carries the reference-order image)")
    return ref


def head_forward(head_blob, vol_feat, rgb_feat, mask):
    """gpnerf_head_forward: vol_feat [P,128], rgb_feat [P,V,35], mask [P,V] -> raw [P,4]."""
    lib = L.lib()
    for t, w in ((vol_feat, "vol_feat"), (rgb_feat, "rgb_feat"), (mask, "mask")):
        _require_gpu(t, w)
    vol_feat, rgb_feat, mask = vol_feat.contiguous().float(), rgb_feat.contiguous().float(), mask.contiguous().float()
    P = vol_feat.shape[0]
    raw = torch.empty((P, 4), device=vol_feat.device)
    L.check(lib.gpnerf_head_forward(_ref_image(head_blob).data_ptr(), vol_feat.data_ptr(), rgb_feat.data_ptr(), mask.data_ptr(), P,
                                    raw.data_ptr(), _stream_ptr(vol_feat.device)), "gpnerf_head_forward")
    return raw


def sigma_features(head_blob, vol_feat, rgb_feat):
    """gpnerf_sigma_features: vol_feat [P,128], rgb_feat [P,V,35] -> sigma_feat [P,64], globalfeat [P,134]
    (NeRFSigmaHead.test_forward after its volume sampling, trainhead.py:61-76)."""
    lib = L.lib()
    for t, w in ((vol_feat, "vol_feat"), (rgb_feat, "rgb_feat")):
        _require_gpu(t, w)
    vol_feat, rgb_feat = vol_feat.contiguous().float(), rgb_feat.contiguous().float()
    P = vol_feat.shape[0]
    sf = torch.empty((P, 64), device=vol_feat.device)
    gf = torch.empty((P, 134), device=vol_feat.device)
    L.check(lib.gpnerf_sigma_features(_ref_image(head_blob).data_ptr(), vol_feat.data_ptr(), rgb_feat.data_ptr(), P, sf.data_ptr(), gf.data_ptr(),
                                      _stream_ptr(vol_feat.device)), "gpnerf_sigma_features")
    return sf, gf


def rgb_head_forward(head_blob, sigma_feat, rgb_feat, mask):
    """gpnerf_rgb_head_forward: sigma_feat [P,64], rgb_feat [P,V,35], mask [P,V] -> raw [P,4] (NeRFRGBHead.forward, trainhead.py:118-145)."""
    lib = L.lib()
    for t, w in ((sigma_feat, "sigma_feat"), (rgb_feat, "rgb_feat"), (mask, "mask")):
        _require_gpu(t, w)
    sigma_feat, rgb_feat, mask = sigma_feat.contiguous().float(), rgb_feat.contiguous().float(), mask.contiguous().float()
    P = sigma_feat.shape[0]
    raw = torch.empty((P, 4), device=sigma_feat.device)
    L.check(lib.gpnerf_rgb_head_forward(_ref_image(head_blob).data_ptr(), sigma_feat.data_ptr(), rgb_feat.data_ptr(), mask.data_ptr(), P,
                                        raw.data_ptr(), _stream_ptr(sigma_feat.device)), "gpnerf_rgb_head_forward")
    return raw


def composite(raw, z_vals, nvalid=None, neg=False):
    """gpnerf_composite: Renderer.raw2outputs (BaseRender.py:75-107)."""
    lib = L.lib()
    _require_gpu(raw, "raw")
    raw, z_vals = raw.contiguous().float(), z_vals.contiguous().float()
    N, S = z_vals.shape
    dev = raw.device
    res = {"rgb_map": torch.empty((N, 3), device=dev), "depth_map": torch.empty((N,), device=dev),
           "acc_map": torch.empty((N,), device=dev), "disp_map": torch.empty((N,), device=dev),
           "weights": torch.empty((N, S), device=dev), "ray_mask": torch.zeros((N,), device=dev, dtype=torch.uint8)}
    o = L.GpnerfOutputs()
    o.rgb, o.depth, o.acc, o.disp = (res[k].data_ptr() for k in ("rgb_map", "depth_map", "acc_map", "disp_map"))
    o.weights, o.ray_mask = res["weights"].data_ptr(), res["ray_mask"].data_ptr()
    nv = nvalid.contiguous().float() if nvalid is not None else None
    L.check(lib.gpnerf_composite(raw.data_ptr(), z_vals.data_ptr(), nv.data_ptr() if nv is not None else None, N, S,
                                 int(bool(neg)), C.byref(o), _stream_ptr(dev)), "gpnerf_composite")
    return res


def make_rays(H, W, K, R, T, bounds, device):
    """gpnerf_make_rays: get_rays + get_near_far (data_utils.py:47-63,96-130) on the device, bit-exact against numpy's run.
    K, R, T are used in the dtype they come in (the dataset reads them as float64, ZjumocapDataset.py:360-380): the
    inverses are np.linalg.inv in that dtype, as get_rays takes them (:49-51,57).

    Returns (rays [n,8] packed in raster order of the hit pixels, mask_at_box [H*W] bool)."""
    lib = L.lib()
    K, R, T = np.asarray(K), np.asarray(R), np.asarray(T).reshape(3, 1)
    R_inv = np.linalg.inv(R)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    Kinv, Rinv, o = f64(np.linalg.inv(K)), f64(R_inv), f64((-R_inv @ T).ravel())
    b = np.ascontiguousarray(np.asarray(bounds, np.float32))
    rays = torch.empty((H * W, 8), device=device)
    hit = torch.empty((H * W,), device=device, dtype=torch.uint8)
    L.check(lib.gpnerf_make_rays(H, W, Kinv.ctypes.data_as(L.DP), Rinv.ctypes.data_as(L.DP), o.ctypes.data_as(L.DP),
                                 b.ctypes.data_as(L.FP), rays.data_ptr(), hit.data_ptr(), _stream_ptr(rays.device)),
            "gpnerf_make_rays")
    mask = hit.bool()
    return rays[mask], mask


def sample_points(frame, rays, n_samples):
    """gpnerf_sample_points: (pts [N,S,3], z_vals [N,S], grid_coords [N,S,3]) as BaseRender.py:35-73 computes them."""
    lib = L.lib()
    _require_gpu(rays, "rays")
    rays = rays.contiguous().float()
    N, S, dev = rays.shape[0], int(n_samples), rays.device
    pts, z, grid = torch.empty((N, S, 3), device=dev), torch.empty((N, S), device=dev), torch.empty((N, S, 3), device=dev)
    L.check(lib.gpnerf_sample_points(C.byref(frame.c), rays.data_ptr(), N, S, pts.data_ptr(), z.data_ptr(), grid.data_ptr(),
                                     _stream_ptr(dev)), "gpnerf_sample_points")
    return pts, z, grid


def sample_volume(frame, grid):
    """gpnerf_sample_volume: grid [P,3] (normalised xyz) -> [P,128] features of the 4 dense levels."""
    lib = L.lib()
    _require_gpu(grid, "grid")
    grid = grid.reshape(-1, 3).contiguous().float()
    P = grid.shape[0]
    out = torch.empty((P, 128), device=grid.device)
    L.check(lib.gpnerf_sample_volume(C.byref(frame.c), grid.data_ptr(), P, out.data_ptr(), _stream_ptr(grid.device)),
            "gpnerf_sample_volume")
    return out


def project_gather(frame, pts, neg_ray=False):
    """gpnerf_project_gather: pts [P,3] -> (rgb_feat [P,V,35], mask [P,V]) as Projector.compute does (BaseRender.py:326-363)."""
    lib = L.lib()
    _require_gpu(pts, "pts")
    pts = pts.reshape(-1, 3).contiguous().float()
    P = pts.shape[0]
    feat = torch.empty((P, L.VIEWS, 35), device=pts.device)
    mask = torch.empty((P, L.VIEWS), device=pts.device)
    L.check(lib.gpnerf_project_gather(C.byref(frame.c), pts.data_ptr(), P, int(bool(neg_ray)), feat.data_ptr(), mask.data_ptr(),
                                      _stream_ptr(pts.device)), "gpnerf_project_gather")
    return feat, mask


def select_rays(frame, target_K, target_pose, H, W, voxel_size, bounds_min, Rh, Th, neg_ray=False, threshold=0.1,
                target_K_inv=None, compact=True, host=None):
    """Progressive ray selection of the inference renderer (demo_render.py:166-247) on the device:
    occupied voxels -> marked pixels (gpnerf_select_pixels) -> rays with near/far (gpnerf_make_rays_demo).
    Returns (rays [n,8] in raster order of the kept pixels, mask_at_box [H*W] bool); with compact=False the rays of ALL H*W
    pixels (rows of pixels that are not kept are unspecified) and the mask, without any host synchronisation when `host` --
    fetch_host(target_K, target_pose, voxel_size, bounds_min, Rh, Th[, target_K_inv]) done earlier -- is handed in."""
    lib = L.lib()
    if not frame.c.occ:
        frame.build_occupancy()
    occ = frame.occ
    dev = occ.device
    items = [target_K, target_pose, voxel_size, bounds_min, Rh, Th] + ([target_K_inv] if target_K_inv is not None else [])
    host = [np.ascontiguousarray(a.astype(np.float32).ravel()) for a in (host if host is not None else fetch_host(*items))]
    f32 = lambda a, n: a[:n]
    K, pose = f32(host[0], 9), f32(host[1], 12)
    vox, bmin, rh, th = f32(host[2], 3), f32(host[3], 3), f32(host[4], 9), f32(host[5], 3)
    sel = torch.empty((H * W,), device=dev, dtype=torch.uint8)
    mm = torch.empty((6,), device=dev, dtype=torch.int32)
    D1, H1, W1 = occ.shape
    st = _stream_ptr(dev)
    p = lambda a: a.ctypes.data_as(L.FP)
    L.check(lib.gpnerf_select_pixels(occ.data_ptr(), D1, H1, W1, float(threshold), p(vox), p(bmin), p(rh), p(th), p(pose), p(K),
                                     H, W, sel.data_ptr(), mm.data_ptr(), st), "gpnerf_select_pixels")
    # batch["target_K_inv"] is what the reference multiplies by (demo_render.py:204; the dataset makes it with
    # np.linalg.inv on the float32 K, ZjumocapDataset.py:480); without it, do the same here
    Kinv = f32(host[6], 9) if target_K_inv is not None else np.ascontiguousarray(np.linalg.inv(K.reshape(3, 3)).astype(np.float32).ravel())
    rays = torch.empty((H * W, 8), device=dev)
    hit = torch.empty((H * W,), device=dev, dtype=torch.uint8)
    # the box of the occupied voxels stays on the device (mm): no host round trip between the two launches
    L.check(lib.gpnerf_make_rays_demo(H, W, p(Kinv), p(pose), None, mm.data_ptr(), int(bool(neg_ray)),
                                      sel.data_ptr(), rays.data_ptr(), hit.data_ptr(), st), "gpnerf_make_rays_demo")
    mask = hit.bool()
    return (rays[mask], mask) if compact else (rays, mask)


def patch_order_of(idx, W, patch_w=4, patch_h=8):
    """The kept pixels idx (int64, raster order) re-ordered patch by patch (patch_w x patch_h pixel blocks, row-major inside a
    block): one key computation and one device sort.  Returns int32 row indices for render_fused(ray_order=..., subset=True)."""
    y, x = idx // W, idx % W
    key = ((y // patch_h) * ((W + patch_w - 1) // patch_w) + x // patch_w) * (patch_w * patch_h) + (y % patch_h) * patch_w + x % patch_w
    return idx[torch.argsort(key)].to(torch.int32)
