"""`build_render(cfg)` / `Renderer` with the reference's interface (libs/renders/BaseRender.py:11-403),
rendering through the fused HIP kernel.

Drop-in contract:
  * `build_render(cfg) -> nn.Module` whose `.render(batch)` returns the reference's dict
    (`rgb_map, disp_map, acc_map, depth_map, alpha, z_vals, rgb_in_map`, leading dims [1,N,.];
    BaseRender.py:148-156,254) plus `rtime` / `etime`, which tools/inference.py's caller needs
    (libs/trainers/BaseTrainer.py:276) and only the reference's demo renderer returned.
  * sub-modules are named `encoder` and `nerfhead` and own the reference's parameters, so
    `load_state_dict(ckpt['state_dict'], strict=True)` works (tools/inference.py:67-74).
  * sampling is deterministic (is_train=False, as libs/renders/demo_render.py:661 forces for inference;
    SURVEY.md §0-6 explains why the dense renderer's `is_train` latch is not reproduced).

The per-ray work never touches torch ops; the per-frame producers (`encoder.py`: the image encoder, `volume.py`: vertex
attention + sparse volume builder) are nn.Modules under the reference's parameter names whose forward passes are hand-written
HIP kernels as well (csrc/gpnerf_conv.hip, gpnerf_volume.hip; no torch operator computes anything), and can be bypassed by
putting their products into the batch (`batch['featmaps']`, `batch['volumes']`).
"""
import os
import time
from importlib import import_module as impm

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import encoder as E_
from . import frame as F_
from . import parallel as P_


class Renderer(nn.Module):
    def __init__(self, encoder, nerfhead, is_train=False, neg_ray_train=False, neg_ray_val=False, n_rays=1024,
                 n_samples=64, voxel_size=(0.005, 0.005, 0.005), chunk=64, mesh_th=-1, early_term=None, term_eps=1e-5,
                 progressive=False, split_f16=None, sharded_outputs="all", shard_group=None, encoder_graph=None, fold_levels=None, reserve_cus=None):
        super().__init__()
        self.encoder = encoder
        self.nerfhead = nerfhead
        if is_train:
            raise L.GpnerfError("the HIP render path is inference-only: stratified jitter / autograd (is_train=True) "
                                "stay on the reference's PyTorch renderer")
        self.is_train = False
        self.neg_ray_train, self.neg_ray_val = neg_ray_train, neg_ray_val
        self.n_rays, self.n_samples = n_rays, n_samples
        self.voxel_size = np.array(voxel_size)
        self.chunk = chunk          # kept for interface parity; the fused kernel tiles rays itself
        self.mesh_th = mesh_th
        # early_term (not in the reference): rays stop once their transmittance is below term_eps (rgb / acc change by at most
        # term_eps, depth by term_eps * far).  Off by default; GPNERF_EARLY_TERM=1 (and GPNERF_TERM_EPS) turn it on from outside,
        # e.g. for the reference's tools/inference.py run, which has no config key for it.
        if early_term is None:
            early_term = os.environ.get("GPNERF_EARLY_TERM", "0") == "1"
            term_eps = float(os.environ.get("GPNERF_TERM_EPS", term_eps))
        self.early_term, self.term_eps = bool(early_term), float(term_eps)
        # progressive=True: the inference renderer's path (libs/renders/demo_render.py): rays are selected from the
        # occupied voxels of the frame's volume, samples are occupancy-culled, and the result is returned as `pred_img`
        self.progressive = progressive
        # shard_group (not in the reference): OPT-IN strong scaling of one frame over the ranks of a torch.distributed process
        # group -- a ProcessGroup, or "world" for the default one (GPNERF_SHARD=1 in the environment means "world", for runs
        # through the reference's tools/inference.py, which has no config key for it).  Every rank of the group must call
        # render() with the SAME batch; each renders a share of the rays and encodes a share of the source views, and all get
        # the whole frame.  None (default): no collective is ever issued, whatever process groups exist -- under the
        # reference's DistributedSampler the ranks hold different frames (tools/train.py), and sharding would mix them.
        if shard_group is None and os.environ.get("GPNERF_SHARD", "0") == "1":
            shard_group = "world"
        self.shard_group = shard_group
        # with a shard_group: "all" = the reference's full dict on every rank (one packed all-gather),
        # "pixels" = rgb_map + depth_map only (16 B/ray over xGMI)
        self.sharded_outputs = sharded_outputs
        # encoder_graph: replay the image encoder's launches as one HIP graph per (image shape, parameter versions) when the encoder
        # offers it (hip_encoder does): same bits, ~1.3 ms less host time per frame (the host then prepares the frame while the
        # device encodes).  GPNERF_ENCODER_GRAPH=0 switches it off.
        self.encoder_graph = (os.environ.get("GPNERF_ENCODER_GRAPH", "1") != "0") if encoder_graph is None else bool(encoder_graph)
        # split_f16: dense layers on f16 MFMA with hi/lo operand pairs (GPNERF_FLAG_SPLIT_F16); same 1e-4 parity bound,
        # ~1.8x faster.  Default: exact fp32 MFMA, unless GPNERF_SPLIT_F16=1 is set in the environment.
        self.split_f16 = (os.environ.get("GPNERF_SPLIT_F16", "0") == "1") if split_f16 is None else bool(split_f16)
        # fold_levels: the fp32 form with the two coarse volume levels folded into the sigma feature layer once per frame and
        # log2(e)-scaled layers (round 4's default; `render.file hip_render_fold`): ~8 % faster, 1e-5 from the reference at
        # initialisation scale like the default, but 5-10 x further on trained-like parameters.  Default: the reference-order
        # form (frame.render_fused).  GPNERF_FOLD=1 switches it on from outside.
        self.fold_levels = (os.environ.get("GPNERF_FOLD", "0") == "1") if fold_levels is None else bool(fold_levels)
        # reserve_cus (GPNERF_RESERVE_CUS): in a PIPELINED loop (render(..., next_batch=...)) plan the per-ray launch for this many
        # fewer compute units, so that the next frame's encoder / builder run BESIDE it on the CUs it leaves.  Pays on frames of
        # several rounds of wavefronts (512x512 full frame: -3 % kernel for -1.4 ms of producers); on a ZJU-sized frame of ~one
        # round the smaller chip quantises badly and it loses (profiles/r05/d_pipeline.txt).  Default 0.
        self.reserve_cus = int(os.environ.get("GPNERF_RESERVE_CUS", "0")) if reserve_cus is None else int(reserve_cus)

    # ---- helpers the reference exposes as methods (stage entry points) ----------------------------
    def _neg_ray(self, batch):
        # BaseRender.py:165-168; the progressive renderer guards the key lookup (demo_render.py:380-384)
        if "body_msk" in batch and batch["body_msk"].shape[-1] > self.n_rays:
            return self.neg_ray_val
        return self.neg_ray_train

    def encode(self, batch, defer_range_check=False):
        """`featmaps = self.encoder(src_imgs.squeeze(0))` (BaseRender.py:222, demo_render.py:441): the part of a frame the
        reference's demo renderer reports as `etime`.  defer_range_check: see encoder.forward_graphed -- render() looks at the
        encoder's range flag at the end of the call, where it synchronises anyway."""
        src_imgs = batch["src_imgs"]
        if src_imgs.shape[0] != 1:
            raise L.GpnerfError("only batch_size=1 is supported (as BaseRender.py:336 asserts)")
        graphed = self.encoder_graph and src_imgs.is_cuda and isinstance(self.encoder, E_.ResUNet)
        if "featmaps" in batch:
            featmaps = batch["featmaps"]
        elif graphed and P_.resolve_group(self.shard_group) is None:
            featmaps = E_.forward_graphed(self.encoder, src_imgs.squeeze(0), defer_range_check=defer_range_check)
        else:
            # with a shard_group: the source views dealt out over the ranks + a broadcast of each feature map (parallel.py); a
            # rank replays its own views as ONE graph too (per view count: the graph's key holds the input shape) -- enqueued
            # launch by launch through Python the owner of a view was host-bound (~1.4 ms for ~0.4 ms of device time)
            # (the range check is deferred here too: render() agrees on ONE verdict across the ranks at the end of the call)
            fn = (lambda t: E_.forward_graphed(self.encoder, t, defer_range_check=defer_range_check)) if graphed else None
            featmaps = P_.encode_views_sharded(self.encoder, src_imgs.squeeze(0), group=self.shard_group, encode_fn=fn)
        return featmaps[0] if featmaps.dim() == 5 else featmaps

    def prepare_sp_input(self, batch, out_sh=None):
        """BaseRender.py:187-209 (the fields the volume builder reads).  out_sh: the host copy of max(batch['out_sh'], 0) when the
        caller already has it (the `.tolist()` below is a device synchronisation)."""
        sh = batch["coord"].shape
        idx = torch.arange(sh[0], device=batch["coord"].device, dtype=batch["coord"].dtype).repeat_interleave(sh[1])
        coord = torch.cat([idx[:, None], batch["coord"].view(-1, sh[-1])], dim=1)
        if out_sh is None:
            out_sh = torch.max(batch["out_sh"], dim=0)[0].tolist()
        sp = {"feature": batch["feature"].view(-1, batch["feature"].shape[-1]), "coord": coord,
              "out_sh": [int(v) for v in out_sh], "batch_size": sh[0], "Rh": batch["Rh"], "R": batch.get("R", batch["Rh"]),
              "src_imgs": batch["src_imgs"]}
        if "volumes" in batch:
            sp["volumes"] = batch["volumes"]
        return sp

    def prepare_builder_inputs(self, batch, consts=None):
        """What the volume builder needs that does not depend on the encoder: the sparse-conv input dict and the SMPL vertices in
        world space (BaseRender.py:128-131).  Renderer.render enqueues this BEFORE the encoder, so that the host-bound little
        launches run while the queue is empty and the encoder's ~110 launches are followed directly by the builder's."""
        if "volumes" in batch:
            return None
        out_sh = consts[6].ravel() if (consts is not None and batch["out_sh"].shape[0] == 1) else None
        xyz = batch["feature"][..., :3].float()
        smpl_xyz = torch.bmm(xyz, batch["Rh"].float().transpose(1, 2)) + batch["Th"].float()
        sp = self.prepare_sp_input(batch, out_sh)
        net = getattr(getattr(self.nerfhead, "sigmahead", None), "xyzc_net", None)
        if hasattr(net, "plan_levels") and not net.training and sp["coord"].is_cuda:
            # the pyramid's structure (index grids, coarse site lists, zeroed volumes) needs the voxel coordinates only: laid out
            # here, i.e. on Renderer.render's side stream while the encoder runs, instead of between the encoder and the per-ray kernel
            sp["plan"] = net.plan_levels(sp["coord"], sp["out_sh"])
        return sp, smpl_xyz

    def _placeholder_volumes(self, dev):
        """Four one-voxel channels-last levels: what a Frame carries until the builder's volumes are attached (cached per device)."""
        cache = self.__dict__.setdefault("_placeholders", {})
        if str(dev) not in cache:
            vols = [torch.zeros((1, 1, 1, L.CH), device=dev) for _ in range(L.LEVELS)]
            for v in vols:
                v._gpnerf_ndhwc = True
            cache[str(dev)] = vols
        return cache[str(dev)]

    def build_frame(self, batch, featmaps=None, consts=None, prepared=None, imgs4=None):
        """Per-frame work after the encoder: volume pyramid, channels-last re-layout, weight image.  consts: Frame.consts_of_batch()
        fetched earlier; with it nothing below synchronises with the device (a batch of one frame: out_sh[0] is the maximum).
        prepared: prepare_builder_inputs() done earlier."""
        dev = batch["src_imgs"].device
        if featmaps is None:
            featmaps = self.encode(batch)
        blob = self.nerfhead.head_blob(dev)
        if "volumes" in batch:
            return F_.Frame.from_batch(batch, featmaps, batch["volumes"], self.voxel_size, blob, consts=consts, imgs4=imgs4)
        # No pre-built pyramid: gather the SMPL vertices' per-view features with the image half of the frame
        # (BaseRender.py:128-131,344-347), run the per-frame volume builder (trainhead.py:48-56), then attach it.
        sp_input, smpl_xyz = prepared if prepared is not None else self.prepare_builder_inputs(batch, consts)
        frame = F_.Frame.from_batch(batch, featmaps, self._placeholder_volumes(dev), self.voxel_size, blob, consts=consts, imgs4=imgs4)
        feat, _ = F_.project_gather(frame, smpl_xyz[0], neg_ray=False)
        smpl_feat = feat[:, :, 3:].unsqueeze(0)                     # [1,6890,V,32]
        volumes = self.nerfhead.sigmahead.build_volumes(sp_input, smpl_feat)
        frame._set_volumes(frame.c, volumes, frame._keep)
        return frame

    def get_sampling_points(self, ray_o, ray_d, near, far, frame):
        rays = torch.cat([ray_o, ray_d, near[..., None], far[..., None]], -1)[0]
        pts, z, _ = F_.sample_points(frame, rays, self.n_samples)
        return pts[None], z[None]

    @staticmethod
    def raw2outputs(raw, z_vals, mask, neg):
        """BaseRender.py:75-107 on gpnerf_composite; mask [R,S] = pixel_mask (>1 valid view)."""
        nvalid = mask.float() * 2.0   # the kernel tests nvalid > 1
        o = F_.composite(raw, z_vals, nvalid, neg=bool(neg))
        alpha = 1.0 - torch.exp(-(torch.flip(raw[..., 3], [1]) if neg else raw[..., 3]))
        return o["rgb_map"], o["disp_map"], o["acc_map"], o["weights"], o["depth_map"], o["ray_mask"].bool(), alpha

    def _host_buffers(self, H, W, dev):
        """Pinned staging for the progressive renderer's numpy outputs (pred_img float64 [H,W,3], mask_at_box bool [H*W])."""
        key = (H, W, str(dev))
        if self.__dict__.get("_host_key") != key:
            self.__dict__["_host"] = {"img": torch.empty((H, W, 3), dtype=torch.float64, pin_memory=True),
                                      "mask": torch.empty((H * W,), dtype=torch.bool, pin_memory=True)}
            self.__dict__["_host_key"] = key
        return self.__dict__["_host"]

    def render_progressive(self, batch):
        """libs/renders/demo_render.py:429-498 + :96-376: returns `pred_img` [H,W,3] (float64 numpy, background 0),
        `mask_at_box`, `rgb_map`, `time_slots`, `rtime`, `etime` (libs/evaluators/if_nerf.py:50-56 reads pred_img[mask])."""
        dev = batch["src_imgs"].device
        H, W = batch["src_imgs"].shape[-2:]
        torch.cuda.synchronize(dev)
        te = time.time()
        # as in render(): the small constants come over in ONE copy while the queue is empty, what does not depend on the encoder
        # is enqueued before it, and the phases are timed with stream events -- the host waits for the device once, for the
        # number of selected pixels (the reference synchronises around every phase, demo_render.py:97-357)
        sel_items = [batch["target_K"][0], batch["target_pose"][0], self.voxel_size, batch["bounds"][0, 0], batch["Rh"][0], batch["Th"][0]]
        if "target_K_inv" in batch:
            sel_items.append(batch["target_K_inv"][0])
        fetched = F_.fetch_host(batch["src_Ks"][0], batch["src_poses"][0], batch["Rh"][0], batch["Th"][0], batch["bounds"][0, 0],
                                self.voxel_size, batch["out_sh"][0], *sel_items)
        consts, sel_host = fetched[:7], fetched[7:]
        prepared = self.prepare_builder_inputs(batch, consts)
        self.nerfhead.head_blob(dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        featmaps = self.encode(batch)
        ev[1].record()                                       # the reference restarts its clock here (demo_render.py:443-446)
        frame = self.build_frame(batch, featmaps, consts, prepared)
        frame.build_occupancy()
        ev[2].record()
        neg = self._neg_ray(batch)
        # every pixel's ray + the mask of the kept ones, no compaction and no host round trip (the box stays on the device)
        rays, mask = F_.select_rays(frame, batch["target_K"][0], batch["target_pose"][0], H, W, self.voxel_size,
                                    batch["bounds"][0, 0], batch["Rh"][0], batch["Th"][0], neg_ray=neg,
                                    target_K_inv=batch["target_K_inv"][0] if "target_K_inv" in batch else None, compact=False,
                                    host=sel_host)
        idx = torch.nonzero(mask).squeeze(1)                 # kept pixels in raster order (the frame's one synchronisation)
        ev[3].record()
        torch.cuda.synchronize(dev)
        t2 = time.time()
        etime = ev[0].elapsed_time(ev[1]) * 1e-3
        t_frame, t_select = ev[1].elapsed_time(ev[2]) * 1e-3, ev[2].elapsed_time(ev[3]) * 1e-3
        if idx.numel():
            # the kernel renders the listed rows of the H*W ray array, 4x8-pixel patches per wavefront (a compact tile is empty
            # or full together far more often: whole tile-steps are culled), and writes each pixel's colour at its own row:
            # the zero-initialised output IS the image
            o = F_.render_fused(frame, rays, self.n_samples, neg_ray=neg, early_term=self.early_term, term_eps=self.term_eps,
                                occ_cull=True, want=(), split_f16=self.split_f16, ray_order=F_.patch_order_of(idx, W), subset=True)
            img = o["rgb_map"]
        else:
            img = torch.zeros((H * W, 3), device=dev)
        rgb = img.index_select(0, idx)
        torch.cuda.synchronize(dev)
        t3 = time.time()
        host = self._host_buffers(H, W, dev)
        host["img"].copy_(img.view(H, W, 3), non_blocking=True)          # float32 -> the reference's float64 image, in flight
        host["mask"].copy_(mask, non_blocking=True)
        rgb_np = rgb.cpu().numpy()                                        # synchronises: the two copies above are done
        pred_img = host["img"].numpy().copy()
        mask_np = host["mask"].numpy().copy()
        t4 = time.time()
        return {"rgb_map": rgb_np, "pred_img": pred_img, "mask_at_box": mask_np.reshape(-1),
                # this path's own phases, plus the reference's ten keys (demo_render.py:97-357) so that consumers indexing them
                # keep working: its per-frame phases collapse into `sp_encode`, its sigma / rgb passes into `sigma_f`
                "time_slots": {"frame": t_frame, "ray_select": t_select, "render": t3 - t2, "bc_render": t4 - t3,
                               "bc_time": t_select, "sigma_c": 0.0, "bc_attn": 0.0, "sigma_attn": 0.0, "sp_encode": t_frame,
                               "bf_sigma": 0.0, "sigma_f": t3 - t2, "bf_rgb": 0.0, "rgb_f": 0.0},
                # etime = the encoder alone (its device time), rtime = everything else of the call, as demo_render.py:441-446,494-497
                # keeps its two clocks (libs/trainers/BaseTrainer.py:276 sums rtime into the reported render time)
                "etime": etime, "rtime": max(0.0, (t4 - te) - etime)}

    # ---- the hot path ---------------------------------------------------------------------------------
    def _produce(self, batch):
        """Everything of a frame that comes BEFORE the per-ray kernel, enqueued on the current stream (+ a side stream): image
        encoder, the frame's constants, volume builder, channels-last re-layouts, ray list and patch order.  Returns the
        `Prefetched` record the per-ray launch consumes.  No device-wide synchronisation."""
        dev = batch["ray_o"].device
        # The encoder goes FIRST: it needs nothing but the source images, and its ~1 ms on the device cover everything the host
        # has to wait for -- the frame's small constants (one device-to-host copy) and the patch order -- which happen on a SIDE
        # stream meanwhile (the copy's synchronisation then waits for that stream's few microseconds, not for the encoder).  From
        # there to the end of the per-ray kernel the host only enqueues and runs ahead of the device, so a frame's ~130 launches
        # go back to back.  (Round 2 fetched the constants before the encoder's first launch: the device idled ~0.3 ms per call.)
        # the encoder's time comes from two events on the stream instead of two host synchronisations around it
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        own_encoder = "featmaps" not in batch
        if own_encoder:
            self._settle_encoder_verdicts()        # (split precision only) every earlier pass's range verdict is read before the graph runs again
        ev0.record()                               # (also what the side stream waits for: the batch's tensors, NOT the encoder)
        featmaps = self.encode(batch, defer_range_check=True)
        ev1.record()
        # the split-precision encoder's range flag of THIS pass (None in the default fp32 form, which has no range): kept with the
        # frame's record, so that two frames in flight never read each other's verdict
        enc_run = self.encoder.__dict__.pop("_gpnerf_pending_run", None) if isinstance(self.encoder, E_.ResUNet) else None
        self.nerfhead.head_blob(dev)                                   # (cached; packs on a parameter change)
        main = torch.cuda.current_stream(dev)
        sides = self.__dict__.setdefault("_side_streams", {})
        # (a batch that brings its feature maps has no encoder for the side stream's work to hide behind: everything then goes on
        # `main` in order -- every stream switch is ~10-20 us of dependency latency on an otherwise idle device, and the
        # record_stream bookkeeping below is host time: 0.71 -> 0.5 ms of glue on the bench frame, profiles/r06/g_render_glue.txt)
        side = sides.get(main.cuda_stream) if own_encoder else main    # one side stream per producing stream (render's own, prefetch's)
        if side is None or side.device != dev:
            if len(sides) > 8:
                sides.clear()
            side = sides[main.cuda_stream] = torch.cuda.Stream(device=dev)
        two_streams = side is not main
        group = P_.resolve_group(self.shard_group)
        sharded = group is not None
        # The host enqueues in the order the DEVICE needs things: first what the builder needs (it runs right behind the encoder),
        # then the builder and the frame, and only then the ray list and its patch order, which nobody reads before the per-ray
        # kernel -- with the encoder at 1.0 ms the host's ~1.4 ms of enqueueing had caught up with the device, and those 0.2 ms of
        # small tensor operations sat in front of the builder's launches.
        # INVARIANT for everything allocated under `side` and consumed on `main` (consts' device copies, the pyramid plan's buffers,
        # imgs4, rays, order): the caching allocator knows them as `side`'s blocks, and `main.wait_stream(side)` orders the USE, not
        # the FREE -- a block dropped early could be handed to side's next allocation while main's kernels still read it.  They are
        # therefore recorded on `main` below (record_stream), on top of every one of them staying referenced until the
        # synchronisation at the end of render().
        # the side stream starts from where `main` stood BEFORE the encoder was enqueued (the batch's tensors are complete there:
        # render() synchronised, prefetch() made its stream wait for the caller's).  Waiting for `main` as it stands now would put
        # the constants' device-to-host copy -- which the host blocks on -- behind the encoder, and in a pipelined loop behind the
        # previous frame's per-ray kernel that the encoder itself is queued behind (measured: prefetch() then took a whole frame).
        if two_streams:
            side.wait_event(ev0)
        with torch.cuda.stream(side):
            # (a pipelined loop fetches the constants when it moves the batch to the device -- host_consts() -- because HERE the
            # device-to-host copy would wait for a CU of the previous frame's persistent per-ray kernel: measured, the host then sat
            # in prefetch() for that kernel's whole 4.6 ms)
            consts = batch.get("_gpnerf_consts")
            if consts is None:
                consts = F_.Frame.consts_of_batch(batch, self.voxel_size)
            prepared = self.prepare_builder_inputs(batch, consts)      # what the builder needs that does not depend on the encoder
            imgs4 = F_.relayout_images(batch["src_imgs"][0])           # the frame's channels-last source images
        if two_streams:
            main.wait_stream(side)
            _record_on(main, consts, prepared, imgs4)
        neg = self._neg_ray(batch)
        frame = self.build_frame(batch, featmaps, consts, prepared, imgs4=imgs4)
        with torch.cuda.stream(side):
            rays = torch.cat([batch["ray_o"], batch["ray_d"], batch["near"].unsqueeze(-1), batch["far"].unsqueeze(-1)], dim=-1)[0]
            n = rays.shape[0]
            # Which 32 rays share a wavefront is the launch's choice (results do not depend on it): when the batch says which
            # pixels the rays are (`mask_at_box`, ZjumocapDataset.py:505), lay them out as 32x8-pixel patches so that the rays of
            # a workgroup hit the same cache lines.  A sharded frame cuts its round-robin bands from the same patch-major list.
            order = None
            if "mask_at_box" in batch:
                Hs, Ws = batch["src_imgs"].shape[-2:]
                m = batch["mask_at_box"].reshape(-1)
                if m.numel() == Hs * Ws and m.is_cuda:
                    # with early termination a wavefront's 32 rays are a compact 8x4-pixel block rather than a 32-pixel row: the
                    # rays of a block become opaque together far more often (bench frame: 43 % -> 35 % of the samples evaluated)
                    pw, ph = (8, 4) if self.early_term else (32, 8)
                    # (a mask that does not keep exactly the n pixels the rays belong to cannot order them: the kernels then leave
                    # list order -- decided on the device, reading the count on the host would be a synchronisation)
                    order = F_.patch_order_rays(m, Hs, Ws, n, patch_w=pw, patch_h=ph)
        if two_streams:
            main.wait_stream(side)
            _record_on(main, rays, order)
        p = Prefetched(batch=batch, frame=frame, rays=rays, order=order, n=n, neg=neg, ev0=ev0, ev1=ev1, group=group,
                       keep=(featmaps, consts, prepared, imgs4), own_encoder=own_encoder, enc_run=enc_run)
        if enc_run is None:
            p.flagged = False                      # nothing can have left a range
        else:
            self.__dict__.setdefault("_enc_outstanding", []).append(p)
        return p

    def host_consts(self, batch):
        """The frame's small constants (camera matrices, Rh, Th, bounds, out_sh) on the host: ONE device-to-host copy (none for
        a batch still on the CPU).  A caller that prefetches puts the result into batch["_gpnerf_consts"] while no per-ray
        kernel is in flight (evaluator.evaluate_loop does)."""
        return F_.Frame.consts_of_batch(batch, self.voxel_size)

    def _settle_encoder_verdicts(self):
        """Split-precision encoder only: the encoder's HIP graph has ONE range flag, so the verdict of every pass still in flight
        is read -- in the order the passes were enqueued, each after waiting for ITS encoder to finish (the event behind it, not
        the frame's per-ray kernel) -- before the graph is replayed again or a frame's result is trusted.  A raised flag is cleared
        once its frame has taken it.  (Round 5 kept one pending flag per module: render(next_batch=...) on a frame that was not
        itself prefetched then let two frames share it.)"""
        pending = self.__dict__.get("_enc_outstanding")
        while pending:
            q = pending.pop(0)
            q.ev1.synchronize()
            q.flagged = bool(q.enc_run.raised())
            if q.flagged:
                q.enc_run.clear()
            q.enc_run = None

    def prefetch(self, batch, after=None):
        """Enqueue everything of `batch`'s frame that comes before the per-ray kernel (`_produce`) on a stream of its own and return
        at once; `render(batch, prefetched=p)` then only launches the per-ray kernel.  In an evaluation loop this is called for
        frame t + 1 while frame t's per-ray kernel runs (`render(..., next_batch=...)` does that): the host's ~0.5 ms of enqueueing
        and the evaluator's work of frame t no longer sit between two frames' device work, and the next frame's encoder / builder
        launches start the moment frame t's persistent workgroups let go of CUs.  (They do not run BESIDE the per-ray kernel: its
        workgroups hold every CU's registers and LDS.  Leaving CUs free for them -- GPNERF_FLAG_RESERVE_CUS -- was measured and
        costs the kernel more than the overlap returns: profiles/r05/d_pipeline.txt.)  Same bits as the serial call.
        after: an event on the caller's stream behind which the batch's tensors are complete (default: everything enqueued on the
        caller's stream so far -- render(next_batch=...) passes the point BEFORE its own per-ray kernel, or the producers' first
        host-side wait would sit behind that kernel).
        Not in the reference (its loop is strictly serial: libs/trainers/BaseTrainer.py:255-280)."""
        if self.progressive or P_.resolve_group(self.shard_group) is not None:
            raise L.GpnerfError("prefetch() is for the dense single-GPU path")
        dev = batch["ray_o"].device
        cur = torch.cuda.current_stream(dev)
        prod = self.__dict__.get("_prod_stream")
        if prod is None or prod.device != dev:
            prod = self.__dict__["_prod_stream"] = torch.cuda.Stream(device=dev)
        t0 = time.time()
        if after is not None:
            prod.wait_event(after)
        else:
            prod.wait_stream(cur)                  # whatever produced the batch's tensors on the caller's stream
        with torch.cuda.stream(prod):
            p = self._produce(batch)
            p.done = torch.cuda.Event(enable_timing=True)
            p.done.record(prod)
        p.stream = prod
        p.host_s = time.time() - t0
        return p

    def render(self, batch, prefetched=None, next_batch=None):
        """`Renderer.render(batch)` (libs/renders/BaseRender.py:211-274).  prefetched / next_batch (not in the reference): see
        prefetch(); `next_batch`'s frame is prefetched right after this frame's per-ray kernel is enqueued and returned as
        ret["next_prefetched"]."""
        if self.progressive:
            return self.render_progressive(batch)
        if not self.nerfhead.use_rgbhead:
            raise L.GpnerfError("mesh extraction (use_rgbhead=False, BaseRender.py:255-272) is outside the per-ray render path")
        dev = batch["ray_o"].device
        main = torch.cuda.current_stream(dev)
        if prefetched is None:
            torch.cuda.synchronize(dev)
            te = time.time()
            p = self._produce(batch)
        else:
            p = prefetched
            if p.batch is not batch:
                raise L.GpnerfError("render(batch, prefetched=p): p was prefetched for another batch")
            te = time.time()
            main.wait_event(p.done)                # the per-ray kernel reads what the producer stream wrote
        frame, rays, order, n, neg, group = p.frame, p.rays, p.order, p.n, p.neg, p.group
        sharded = group is not None
        ready = None
        if next_batch is not None:                 # where the caller's stream stands before this frame's per-ray kernel
            ready = torch.cuda.Event()
            ready.record(main)

        def fn(r):
            # sharded: `r` is this rank's share, already in patch-major order
            return F_.render_fused(frame, r, self.n_samples, neg_ray=neg, early_term=self.early_term, term_eps=self.term_eps,
                                   split_f16=self.split_f16, ray_order=None if sharded else order, want=("weights", "z_vals", "rgb_in"),
                                   fold="keep" if self.fold_levels else False,
                                   reserve_cus=self.reserve_cus if (next_batch is not None and not sharded) else 0)

        # every map of the reference's dict travels in ONE packed all-gather; sharded_outputs = "pixels" keeps the exchange at
        # the 16 B/ray of rgb + depth (what an evaluation loop reads, libs/evaluators/if_nerf.py:50-56) and returns only those
        all_keys = ("rgb_map", "depth_map", "acc_map", "disp_map", "weights", "z_vals", "rgb_in_map")
        keys = P_.PIXEL_KEYS if (sharded and self.sharded_outputs == "pixels") else all_keys
        o = P_.render_sharded(fn, rays, keys=keys, group=group, order=order if sharded else None)
        nxt = self.prefetch(next_batch, after=ready) if next_batch is not None else None      # enqueued behind this frame's per-ray kernel
        if prefetched is None and nxt is None:
            torch.cuda.synchronize(dev)
        else:
            main.synchronize()                     # this frame only: the next frame's producers may still be running
        t2 = time.time()
        # etime = the encoder alone, rtime = everything else of the call (demo_render.py:441-446,494-497 keeps these two clocks;
        # BaseTrainer.py:276 sums rtime): the encoder's share is its device time between the two events.  With a prefetched frame
        # the producers ran before the call, on their own stream: rtime = the call's wall time (the per-ray kernel) + the DEVICE
        # time of everything the producer stream ran behind the encoder (volume builder, re-layouts, ray list: ev1 .. done), so
        # that it still measures what the reference's rtime measures -- everything of the frame but the encoder.
        etime = p.ev0.elapsed_time(p.ev1) * 1e-3
        rtime = max(0.0, (t2 - te) - etime) if prefetched is None else (t2 - te) + p.ev1.elapsed_time(p.done) * 1e-3
        if p.flagged is None:
            self._settle_encoder_verdicts()
        flagged = bool(p.flagged)
        if sharded and p.own_encoder and getattr(self.encoder, "precision", "split") != "fp32":
            # (split-precision encoder only) every rank encoded ITS views: one rank's flag is every rank's (the re-render below issues collectives, so the ranks
            # must take the same branch) -- one 4-byte all-reduce per frame
            fl = torch.tensor([1 if flagged else 0], dtype=torch.int32, device=rays.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(fl, op=dist.ReduceOp.MAX, group=group)
            flagged = p.flagged = bool(int(fl.item()))
        if flagged:
            # this frame drove the split-f16 encoder out of its operand range (its feature maps are NaN-ridden): encode it in the
            # exact form and render again from those maps; the wasted per-ray pass stays in rtime (not the wasted encoder pass,
            # which a normal frame's rtime does not hold either), the exact encoder's time goes to etime
            t3 = time.time()
            fm = self.encoder.forward_exact(batch["src_imgs"].squeeze(0))
            torch.cuda.synchronize(dev)
            t_exact = time.time() - t3
            ret = self.render(dict(batch, featmaps=fm))
            ret["etime"] = t_exact
            ret["rtime"] = ret["rtime"] + rtime
            if nxt is not None:
                ret["next_prefetched"] = nxt
            return ret
        if keys is P_.PIXEL_KEYS:
            ret = {"rgb_map": o["rgb_map"].view(1, n, 3), "depth_map": o["depth_map"].view(1, n, 1), "etime": etime, "rtime": rtime}
        else:
            ret = {
                "rgb_map": o["rgb_map"].view(1, n, 3), "disp_map": o["disp_map"].view(1, n, 1),
                "acc_map": o["acc_map"].view(1, n, 1), "depth_map": o["depth_map"].view(1, n, 1),
                "alpha": o["weights"].view(1, n, -1), "z_vals": o["z_vals"].view(1, n, -1),
                "rgb_in_map": o["rgb_in_map"].view(1, n, 9),
                # BaseRender.render returns neither; the evaluation loop reads ret["rtime"] (BaseTrainer.py:276), which only the
                # demo renderer provides: encoder time and everything after it, each on its own clock (demo_render.py:441-446,494-497)
                "etime": etime, "rtime": rtime,
            }
        if nxt is not None:
            ret["next_prefetched"] = nxt
        return ret


class Prefetched:
    """What Renderer._produce leaves for the per-ray launch: the frame, its ray list and patch order, the encoder's two events,
    and (prefetch()) the event that ends the producer stream's work.  `flagged`: the encoder's range verdict once it is known."""

    def __init__(self, **kw):
        self.done, self.stream, self.host_s, self.flagged, self.enc_run = None, None, 0.0, None, None
        self.__dict__.update(kw)


def _record_on(stream, *items):
    """tensor.record_stream(stream) for every CUDA tensor in `items` (tensors, or lists / tuples / dicts of them, nested)"""
    for it in items:
        if isinstance(it, torch.Tensor):
            if it.is_cuda:
                it.record_stream(stream)
        elif isinstance(it, dict):
            _record_on(stream, *it.values())
        elif isinstance(it, (list, tuple)):
            _record_on(stream, *it)


def build_render(cfg, progressive=False):
    """Same cfg keys as BaseRender.py:367-403; encoder / head come from the plugins cfg names.  One optional key the reference
    does not have: `cfg.render.shard` ("world" or a ProcessGroup) opts into sharding one frame over a process group."""
    encoder = getattr(impm(cfg.encoder.file), "build_encoder")(cfg)
    nerfhead = getattr(impm(cfg.head.file), "build_head")(cfg)
    if not hasattr(nerfhead, "head_blob"):
        raise L.GpnerfError(f"head.file='{cfg.head.file}' does not provide the HIP head; use head.file 'hip_head'")
    neg_ray_train = "thuman" in cfg.dataset.train.name
    neg_ray_val = "thuman" in cfg.dataset.test.name
    mesh_th = -1 if cfg.head.rgb.use_rgbhead else 1.0 / cfg.test.mesh_th
    return Renderer(encoder=encoder, nerfhead=nerfhead, is_train=False, neg_ray_train=neg_ray_train,
                    neg_ray_val=neg_ray_val, n_rays=cfg.train.n_rays, n_samples=cfg.train.n_samples,
                    voxel_size=cfg.dataset.voxel_size, chunk=cfg.dataset.test.chunk, mesh_th=mesh_th, progressive=progressive,
                    shard_group=getattr(getattr(cfg, "render", None), "shard", None) or None)
