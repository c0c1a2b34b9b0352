"""Ray sharding over the GPUs of one node: strong scaling of ONE frame.

Rays are independent given the per-frame constants (BaseRender.py:110-157 touches no cross-ray state), so the ranks
split the ray list and the only exchange is ONE all-gather of equal-sized packed shares per frame.  The reference has no
collective on this path (its inference is single-device, tools/inference.py:62); this is the multi-GPU form of its serial
chunk loop (BaseRender.py:160-184).

Partition (`ShardPlan`): round-robin bands of `band` rays -- rank r owns bands r, r + G, r + 2G, ... -- because per-ray
cost varies smoothly over the image (empty space, early termination, culling), so interleaving evens the ranks' work where
one contiguous block per rank does not.  Every rank gets the same number of bands; the slots of the last round that fall
beyond the ray list are padding (they re-render the last ray and are dropped when the frame is re-assembled), so the
collective is a plain `all_gather_into_tensor` of a preallocated [share, C] buffer: no ragged lists, no per-key calls,
no host-side index building per frame (the index tensors live on the device and are cached per (n_rays, world)).

Backend: torch.distributed -- "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

Sharding is OPT-IN: every entry point takes the process group explicitly and does nothing collective without one
(`group=None`).  It is meant for ranks that hold the SAME batch and want one frame faster.  The reference's own distributed
flow (tools/train.py: DistributedDataParallel + a DistributedSampler for the evaluation loader) hands every rank a DIFFERENT
frame; sharding a frame there would mix the ranks' images, which is why the existence of a default process group alone never
switches it on.
"""
import torch
import torch.distributed as dist

TILE = 32  # rays one wavefront renders together; bands are a multiple of it
INTERLEAVE_BAND = 2048  # rays per round-robin band: 64 tiles = 8 workgroups of the fused kernel

# widths of the maps Renderer.render returns (BaseRender.py:148-156); weights / z_vals are [N, S]
PIXEL_KEYS = ("rgb_map", "depth_map")


def shard_bounds(n_rays, world, tile=TILE):
    """Contiguous, tile-aligned [start, end) per rank; sizes differ by at most one tile (the weak-scaling bench's bands)."""
    tiles = (n_rays + tile - 1) // tile
    base, rem = divmod(tiles, world)
    bounds, s = [], 0
    for r in range(world):
        t = base + (1 if r < rem else 0)
        e = min(n_rays, s + t * tile)
        bounds.append((s, e))
        s = e
    return bounds


class ShardPlan:
    """Which rays each of `world` ranks renders, and how the gathered shares go back into ray order.

    share            rays per rank (equal for all ranks, a multiple of the band)
    index[r]         int64 [share]: the rays of rank r (padding slots repeat ray n_rays - 1)
    inverse          int64 [n_rays]: row of the gathered [world * share, C] buffer that holds ray i
    """

    def __init__(self, n_rays, world, device, band=INTERLEAVE_BAND):
        if n_rays < 1 or world < 1 or band % TILE:
            raise ValueError("ShardPlan: need n_rays >= 1, world >= 1 and a band that is a multiple of the 32-ray tile")
        self.n_rays, self.world, self.band = int(n_rays), int(world), int(band)
        bands = -(-self.n_rays // self.band)
        rounds = -(-bands // self.world)
        self.share = rounds * self.band
        # slot s of rank r is ray ((s // band) * world + r) * band + s % band
        s = torch.arange(self.share, device=device)
        r = torch.arange(self.world, device=device)[:, None]
        ray = ((s // self.band)[None, :] * self.world + r) * self.band + (s % self.band)[None, :]      # [world, share]
        valid = ray < self.n_rays
        self.index = torch.where(valid, ray, torch.full_like(ray, self.n_rays - 1))
        flat = torch.arange(self.world * self.share, device=device).view(self.world, self.share)
        self.inverse = torch.empty((self.n_rays,), dtype=torch.int64, device=device)
        self.inverse[ray[valid]] = flat[valid]
        self.n_valid = valid.sum(dim=1).tolist()

    def take(self, rays, rank):
        """This rank's rays [share, ...] (device-side gather, padding included)."""
        return rays.index_select(0, self.index[rank])

    def unpermute(self, gathered):
        """gathered [world * share, C] (rank-major, as all_gather_into_tensor lays it out) -> [n_rays, C] in ray order."""
        return gathered.index_select(0, self.inverse)


_plans = {}


def plan_for(n_rays, world, device, band=INTERLEAVE_BAND):
    key = (int(n_rays), int(world), str(device), int(band))
    if key not in _plans:
        if len(_plans) > 16:
            _plans.clear()
        _plans[key] = ShardPlan(n_rays, world, device, band)
    return _plans[key]


def pack_pixels(out):
    """[n,4] = rgb(3) + depth(1): the per-frame payload of the all-gather (16 B/ray)."""
    return torch.cat([out["rgb_map"], out["depth_map"][:, None]], dim=1)


def pack_maps(out, keys):
    """One [n, C] buffer holding the requested maps side by side, and the column slices to take them apart again."""
    cols, parts, c = {}, [], 0
    for k in keys:
        v = out[k]
        v2 = v if v.dim() > 1 else v[:, None]
        v2 = v2 if v2.dtype == torch.float32 else v2.float()
        cols[k] = (c, c + v2.shape[1], v.dim(), v.dtype)
        parts.append(v2)
        c += v2.shape[1]
    return torch.cat(parts, dim=1), cols


def all_gather_pixels(out, gathered):
    """Equal-sized shards (weak-scaling bench): gathered [world, n_local, 4]."""
    dist.all_gather_into_tensor(gathered.view(-1, 4), pack_pixels(out))
    return gathered


def gather_frame(local, plan, keys, group=None, buffer=None, row_of_ray=None):
    """ONE collective: every rank contributes its packed [share, C] maps, every rank gets the frame's maps in ray order.
    buffer: optional preallocated [world * share, C] tensor (the bench passes one; Renderer.render lets the allocator cache it).
    row_of_ray: int64 [n_rays], row of the gathered buffer holding ray i, when the bands were cut from a permuted list
    (default: plan.inverse)."""
    packed, cols = pack_maps(local, keys)
    if buffer is None or buffer.shape != (plan.world * plan.share, packed.shape[1]) or buffer.device != packed.device:
        buffer = torch.empty((plan.world * plan.share, packed.shape[1]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(buffer, packed.contiguous(), group=group)
    full = plan.unpermute(buffer) if row_of_ray is None else buffer.index_select(0, row_of_ray)
    res = {}
    for k, (a, b, nd, dt) in cols.items():
        v = full[:, a:b] if nd > 1 else full[:, a]
        res[k] = v if dt == torch.float32 else v.to(dt)
    return res


def resolve_group(group):
    """None -> None (no sharding).  "world" / True -> the default process group (it must be initialised).  A ProcessGroup -> itself.
    A group of one rank resolves to None."""
    if group is None or group is False:
        return None
    if group is True or (isinstance(group, str) and group == "world"):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("sharding over the default process group was asked for, but torch.distributed is not initialised")
        group = dist.group.WORLD
    return None if dist.get_world_size(group) == 1 else group


def render_sharded(render_fn, rays, keys=("rgb_map", "depth_map", "acc_map", "disp_map"), group=None, band=INTERLEAVE_BAND, order=None):
    """Strong scaling of one frame: every rank renders its share of `rays` with `render_fn(rays_share) -> dict` and receives
    the requested maps of the whole frame (one packed all-gather).  Every rank of `group` must call this with the SAME rays.
    group=None: no collective, this is just render_fn(rays).
    order: optional int64 permutation of the ray list (Renderer.render's patch-major order): the bands are cut from
    rays[order], so that a rank's 32-ray tiles are the same compact pixel patches the unsharded launch renders; the maps
    come back in the ORIGINAL list order (the two permutations are composed into one index_select)."""
    group = resolve_group(group)
    if group is None:
        return render_fn(rays)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    plan = plan_for(rays.shape[0], world, rays.device, band)
    if order is None:
        local = render_fn(plan.take(rays, rank))
        return gather_frame(local, plan, keys, group)
    order = order.long()
    local = render_fn(rays.index_select(0, order.index_select(0, plan.index[rank])))
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=order.device)
    return gather_frame(local, plan, keys, group, row_of_ray=plan.inverse.index_select(0, inv))


def view_owner(v, V, world):
    """Rank that encodes source view v: with at least V ranks view v belongs to rank v (the others only receive), with fewer the
    views go round-robin (two ranks, three views: rank 0 encodes views 0 and 2, rank 1 view 1)."""
    return v if world >= V else v % world


def encode_views_sharded(encoder, src_imgs, group=None, out_shape=None, encode_fn=None):
    """The per-frame image encoder with the source views dealt out over the ranks (views are independent: the encoder
    normalises per image, libs/encoders/UNet.py:40-53).  Every view is encoded by exactly one rank (`view_owner`) and handed
    to the others by a broadcast from its owner into one [V,h,w,C] buffer -- the PHYSICAL layout of the encoder's
    channels-last result, so nothing is re-laid out on either side and `Frame` takes the returned tensor by pointer.  Ranks
    beyond the V-th encode nothing.  Every rank of `group` must hold the same `src_imgs`.
    group=None: the plain call.  out_shape: (C, h, w) of the encoder's result when the caller knows it; otherwise
    `encoder.out_shape(H, W)` is asked, and an encoder without it makes every rank encode its own copy (no collective).
    encode_fn: how to call the encoder on a [v,3,H,W] subset (default `encoder(...)`; Renderer.render passes the HIP-graph replay
    `encoder.forward_graphed`, the same bits with one host call instead of ~20 launches per view)."""
    group = resolve_group(group)
    if encode_fn is None:
        encode_fn = encoder
    if group is None:
        return encode_fn(src_imgs)
    V = src_imgs.shape[0]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if out_shape is None and hasattr(encoder, "out_shape"):
        out_shape = encoder.out_shape(int(src_imgs.shape[-2]), int(src_imgs.shape[-1]))
    if out_shape is None:
        return encode_fn(src_imgs)
    C, h, w = (int(x) for x in out_shape)
    mine = [v for v in range(V) if view_owner(v, V, world) == rank]
    allv = torch.empty((V, h, w, C), dtype=torch.float32, device=src_imgs.device)
    if mine:
        out = encode_fn(src_imgs[mine])
        phys = out.permute(0, 2, 3, 1)                       # the channels-last result viewed in its physical order: no copy
        if tuple(phys.shape[1:]) != (h, w, C):
            raise RuntimeError(f"encoder.out_shape promised {(C, h, w)}, the encoder returned {tuple(out.shape[1:])}")
        allv[mine] = phys.to(torch.float32)
    ranks = dist.get_process_group_ranks(group)
    work = [dist.broadcast(allv[v], src=ranks[view_owner(v, V, world)], group=group, async_op=True) for v in range(V)]
    for wk in work:
        wk.wait()
    return allv.permute(0, 3, 1, 2)                          # logical [V,C,h,w] with channels-last strides
