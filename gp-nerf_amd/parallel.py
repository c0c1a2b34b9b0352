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


def gather_frame(local, plan, keys, group=None, buffer=None):
    """ONE collective: every rank contributes its packed [share, C] maps, every rank gets the frame's maps in ray order.
    buffer: optional preallocated [world * share, C] tensor (the bench passes one; Renderer.render lets the allocator cache it)."""
    packed, cols = pack_maps(local, keys)
    if buffer is None or buffer.shape != (plan.world * plan.share, packed.shape[1]) or buffer.device != packed.device:
        buffer = torch.empty((plan.world * plan.share, packed.shape[1]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(buffer, packed.contiguous(), group=group)
    full = plan.unpermute(buffer)
    res = {}
    for k, (a, b, nd, dt) in cols.items():
        v = full[:, a:b] if nd > 1 else full[:, a]
        res[k] = v if dt == torch.float32 else v.to(dt)
    return res


def render_sharded(render_fn, rays, keys=("rgb_map", "depth_map", "acc_map", "disp_map"), group=None, band=INTERLEAVE_BAND):
    """Strong scaling of one frame: every rank renders its share of `rays` with `render_fn(rays_share) -> dict` and receives
    the requested maps of the whole frame (one packed all-gather).  With no process group this is just render_fn(rays)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return render_fn(rays)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    plan = plan_for(rays.shape[0], world, rays.device, band)
    local = render_fn(plan.take(rays, rank))
    return gather_frame(local, plan, keys, group)


def encode_views_sharded(encoder, src_imgs, group=None):
    """The per-frame image encoder with one source view per rank (views are independent: the encoder normalises per image,
    libs/encoders/UNet.py:40-53): rank r encodes view r mod V and one all-gather hands every rank all V feature maps.
    Falls back to the plain call without a process group or with fewer ranks than views."""
    V = src_imgs.shape[0]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < V:
        return encoder(src_imgs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = encoder(src_imgs[rank % V: rank % V + 1]).contiguous()          # [1,C,h,w]
    allv = torch.empty((world,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(allv, mine, group=group)
    return allv[:V]
