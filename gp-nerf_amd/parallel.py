"""Ray sharding over the GPUs of one node.

Rays are independent given the per-frame constants (BaseRender.py:110-157 touches no
cross-ray state), so each rank renders a contiguous block of the ray list and the only
exchange is one all-gather of the packed pixels per frame.  The reference has no
collective on this path (its inference is single-device, tools/inference.py:62); this is
the multi-GPU form of its serial chunk loop (BaseRender.py:160-184).

Backend: torch.distributed -- "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist

TILE = 32  # rays one wavefront renders together; shard boundaries stay tile-aligned
INTERLEAVE_BAND = 2048  # rays per round-robin band of render_sharded: 64 tiles = 8 workgroups of the fused kernel


def shard_bounds(n_rays, world, tile=TILE):
    """Contiguous, tile-aligned [start, end) per rank; sizes differ by at most one tile."""
    tiles = (n_rays + tile - 1) // tile
    base, rem = divmod(tiles, world)
    bounds, s = [], 0
    for r in range(world):
        t = base + (1 if r < rem else 0)
        e = min(n_rays, s + t * tile)
        bounds.append((s, e))
        s = e
    return bounds


def pack_pixels(out):
    """[n,4] = rgb(3) + depth(1): the per-frame payload of the all-gather (16 B/ray)."""
    return torch.cat([out["rgb_map"], out["depth_map"][:, None]], dim=1).contiguous()


def all_gather_pixels(out, gathered):
    """Equal-sized shards (bench, weak scaling): gathered [world, n_local, 4]."""
    packed = pack_pixels(out)
    dist.all_gather_into_tensor(gathered.view(-1, 4), packed)
    return gathered


def all_gather_ragged(local, bounds, group=None):
    """All-gather shards of unequal length along dim 0; returns the concatenation on every rank."""
    world = len(bounds)
    longest = max(e - s for s, e in bounds)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: e - s] for p, (s, e) in zip(parts, bounds)], dim=0)


def interleaved_indices(n_rays, world, band=INTERLEAVE_BAND):
    """Round-robin bands of `band` rays (a multiple of the 32-ray tile): rank r owns bands r, r + world, ...  Per-ray cost
    varies smoothly over the image (empty space, early termination, culling), so interleaving evens the ranks' work where
    contiguous blocks do not.  Returns one index tensor per rank; together they cover 0..n_rays-1 once."""
    idx = torch.arange(n_rays)
    owner = (idx // band) % world
    return [idx[owner == r] for r in range(world)]


def render_sharded(render_fn, rays, keys=("rgb_map", "depth_map", "acc_map", "disp_map"), group=None, interleave=True):
    """Strong scaling of one frame: every rank renders its share of `rays` with `render_fn(rays_share) -> dict`
    and receives the full maps.  With no process group this is just render_fn(rays).
    interleave: round-robin bands (load balance) instead of one contiguous block per rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return render_fn(rays)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = rays.shape[0]
    if interleave:
        shares = interleaved_indices(n, world)
        mine = shares[rank].to(rays.device)
        local = render_fn(rays.index_select(0, mine))
        sizes, start = [int(s.numel()) for s in shares], 0
        bounds = []
        for sz in sizes:
            bounds.append((start, start + sz))
            start += sz
        order = torch.cat(shares).to(rays.device)              # gathered row i holds ray order[i]
    else:
        bounds = shard_bounds(n, world)
        s, e = bounds[rank]
        local = render_fn(rays[s:e])
        order = None
    full = {}
    for k in keys:
        v = local[k]
        v2 = v if v.dim() > 1 else v[:, None]
        g = all_gather_ragged(v2.contiguous(), bounds, group)
        if order is not None:
            out = torch.empty_like(g)
            out.index_copy_(0, order, g)
            g = out
        full[k] = g if v.dim() > 1 else g[:, 0]
    return full
