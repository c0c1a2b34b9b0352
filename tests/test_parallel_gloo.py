"""CPU: ray sharding + the packed all-gather over torch.distributed (gloo, world_size 2, 4 and 8)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_bounds_are_tile_aligned_and_cover():
    par = importlib.import_module("gp-nerf_amd.parallel")
    for n in (0, 1, 31, 32, 33, 667, 4096, 262144, 262145):
        for world in (1, 2, 3, 4, 8):
            b = par.shard_bounds(n, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
            for (s, e), (s2, e2) in zip(b, b[1:]):
                assert e == s2
                assert s2 % 32 == 0 or s2 == e2      # non-empty shards start on a 32-ray tile
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 32 + 31


def test_shard_plan_partitions_the_rays_into_equal_shares():
    par = importlib.import_module("gp-nerf_amd.parallel")
    for n in (1, 2047, 2048, 2049, 10000, 262144, 1048576):
        for world in (1, 2, 3, 8):
            plan = par.ShardPlan(n, world, torch.device("cpu"))
            assert plan.index.shape == (world, plan.share) and plan.share % par.INTERLEAVE_BAND == 0
            assert plan.share * world >= n and plan.share * world - n < world * par.INTERLEAVE_BAND + par.INTERLEAVE_BAND * world
            assert sum(plan.n_valid) == n
            # every ray is rendered by exactly the rank its band belongs to, and comes back to its own row
            flat = plan.index.reshape(-1)
            assert torch.equal(flat[plan.inverse], torch.arange(n))
            owner = (torch.arange(n) // par.INTERLEAVE_BAND) % world
            assert torch.equal(plan.inverse // plan.share, owner)
            # round-trip of a payload through take -> (all-gather layout) -> unpermute
            payload = torch.arange(n, dtype=torch.float32)[:, None] * torch.tensor([[1.0, -2.0]])
            gathered = torch.cat([plan.take(payload, r) for r in range(world)], 0)
            assert torch.equal(plan.unpermute(gathered), payload)
    assert par.plan_for(1000, 4, torch.device("cpu")) is par.plan_for(1000, 4, torch.device("cpu"))      # cached per frame size


def _fake_render(rays):
    # any per-ray function stands in for the kernel: shard-invariance is what is under test
    s = rays.sum(1)
    S = 5
    return {"rgb_map": torch.stack([s, s * 2, s * 3], 1), "depth_map": s + 1, "acc_map": s * 0 + 1, "disp_map": 1 / (s + 1),
            "weights": s[:, None] * torch.arange(S)[None].float(), "z_vals": s[:, None] + torch.arange(S)[None].float(),
            "rgb_in_map": s[:, None] * torch.ones(9)[None], "ray_mask": (s > 4).to(torch.uint8)}


class _Enc(torch.nn.Module):
    calls = 0

    def out_shape(self, H, W):
        return 2, H // 2, W // 2

    def forward(self, x):              # per-image function: [v,3,H,W] -> [v,2,H/2,W/2]
        _Enc.calls += x.shape[0]
        return torch.stack([x.mean(1)[:, ::2, ::2], x.amax(1)[:, ::2, ::2] - x.mean(dim=(1, 2, 3))[:, None, None]], 1)


def _worker(rank, world, port, n, q, light=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("gp-nerf_amd.parallel")
    g = torch.Generator().manual_seed(7)
    rays = torch.rand((n, 8), generator=g)
    ref = _fake_render(rays)
    ok = True
    # the 16 B/ray form, the default key set, and every map Renderer.render returns: one collective each, bit-equal
    # (light: the bench's two forms on the library's own band, one plain and one permuted list -- the 1 048 576-ray frame of
    # BASELINE.json configs[3] on eight ranks)
    for keys in ((par.PIXEL_KEYS, tuple(ref)) if light else (par.PIXEL_KEYS, ("rgb_map", "depth_map", "acc_map", "disp_map"), tuple(ref))):
        for band in ((par.INTERLEAVE_BAND,) if light else (par.INTERLEAVE_BAND, 64)):
            for order in ((torch.randperm(n, generator=g),) if light and keys is par.PIXEL_KEYS else (None, torch.randperm(n, generator=g))):      # bands cut from a permuted (patch-major) list
                full = par.render_sharded(_fake_render, rays, keys=keys, band=band, group="world", order=order)
                ok = ok and set(full) == set(keys)
                ok = ok and all(torch.equal(full[k], ref[k]) and full[k].dtype == ref[k].dtype and full[k].shape == ref[k].shape for k in keys)
    # sharding is opt-in: without a group nothing is split, whatever process groups exist (under the reference's
    # DistributedSampler the ranks hold different frames)
    seen = []
    par.render_sharded(lambda r: seen.append(r.shape[0]) or _fake_render(r), rays)
    ok = ok and seen == [n]
    # preallocated gather buffer, as the bench uses it
    plan = par.plan_for(n, world, rays.device)
    buf = torch.empty((world * plan.share, 4))
    full = par.gather_frame(_fake_render(plan.take(rays, rank)), plan, par.PIXEL_KEYS, buffer=buf)
    ok = ok and torch.equal(full["rgb_map"], ref["rgb_map"]) and torch.equal(full["depth_map"], ref["depth_map"])
    # equal-sized contiguous shards, the weak-scaling collective
    s, e = par.shard_bounds(n - n % (32 * world), world)[rank]
    local = _fake_render(rays[s:e])
    gathered = torch.empty((world, e - s, 4))
    par.all_gather_pixels(local, gathered)
    exp = par.pack_pixels(_fake_render(rays[: (e - s) * world]))
    ok = ok and torch.equal(gathered.view(-1, 4), exp)
    # one source view per rank
    imgs = torch.rand((3, 3, 8, 12), generator=g)
    enc = _Enc()
    # every source view is encoded by exactly one rank and broadcast in its physical [h,w,C] layout; ranks beyond the V-th encode nothing
    fm = par.encode_views_sharded(enc, imgs, group=dist.group.WORLD)
    mine = sum(1 for v in range(3) if par.view_owner(v, 3, world) == rank)
    ok = ok and _Enc.calls == mine and mine == ({0: 2, 1: 1}[rank] if world == 2 else (1 if rank < 3 else 0))
    ok = ok and torch.equal(fm, _Enc()(imgs)) and fm.shape == (3, 2, 4, 6) and fm.permute(0, 2, 3, 1).is_contiguous()
    calls = _Enc.calls
    ok = ok and torch.equal(par.encode_views_sharded(enc, imgs), fm) and _Enc.calls == calls + 3          # no group: the plain call
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 667), (2, 9000), (4, 4096), (4, 20001), (8, 70001), (8, 1048576)])
def test_render_sharded_equals_unsharded(world, n):
    """world 8 on the 1 048 576-ray plan = the frame the driver's SCALE run shards over a node's eight GPUs (BASELINE.json configs[3]):
    512 bands of 2 048 rays, 64 per rank, ONE collective per frame (the multi-GPU path has never seen RCCL; its
    partition, exchange and re-assembly are what can be held to the bit on CPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, n > 500000)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
