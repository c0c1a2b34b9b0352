"""GPU, two ranks: Renderer.render sharded over a process group with the REAL kernels, against the same frame rendered
unsharded (the serial chunk loop of libs/renders/BaseRender.py:160-184 as two ranks x one launch).

With at least two devices visible the ranks take one GPU each and the backend is "nccl" (= RCCL over xGMI): the packed
all-gather of the maps and the broadcasts of the encoder's feature maps run on RCCL.  On a one-GPU box the same two ranks
share cuda:0 over "gloo" (which stages device tensors through the host): everything but RCCL itself is exercised -- the
opt-in group, the patch-major bands, the view split of the encoder, the re-assembly."""
import importlib
import os
import socket
import sys
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, backend, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gp-nerf_amd", "plugins")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        dev = torch.device("cuda", rank if backend == "nccl" else 0)
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        syn = importlib.import_module("gp-nerf_amd.synthetic")
        hip_render = importlib.import_module("hip_render")
        c = NS(encoder=NS(file="hip_encoder", name="resnet34", out_ch=32),
               head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32] * 4)),
               dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3),
               train=NS(n_rays=1024, n_samples=32), test=NS(mesh_th=50))
        # every rank builds the same frame from the same seeds: 128x128 pixels, a third of them hit the box -> three bands, the last ragged
        sc = syn.make_scene(H=128, W=128, seed=12, focal_mul=5.0, pose="random", aabb_half=(0.12, 0.16, 0.05), bias_std=0.1, sigma_bias=0.3)
        sc["src_imgs"] = syn.make_encoder_images(128, 128, 12)[None]
        r = hip_render.build_render(c).to(dev).eval()
        sd = r.state_dict()
        for k, v in sc["head"].items():
            sd["nerfhead." + k] = torch.from_numpy(v.copy())
        r.load_state_dict(sd, strict=True)
        r.encoder.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(12).items()}, strict=True)
        keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk", "mask_at_box")
        b = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in keys}
        b["volumes"] = [torch.from_numpy(v).to(dev) for v in sc["volumes"]]
        n = sc["ray_o"].shape[1]
        assert r.shard_group is None
        with torch.no_grad():
            whole = r.render(b)                                   # a default group exists, nothing was asked for: no collective
            fm_whole = r.encoder(b["src_imgs"][0])
            r.shard_group = "world"
            part = r.render(b)
            fm_part = importlib.import_module("gp-nerf_amd.parallel").encode_views_sharded(r.encoder, b["src_imgs"][0], group="world")
            r.sharded_outputs = "pixels"
            pix = r.render(b)
        torch.cuda.synchronize()
        bad = []

        def same(a, b):          # bit-equal, NaNs (disp of a ray with acc == 0) in the same places
            return a.shape == b.shape and torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))

        if not torch.equal(fm_whole, fm_part) or not fm_part.is_contiguous(memory_format=torch.channels_last):
            bad.append(("featmaps", float((fm_whole - fm_part).abs().max())))
        for k in ("rgb_map", "depth_map", "acc_map", "disp_map", "alpha", "z_vals", "rgb_in_map"):
            if not same(part[k], whole[k]):
                bad.append((k, float(torch.nan_to_num(part[k] - whole[k]).abs().max())))
        if set(pix) != {"rgb_map", "depth_map", "etime", "rtime"} or not torch.equal(pix["rgb_map"], whole["rgb_map"]) or not torch.equal(pix["depth_map"], whole["depth_map"]):
            bad.append(("pixels form", 0.0))
        q.put((rank, n, bad))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                        # report instead of hanging the parent on q.get
        import traceback
        q.put((rank, -1, [("exception", traceback.format_exc()[-1500:])]))


def test_two_ranks_render_one_frame_bit_equal_to_one_rank():
    import torch.multiprocessing as mp
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=300) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    print(f"backend {backend}: {res}")
    assert [r for r, _, _ in res] == [0, 1]
    assert all(n > 4096 for _, n, _ in res), res
    assert all(not bad for _, _, bad in res), res
