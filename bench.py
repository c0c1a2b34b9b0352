#!/usr/bin/env python3
"""Headline benchmark: rays/s of the fused per-ray render path on synthetic 512x512x64 frames.

    python bench.py --gpus N --steps K --warmup W
(N > 1: launched by torch.distributed.run, one rank per GPU over RCCL.)

A "step" renders one frame's rays per GPU through gpnerf_render_fused (sample -> gather -> MLP ->
composite) and, for N > 1, all-gathers the packed pixels (rgb + depth) of every rank.  Inputs are
resident in HBM before the timed region.  Weak scaling: every rank renders its own 512x512x64
band of an (N*512)x512 image, so value = N * 262144 rays / step time.

One JSON line is printed by rank 0; see DESIGN.md for the accounting behind `roofline`.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 110848           # 2*MAC of the reference's dense layers, V=3, C=32
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="frame is size x size rays")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--fill", default="full", choices=["full", "survey"],
                    help="full: every pixel's ray crosses the SMPL bound (N = size^2); survey: f = 1.05 W (SURVEY.md §8d)")
    ap.add_argument("--early-term", action="store_true", help="config 3: wave-level early ray termination")
    ap.add_argument("--term-eps", type=float, default=1e-5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--split-f16", action="store_true",
                    help="dense layers on f16 MFMA with fp32 operands split into hi+lo (GPNERF_FLAG_SPLIT_F16)")
    ap.add_argument("--occ-cull", action="store_true",
                    help="progressive sample culling (demo_render.py semantics) on a sparse synthetic pyramid")
    ap.add_argument("--occupancy", type=float, default=None, help="fraction of coarse volume blocks that are occupied")
    ap.add_argument("--ray-order", default="patch", choices=["patch", "raster"],
                    help="patch: 32x8-pixel workgroup tiles (gpnerf_render_fused's ray_order); raster: the list as given")
    ap.add_argument("--patch", default="32x8", help="WxH of the patches of --ray-order patch (W*H a multiple of 32: one patch row or one whole patch per wavefront)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # GPNERF_BENCH_BACKEND=gloo is a dry run of the N>1 flow on a box with fewer GPUs than ranks (ranks share devices and
    # the all-gather is staged through the host); every measured run uses "nccl" (= RCCL) with one GPU per rank.
    backend = os.environ.get("GPNERF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    fm = importlib.import_module("gp-nerf_amd.frame")
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    par = importlib.import_module("gp-nerf_amd.parallel")

    H = W = args.size
    S = args.samples
    sc = syn.make_scene(H=H, W=W, seed=args.seed, fill=args.fill, pose="identity", vol_occupancy=args.occupancy)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    blob = fm.pack_head(sc["head"], dev)
    frame = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]),
                     t(sc["src_poses"][0]), sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], blob)
    rays_h = np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32)
    rays = t(rays_h)                       # this rank's band: weak scaling, same ray count on every rank
    n_local = rays.shape[0]
    order = None
    if args.ray_order == "patch":
        pw, ph = (int(v) for v in args.patch.split("x"))
        order = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], H, W, patch_w=pw, patch_h=ph)).to(dev)
    n_total = n_local * world
    torch.cuda.synchronize()

    want = ()                              # headline outputs only: rgb, depth, acc, disp
    gathered = torch.empty((world, n_local, 4), device=dev if backend == "nccl" else "cpu") if world > 1 else None
    k_start = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    k_stop = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            k_start[i].record()
        out = fm.render_fused(frame, rays, S, early_term=args.early_term, term_eps=args.term_eps, want=want, ray_order=order, occ_cull=args.occ_cull, split_f16=args.split_f16)
        if i is not None:
            k_stop[i].record()
        if world > 1:
            if backend == "nccl":
                par.all_gather_pixels(out, gathered)
            else:
                par.all_gather_pixels({k: v.cpu() for k, v in out.items() if k in ("rgb_map", "depth_map")}, gathered)
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(k_start, k_stop)]))

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt
        flops_per_launch = float(n_local) * S * FLOP_PER_SAMPLE
        achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "rays_per_sec", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_frame": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{H}x{W} frame, {S} samples/ray, fused HIP render kernel, synthetic SMPL bound + random "
                                   f"feature volume (BASELINE.json configs[{2 if args.early_term else 1}])",
                       "rays_per_gpu": int(n_local), "rays_total": int(n_total), "samples_per_ray": S, "fill": args.fill, "ray_order": args.ray_order,
                       "early_term": bool(args.early_term), "occ_cull": bool(args.occ_cull), "split_f16": bool(args.split_f16), "vol_occupancy": args.occupancy, "out_sh_dhw": [int(x) for x in sc["out_sh"][0]],
                       "parallelism": f"ray bands over {world} GPU(s), all-gather of rgb+depth" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": measured_traffic(args),
                         "kernel": "render_fused_kernel", "kernel_ms": kernel_ms,
                         "flop_per_launch": flops_per_launch},
        }
        if world == 1 and not args.split_f16:
            # the optional split-precision mode of the same kernel (GPNERF_FLAG_SPLIT_F16), measured after the headline
            # region: f16 hi/lo MFMAs with f32 accumulation, same parity bound; NOT part of `value`
            for _ in range(args.warmup):
                fm.render_fused(frame, rays, S, want=want, ray_order=order, split_f16=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.steps):
                alt = fm.render_fused(frame, rays, S, want=want, ray_order=order, early_term=args.early_term,
                                      term_eps=args.term_eps, occ_cull=args.occ_cull, split_f16=True)
            e1.record()
            torch.cuda.synchronize()
            alt_ms = e0.elapsed_time(e1) / args.steps
            line["split_f16_mode"] = {
                "value": n_local / (alt_ms * 1e-3), "unit": "rays/s", "ms_per_step": alt_ms,
                "max_abs_vs_f32_path": {"rgb": float((alt["rgb_map"] - out["rgb_map"]).abs().max()),
                                        "depth": float((alt["depth_map"] - out["depth_map"]).abs().max())},
                "note": "dense layers as 3 x v_mfma_f32_32x32x16_f16 on f16 hi/lo operand pairs, f32 accumulation; VALU/gather-bound",
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sc, rays_h, S, args.cpu_seconds)
            line["vs_cpu"] = value / line["cpu_baseline"]["value"]
        # sanity on the product's own output (not a parity check; tests/ do that)
        rgb = out["rgb_map"]
        assert bool(torch.isfinite(rgb).all()), "non-finite rgb"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measured_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (bench.py cannot run the
    profiler itself); only reported for the configuration the counters were collected on."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if (args.size == 512 and args.samples == 64 and args.fill == "full" and not args.early_term and not args.occ_cull and not args.split_f16
            and args.occupancy is None and os.path.exists(p)):
        return json.load(open(p))["hbm_bytes_per_launch"]
    return None


def cpu_baseline(sc, rays_h, S, target_s):
    """The CPU oracle (a C/OpenMP port of the reference path; the Python reference cannot travel)
    timed on this host over a bounded, evenly spaced sample of the same rays."""
    from oracle import oracle
    threads = oracle.max_threads()
    n = rays_h.shape[0]
    probe = rays_h[:: max(1, n // 256)][:256]
    t0 = time.perf_counter()
    oracle.render(sc, S, rays=probe, want_weights=False)
    per_ray = (time.perf_counter() - t0) / probe.shape[0]
    m = int(max(256, min(n, target_s / max(per_ray, 1e-9))))
    for _ in range(3):      # the first probe includes thread start-up; re-size until the sample costs about target_s
        sample = rays_h[:: max(1, n // m)][:m]
        t0 = time.perf_counter()
        oracle.render(sc, S, rays=sample, want_weights=False)
        dt = time.perf_counter() - t0
        if dt >= 0.6 * target_s or sample.shape[0] >= n:
            break
        m = int(min(n, m * target_s / max(dt, 1e-9)))
    return {"value": sample.shape[0] / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{sample.shape[0]} evenly spaced rays of the same frame x {S} samples, {dt:.1f} s on {threads} OpenMP threads "
                      f"(oracle/gpnerf_oracle.c)"}


if __name__ == "__main__":
    main()
