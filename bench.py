#!/usr/bin/env python3
"""Headline benchmark: rays/s of the fused per-ray render path on synthetic 512x512x64 frames.

    python bench.py --gpus N --steps K --warmup W
(N > 1: one rank per GPU over RCCL.  Under torch.distributed.run -- WORLD_SIZE / RANK / LOCAL_RANK in the environment -- this
process IS a rank; started plainly, it launches `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
before touching any GPU, relays rank 0's JSON line and exits with the child's status.)

A "step" renders ONE frame through gpnerf_render_fused (sample -> gather -> MLP -> composite) with every output
`Renderer.render` returns (rgb, depth, acc, disp, weights, z_vals, rgb_in), in the 32x8-pixel patch order `Renderer.render`
launches with.  Inputs are resident in HBM before the timed region.

  N = 1   value = rays of the frame / step time (BASELINE.json configs[1]; --early-term --samples 128 = configs[2]).
  N > 1   STRONG scaling (default): the same frame's rays are split over the N ranks in round-robin bands
          (gp-nerf_amd/parallel.py ShardPlan); a step = render my share + ONE all_gather_into_tensor of the packed pixels
          (rgb + depth, 16 B/ray) + re-assembly in ray order on every rank.  value = rays of the frame / step time, so
          value(N) / value(1) is the speed-up on one frame (north_star: >= 6x at 8 GPUs).  After the headline region the
          same flow is timed on a 1024x1024x64 frame (BASELINE.json configs[3]) and reported as `config4_1024`.
          --scaling weak: every rank renders its own full frame (a band of an N-times larger image) + the same all-gather.

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
FLOP_SIGMA_LAYER = 16384           # ... of which sigmahead.out_geometry_fc (128 -> 64)
FLOP_COLOUR_BRANCH = 52608 + 12288 + 7264      # ... and base_fc x 3 views + vis_fc x 3 + rgb_fc
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_F16_MFMA_TFLOPS = 2500.0      # dense f16 MFMA peak (the split-precision form's matrix instructions)
API_OUTPUTS = ("weights", "z_vals", "rgb_in")      # + rgb, depth, acc, disp (always written) = Renderer.render's dict
# the second fixed workload of the line: parameter and feature DISTRIBUTIONS of a trained checkpoint (tests/golden/trained_h1_s64's:
# head weights x 1 with biases, heavy-tailed x 4 features, ReLU-sparse volume levels, a density bias that leaves most of the box
# empty) at the headline's size -- the headline frame's random-init density net (weights_init) is zero on 71 % of its samples
TRAINED_LIKE = dict(H=512, W=512, seed=46, fill="full", pose="random", bias_std=0.1, sigma_bias=-10.0, head_scale=1.0, feat_scale=4.0,
                    feat_tail=0.5, vol_scale=4.0, vol_relu=True)


def price_work_done(stats, flops_algorithmic):
    """FLOPs of the layers a launch actually evaluated, from its step_stats (include/gpnerf_hip.h GpnerfOutputs.step_stats; one
    step = 32 samples): sample-loop steps x (everything but the colour branch) - volume levels left out x a quarter of the sigma
    feature layer + colour-branch evaluations x the colour branch.  Steps settled behind the sample loop (exactly opaque rays)
    ran no layer at all.  Returns (flops done, the exits as fractions of the launch's steps)."""
    steps = max(1, int(stats[0]))
    opaque, levels, passes = int(stats[3]), int(stats[4]), int(stats[5])
    loop = steps - opaque
    done = loop * (FLOP_PER_SAMPLE - FLOP_COLOUR_BRANCH) - levels * (FLOP_SIGMA_LAYER / 4.0) + passes * FLOP_COLOUR_BRANCH
    frac_done = done / (steps * FLOP_PER_SAMPLE)
    exits = {"steps_32_samples": steps,
             "opaque_tail_frac": opaque / steps,
             "sigma_layer_levels_left_out_frac": levels / (4.0 * steps),
             "sigma_layer_all_levels_empty_frac": max(0, int(stats[1]) - opaque) / steps,
             "colour_branch_not_run_frac": 1.0 - passes / steps,
             "flop_not_done_frac": 1.0 - frac_done}
    return flops_algorithmic * frac_done, exits


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="frame is size x size rays")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--fill", default="full", choices=["full", "survey"],
                    help="full: every pixel's ray crosses the SMPL bound (N = size^2); survey: f = 1.05 W (SURVEY.md §8d)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1 only")
    ap.add_argument("--early-term", action="store_true", help="configs[2]: wave-level early ray termination")
    ap.add_argument("--term-eps", type=float, default=1e-5)
    ap.add_argument("--sigma-bias", type=float, default=None,
                    help="shift of the last density bias of the random-init net (default 0; 1.0 with --early-term so that rays "
                         "become opaque, which a trained net's do and a random one's do not)")
    ap.add_argument("--outputs", default="api", choices=["api", "light"],
                    help="api: every map Renderer.render returns; light: rgb/depth/acc/disp only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the measurements reported beside the headline")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="target CPU time of the cpu_baseline samples (half per CPU program)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--split-f16", action="store_true",
                    help="dense layers on f16 MFMA with fp32 operands split into hi+lo (GPNERF_FLAG_SPLIT_F16)")
    ap.add_argument("--fold", action="store_true",
                    help="the fast fp32 form of round 4 (coarse levels folded per frame, log2e-scaled layers; `render.file hip_render_fold`) "
                         "instead of the default reference-order form")
    ap.add_argument("--no-fold", action="store_true", help="(default since round 5; accepted for old command lines)")
    ap.add_argument("--no-guard", action="store_true", help="with --split-f16: without the range guard (GPNERF_FLAG_SPLIT_GUARD)")
    ap.add_argument("--occ-cull", action="store_true",
                    help="progressive sample culling (demo_render.py semantics) on a sparse synthetic pyramid")
    ap.add_argument("--occupancy", type=float, default=None, help="fraction of coarse volume blocks that are occupied")
    ap.add_argument("--ray-order", default="patch", choices=["patch", "raster"],
                    help="patch: 32x8-pixel workgroup tiles (what Renderer.render passes as ray_order); raster: the list as given")
    ap.add_argument("--patch", default=None,
                    help="WxH of the patches of --ray-order patch: one patch row of W pixels per wavefront when W >= 32, a whole WxH = 32 "
                         "pixel block per wavefront otherwise.  Default 32x8 (Renderer.render's dense order), 8x4 with --early-term and "
                         "4x8 with --occ-cull: the rays of a compact block terminate / are culled together far more often than those "
                         "of a 32-pixel row")
    args = ap.parse_args()
    if args.patch is None:
        args.patch = "8x4" if args.early_term else ("4x8" if args.occ_cull else "32x8")
    return args


class Workload:
    """One synthetic frame resident on `dev`: frame constants, ray list, patch order."""

    def __init__(self, args, size, samples, dev, sigma_bias):
        import torch
        fm = importlib.import_module("gp-nerf_amd.frame")
        syn = importlib.import_module("gp-nerf_amd.synthetic")
        self.H = self.W = size
        self.S = samples
        self.sc = sc = syn.make_scene(H=size, W=size, seed=args.seed, fill=args.fill, pose="identity", vol_occupancy=args.occupancy,
                                      sigma_bias=sigma_bias)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.vols_dev = [t(v) for v in sc["volumes"]]                       # the 4 dense levels as the reference lays them out (NCDHW)
        self.frame = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), self.vols_dev, t(sc["src_Ks"][0]),
                              t(sc["src_poses"][0]), sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0],
                              fm.pack_head(sc["head"], dev))
        self.rays_h = np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32)
        self.rays = t(self.rays_h)
        pw, ph = (int(v) for v in args.patch.split("x"))
        self.patch = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], size, size, patch_w=pw, patch_h=ph)).to(dev)
        self.n = self.rays.shape[0]


def time_launches(fn, steps, warmup):
    """Mean HIP-event duration of fn() (events on torch's current stream, the one the kernel is launched on)."""
    import torch
    for _ in range(warmup):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        out = fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev])), out


def device_identity(dev):
    """What tells two ranks' devices apart in the JSON line: index, marketing name, PCI bus id, uuid (None where the runtime
    does not say).  A CPU "device" (the gloo dry run's host tensors) reports its host name and pid instead."""
    import torch
    if dev.type != "cuda":
        import socket
        return {"device_index": None, "device_name": "cpu", "pci_bus_id": None, "uuid": None, "host": socket.gethostname(), "pid": os.getpid()}
    p = torch.cuda.get_device_properties(dev)
    bus = None
    if all(hasattr(p, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bus = f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    uuid = getattr(p, "uuid", None)
    return {"device_index": int(dev.index if dev.index is not None else torch.cuda.current_device()), "device_name": str(p.name),
            "pci_bus_id": bus, "uuid": str(uuid) if uuid is not None else None, "cus": int(getattr(p, "multi_processor_count", 0)), "pid": os.getpid()}


def gather_rank_reports(report, world):
    """Every rank's report on rank 0 (and everywhere): ONE all_gather_object after the timed region.  `report` = device identity +
    this rank's own kernel / exchange event times, so a SCALE run shows N distinct devices and where each rank's time went."""
    if world == 1:
        return [report]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, report)
    return out


def summarize_ranks(reports):
    """`ranks` (the reports in rank order) + the fields a reader wants first: how many distinct devices, per-rank kernel time
    spread, and the exchange's own time."""
    def ident(r):
        if r.get("device_index") is None:
            return ("cpu", r.get("host"), r.get("pid"))
        return r.get("pci_bus_id") or r.get("uuid") or ("index", r.get("device_index"))
    k = [r["kernel_ms"] for r in reports if r.get("kernel_ms") is not None]
    e = [r["exchange_ms"] for r in reports if r.get("exchange_ms") is not None]
    return {"ranks": reports, "distinct_devices": len({ident(r) for r in reports}),
            "kernel_ms_per_rank": {"min": min(k), "max": max(k), "mean": sum(k) / len(k)} if k else None,
            "exchange_ms": {"min": min(e), "max": max(e), "mean": sum(e) / len(e)} if e else None}


def launch_command(gpus, port, argv):
    """The command a plain `python bench.py --gpus N ...` runs as a child: the driver's own N > 1 launch line."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(args):
    """`--gpus N` with no WORLD_SIZE in the environment: start the N ranks as a CHILD process (this one has not touched a GPU and
    never does -- it only counts devices, which does not initialise HIP on this image --, and it does not exec), pass rank 0's
    JSON line through on stdout and return the child's exit status."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if env.get("GPNERF_BENCH_BACKEND", "nccl") == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but {have} device(s) visible; a measured run needs one GPU per rank "
                  f"(GPNERF_BENCH_BACKEND=gloo is the dry run with ranks sharing devices)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.Popen(launch_command(args.gpus, port, sys.argv[1:]), stdout=subprocess.PIPE, text=True, env=env)
    for line in p.stdout:                             # the ranks' stderr goes straight through; stdout carries rank 0's one line
        if line.lstrip().startswith("{"):
            print(line.rstrip("\n"), flush=True)
        else:
            sys.stderr.write(line)
    return p.wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
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
    par = importlib.import_module("gp-nerf_amd.parallel")
    sigma_bias = args.sigma_bias if args.sigma_bias is not None else (1.0 if args.early_term else 0.0)
    wl = Workload(args, args.size, args.samples, dev, sigma_bias)
    S = wl.S
    want = API_OUTPUTS if args.outputs == "api" else ()
    strong = world > 1 and args.scaling == "strong"
    kw = dict(early_term=args.early_term, term_eps=args.term_eps, occ_cull=args.occ_cull, split_f16=args.split_f16,
              guard=False if args.no_guard else None)

    class Flow:
        """The timed step for one workload: single-GPU frame, strong-scaling share + gather, or weak-scaling band + gather."""

        def __init__(self, wl):
            self.wl = wl
            if strong:
                # the frame's ray list in patch-major order, split in round-robin bands; outputs come back in that list's order
                self.rays_all = wl.rays.index_select(0, wl.patch.long()) if args.ray_order == "patch" else wl.rays
                self.plan = par.plan_for(wl.n, world, dev)
                self.rays = self.plan.take(self.rays_all, rank).contiguous()
                self.order = None
                self.want = ()                             # the exchange carries rgb + depth: 16 B/ray
                self.buf = torch.empty((world * self.plan.share, 4), device=dev if backend == "nccl" else "cpu")
                self.rays_per_step = wl.n
            else:
                self.rays, self.order, self.want = wl.rays, (wl.patch if args.ray_order == "patch" else None), want
                self.buf = torch.empty((world, wl.n, 4), device=dev if backend == "nccl" else "cpu") if world > 1 else None
                self.rays_per_step = wl.n * world
            self.n_local = self.rays.shape[0]
            # the fold is per-frame work and the bench re-uses one Frame: True = fold again in every step, inside the timed region
            self.fold = bool(args.fold and not args.split_f16 and not args.occ_cull)

        def render(self):
            return fm.render_fused(self.wl.frame, self.rays, self.wl.S, want=self.want, ray_order=self.order, fold=self.fold, **kw)

        def step(self, events=None):
            if events:
                events[0].record()
            out = self.render()
            if events:
                events[1].record()
            if strong:
                local = out if backend == "nccl" else {k: out[k].cpu() for k in par.PIXEL_KEYS}
                out = par.gather_frame(local, self.plan if backend == "nccl" else _cpu_plan(par, self.plan), par.PIXEL_KEYS, buffer=self.buf)
            elif world > 1:
                par.all_gather_pixels(out if backend == "nccl" else {k: out[k].cpu() for k in par.PIXEL_KEYS}, self.buf)
            if events and len(events) > 2:
                events[2].record()          # the exchange (pack + all-gather + un-permute) ends here on the device stream (nccl backend)
            return out

        def timed(self, steps, warmup):
            for _ in range(warmup):
                self.step()
            ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3 if world > 1 else 2)) for _ in range(steps)]
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                out = self.step(ev[i])
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            # the exchange's own device time (RCCL all-gather + un-permute on the stream); the gloo dry run stages through the host
            self.exchange_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if (world > 1 and backend == "nccl") else None
            return dt, float(np.mean([e[0].elapsed_time(e[1]) for e in ev])), out

    flow = Flow(wl)
    # how many 32-sample steps take the kernel's bit-exact exits: counted by one launch BEFORE the warm-up (it is the process's cold
    # launch; behind the timed region it would be a second cold one -- the device clocks down while the host reduces the timings --
    # and sit in the profiler's average of this kernel)
    step_stats = fm.render_fused(wl.frame, flow.rays, S, want=flow.want + ("step_stats",), ray_order=flow.order, fold=flow.fold, **kw)["step_stats"].cpu().numpy()
    dt, kernel_ms, out = flow.timed(args.steps, args.warmup)
    # who ran where: every rank's device identity and its own event times, gathered once, after the timed region
    report = dict(device_identity(dev), rank=rank, local_rank=local_rank, kernel_ms=kernel_ms, exchange_ms=getattr(flow, "exchange_ms", None),
                  rays=int(flow.n_local))
    rank_summary = summarize_ranks(gather_rank_reports(report, world))

    # the data-independent figure: the same launch with every layer evaluated for every sample (GPNERF_FLAG_NO_EXITS; same bits),
    # measured right behind the timed region, never part of `value`
    dense = None
    if world == 1 and rank == 0 and not args.split_f16 and not args.early_term and not args.occ_cull:
        d_ms, d_out = time_launches(lambda: fm.render_fused(wl.frame, flow.rays, S, want=flow.want, ray_order=flow.order, fold=flow.fold, exits=False, **kw),
                                    args.steps, args.warmup)
        d_tf = float(flow.n_local) * S * FLOP_PER_SAMPLE / (d_ms * 1e-3) / 1e12
        dense = {"dense_ms": d_ms, "dense_frac": d_tf / PEAK_F32_MFMA_TFLOPS, "dense_rays_per_sec": flow.n_local / (d_ms * 1e-3),
                 "dense_same_bits": bool(all(torch.equal(torch.nan_to_num(d_out[k]), torch.nan_to_num(out[k])) for k in ("rgb_map", "depth_map", "acc_map"))),
                 "dense_note": "GPNERF_FLAG_NO_EXITS: every layer of every sample evaluated (all 110 848 FLOP x samples on the matrix pipe), same maps bit for bit"}
    extras = {}
    if world > 1 and strong and not args.no_extras and args.size != 1024:
        # BASELINE.json configs[3]: one 1024x1024x64 frame over the same ranks, same flow
        wl4 = Workload(args, 1024, 64, dev, 0.0)
        f4 = Flow(wl4)
        dt4, k4, out4 = f4.timed(max(3, args.steps // 2), 2)
        st4 = max(3, args.steps // 2)
        extras["config4_1024"] = {"workload": "1024x1024 frame, 64 samples/ray, rays in round-robin bands over the ranks + pixel all-gather "
                                              "(BASELINE.json configs[3])", "value": wl4.n * st4 / dt4, "unit": "rays/s", "ms_per_frame": dt4 / st4 * 1e3,
                                  "kernel_ms_rank0": k4, "rays_per_rank": int(f4.n_local), "rays_total": int(wl4.n),
                                  "maps_finite": bool(all(torch.isfinite(out4[k]).all() for k in par.PIXEL_KEYS))}
        del wl4, f4

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = flow.rays_per_step * args.steps / dt
        flops_per_launch = float(flow.n_local) * S * FLOP_PER_SAMPLE
        evaluated = None
        alive_after = None
        if args.early_term or args.occ_cull:
            # the units a launch processes are the samples it evaluates: terminated / culled samples are not work done
            done = fm.render_fused(wl.frame, flow.rays, S, want=("samples_done",), ray_order=flow.order, fold=flow.fold, **kw)["samples_done"]
            evaluated = float(done.float().mean()) / S
            alive_after = {str(k): float((done > k).float().mean()) for k in range(32, S, 32)}
            flops_per_launch *= evaluated
        # The fp32 forms leave out work that provably cannot change any output, wave step by wave step (bit-exact, see
        # include/gpnerf_hip.h step_stats).  The roofline prices the work DONE: `achieved` = FLOPs of the layers the launch
        # evaluated / its duration, so `frac` never exceeds 1 and is what the matrix pipe was asked to do; the rate of ANSWERS
        # (every sample the launch is answerable for x 110 848 FLOP, which can exceed the pipe's peak when most colour branches are
        # provably not needed) is roofline.algorithmic_rate; the same launch with every layer evaluated for every sample
        # (GPNERF_FLAG_NO_EXITS, data-independent) is roofline.dense_ms / dense_frac.
        flops_done, exits = price_work_done(step_stats, flops_per_launch)
        exits["note"] = ("bit-exact exits (GPNERF_FLAG_NO_EXITS switches them off: roofline.dense_*): a volume level whose features are zero in all 32 "
                         "samples of a step is left out of the sigma feature layer; a sample whose weight alpha*T is exactly zero (nn.ReLU on the density; "
                         "masked_fill) adds fma(0, rgb, c) = c to the colour map, so its colour branch is not evaluated: the sample loop lists the samples "
                         "that need it, colour_units_kernel runs the branch on 32 list entries at a time and colour_accumulate_kernel adds every ray's terms "
                         "in sample order (kernel_ms is the whole call: the three kernels); steps behind the sample at which all 32 rays' transmittance is exactly "
                         "0 are settled without a gather or an MFMA")
        algorithmic = flops_per_launch / (kernel_ms * 1e-3) / 1e12
        achieved = flops_done / (kernel_ms * 1e-3) / 1e12
        cfg_no = 2 if args.early_term else (3 if (args.size == 1024 and world > 1) else 1)
        if world == 1:
            parallelism = "single GPU"
        elif strong:
            parallelism = (f"strong: one frame's rays in round-robin bands of {par.INTERLEAVE_BAND} over {world} GPUs, one all_gather_into_tensor "
                           f"of rgb+depth (16 B/ray) per frame")
        else:
            parallelism = f"weak: every one of {world} GPUs renders its own frame-sized band, all-gather of rgb+depth"
        traffic, traffic_src = measured_traffic(args, world)
        peak = PEAK_F16_MFMA_TFLOPS if args.split_f16 else PEAK_F32_MFMA_TFLOPS
        line = {
            "metric": "rays_per_sec", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_frame": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl.H}x{wl.W} frame, {S} samples/ray, fused HIP render kernel, synthetic SMPL bound + random "
                                   f"feature volume (BASELINE.json configs[{cfg_no}])",
                       "rays_per_gpu": int(flow.n_local), "rays_total": int(flow.rays_per_step), "samples_per_ray": S, "fill": args.fill,
                       "ray_order": args.ray_order, "patch": args.patch if args.ray_order == "patch" else None, "outputs": "rgb+depth (the all-gather payload)" if strong else
                       ("rgb,depth,acc,disp,weights,z_vals,rgb_in (Renderer.render's dict)" if args.outputs == "api" else "rgb,depth,acc,disp"),
                       "early_term": bool(args.early_term), "term_eps": args.term_eps if args.early_term else None, "sigma_bias": sigma_bias,
                       "form": "split-f16" if args.split_f16 else ("fp32, folded coarse levels (round 4)" if flow.fold else "fp32, reference summation order"),
                       "exits": "none (the split-precision forms evaluate every layer of every sample)" if args.split_f16 else
                                "bit-exact (roofline.exits): zero-weight samples' colour branch, all-zero volume levels, samples behind an exactly zero "
                                "transmittance are not evaluated; the same launch with everything evaluated is roofline.dense_ms / dense_frac",
                       "folded_volumes": bool(flow.fold), "occ_cull": bool(args.occ_cull), "split_f16": bool(args.split_f16), "split_guard": bool(args.split_f16 and not args.no_guard), "vol_occupancy": args.occupancy,
                       "out_sh_dhw": [int(x) for x in wl.sc["out_sh"][0]], "parallelism": parallelism},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "render_fused_kernel" if args.split_f16 else
                         "render_fused_kernel + colour_units_kernel + colour_accumulate_kernel (one gpnerf_render_fused call; kernel_ms = HIP events around the call = the sum of their rocprofv3 averages)",
                         "kernel_ms": kernel_ms, "flop_per_launch": flops_done,
                         "accounting": "work done: FLOPs of the dense layers the launch evaluated (step_stats) / kernel_ms; never above the peak",
                         "algorithmic_rate": {"tflops": algorithmic, "flop_per_launch": flops_per_launch, "over_peak": algorithmic / peak,
                                              "note": "every sample the launch answers for x 110 848 FLOP per second: a rate of answers, not of MFMAs"},
                         "exits": exits},
        }
        if dense is not None:
            line["roofline"].update(dense)
        line.update(extras)
        line.update(rank_summary)
        if args.split_f16:
            line["dtype"] = "f32 operands as f16 hi+lo pairs on v_mfma_f32_32x32x16_f16, f32 accumulation"
            line["roofline"]["split_note"] = ("peak = dense f16 MFMA; the form issues 3 MFMAs per 16-deep k-step (3x the algorithmic FLOPs) and is "
                                              "bound by vector-ALU issue, not by the matrix pipe (30 % busy: profiles/r02/b_pmc_summary_split_guarded.json)")
        if evaluated is not None:
            line["roofline"]["samples_evaluated_frac"] = evaluated
            line["roofline"]["note"] = ("flop_per_launch counts the samples the launch evaluated (a ray stops once its T < term_eps / a wavefront skips "
                                        "steps whose 32 samples are all unoccupied), not the S per ray the reference would")
        if args.early_term:
            line["early_term"] = {"samples_evaluated_frac": evaluated, "rays_alive_after_samples": alive_after,
                                  "note": "a ray stops once its T < term_eps; the launch walks the samples in 16-sample segments and re-packs the rays still alive "
                                          "32 to a wavefront for every segment (frames smaller than one round of wavefronts: a 32-ray tile stops as a whole)"}
        if world == 1 and not args.no_extras:
            try:        # the second fixed workload: trained-like distributions at the headline's size
                line["trained_like"] = trained_like_workload(args, fm)
            except Exception as e:
                line["trained_like"] = {"error": repr(e)[:300]}
            line["beside_headline"] = beside_headline(args, fm, wl, kw, flow)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl.sc, wl.rays_h, S, args.cpu_seconds)
            try:        # BASELINE.json configs[0] (the reference's own CPU-runnable case): the whole 64x64x32 crop on the same cores
                line["cpu_baseline"]["config1_64x64x32"] = cpu_config1(args.seed)
            except Exception as e:
                line["cpu_baseline"]["config1_64x64x32"] = {"error": repr(e)[:200]}
            line["vs_cpu"] = value / line["cpu_baseline"]["value"]                           # against the blocked (fast) CPU twin
            line["vs_cpu_scalar_oracle"] = value / line["cpu_baseline"]["scalar_oracle"]["value"]
        # sanity on the product's own output (not a parity check; tests/ do that)
        assert bool(torch.isfinite(out["rgb_map"]).all()), "non-finite rgb"
        line["maps_finite"] = bool(torch.isfinite(out["rgb_map"]).all() and torch.isfinite(out["depth_map"]).all())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def trained_like_workload(args, fm):
    """The headline launch on a frame with a trained checkpoint's DISTRIBUTIONS (TRAINED_LIKE; tools/probes/skip_probe.py's
    "trained-like x 1"): kernel time with the bit-exact exits (the default), priced on the work done, and with every layer
    evaluated (dense) -- beside the headline, whose random-init density net is zero on 71 % of its samples."""
    import torch
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    dev = torch.device("cuda", torch.cuda.current_device())
    S = 64
    sc = syn.make_scene(**TRAINED_LIKE)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fr = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                  sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
    rays = t(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))
    order = torch.from_numpy(fm.patch_order(sc["mask_at_box"][0], TRAINED_LIKE["H"], TRAINED_LIKE["W"], patch_w=32, patch_h=8)).to(dev)
    n = int(rays.shape[0])
    stats = fm.render_fused(fr, rays, S, want=API_OUTPUTS + ("step_stats",), ray_order=order)["step_stats"].cpu().numpy()
    ms, o = time_launches(lambda: fm.render_fused(fr, rays, S, want=API_OUTPUTS, ray_order=order), args.steps, args.warmup)
    d_ms, d = time_launches(lambda: fm.render_fused(fr, rays, S, want=API_OUTPUTS, ray_order=order, exits=False), args.steps, args.warmup)
    alg = float(n) * S * FLOP_PER_SAMPLE
    done, exits = price_work_done(stats, alg)
    w = o["weights"]
    return {"workload": "512x512 frame, 64 samples/ray, trained-like distributions (heads x 1 with biases, heavy-tailed x 4 features, ReLU-sparse levels, "
                        "density bias -10: tests/golden/trained_h1_s64's), every pixel's ray through the box",
            "scene": {k: v for k, v in TRAINED_LIKE.items()}, "rays": n, "kernel_ms": ms, "rays_per_sec": n / (ms * 1e-3),
            "frac": done / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "algorithmic_over_peak": alg / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "dense_ms": d_ms, "dense_frac": alg / (d_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "dense_same_bits": bool(all(torch.equal(torch.nan_to_num(d[k]), torch.nan_to_num(o[k])) for k in ("rgb_map", "depth_map", "acc_map", "weights"))),
            "zero_weight_samples_frac": float((w == 0).float().mean()), "exits": exits}


def _cpu_plan(par, plan):
    """gloo dry run: the same plan with host index tensors."""
    import torch
    return par.plan_for(plan.n_rays, plan.world, torch.device("cpu"), plan.band)


def beside_headline(args, fm, wl, kw, flow):
    """Measured AFTER the timed region, never part of `value`: the same launch with the light output set, in raster order,
    in the split-precision form, and the wall time of the whole Renderer.render(batch) call on the same frame."""
    import torch
    S, st, wu = wl.S, args.steps, args.warmup
    res = {}
    head_ms = None
    for name, want, order, extra in (("api_outputs_patch_order", API_OUTPUTS, wl.patch, {}), ("light_outputs_patch_order", (), wl.patch, {}),
                                     ("api_outputs_raster_order", API_OUTPUTS, None, {}),
                                     ("fp32_folded_api_outputs_patch_order", API_OUTPUTS, wl.patch, {"fold": True}),
                                     ("split_f16_api_outputs_patch_order", API_OUTPUTS, wl.patch, {"split_f16": True}),
                                     ("split_f16_unguarded_api_outputs_patch_order", API_OUTPUTS, wl.patch, {"split_f16": True, "guard": False})):
        k2 = dict(kw, fold=flow.fold and not extra.get("split_f16", False))
        k2.update(extra)
        ms, o = time_launches(lambda: fm.render_fused(wl.frame, wl.rays, S, want=want, ray_order=order, **k2), st, wu)
        res[name] = {"kernel_ms": ms, "rays_per_sec": wl.n / (ms * 1e-3)}
        if name == "api_outputs_patch_order":
            head_ms, head_out = ms, o
        if extra:
            res[name]["max_abs_vs_f32_form"] = {"rgb": float((o["rgb_map"] - head_out["rgb_map"]).abs().max()),
                                                "depth": float((o["depth_map"] - head_out["depth_map"]).abs().max())}
            res[name]["note"] = ("round 4's fast fp32 form (`render.file hip_render_fold`): coarse levels folded into the sigma feature layer per frame "
                                 "(the fold is inside the timed step), log2e-scaled layers; not in the reference's summation order") if extra.get("fold") else \
                                ("dense layers as 3 x v_mfma_f32_32x32x16_f16 on f16 hi/lo operand pairs, f32 accumulation; " +
                                 ("no range check (operands must stay below 65504)" if extra.get("guard") is False else
                                  "range guard on: tiles with an operand at the f16 range are rendered again in the fp32 form"))
    try:
        res["renderer_api"] = renderer_api_wall(args, wl)
    except Exception as e:                       # the API timing must never take the headline down with it
        res["renderer_api"] = {"error": repr(e)[:200]}
    try:                                         # the ZJU-sized frame: SURVEY.md 8d's literal f = 1.05 W camera, ~74 k rays
        res["renderer_api_survey_frame"] = renderer_api_wall(args, None, survey=True)
    except Exception as e:
        res["renderer_api_survey_frame"] = {"error": repr(e)[:200]}
    try:                                         # the README command's renderer on a person-shaped frame
        res["demo_render_body_frame"] = demo_render_body_frame(args)
    except Exception as e:
        res["demo_render_body_frame"] = {"error": repr(e)[:300]}
    try:                                         # the evaluation loop over such frames, serial (the reference's) against pipelined
        res["eval_loop"] = eval_loop_wall(args)
    except Exception as e:
        res["eval_loop"] = {"error": repr(e)[:300]}
    try:                                         # the same loop behind the split-precision encoder (`encoder.file hip_encoder_fast`)
        res["eval_loop_fast_encoder"] = eval_loop_wall(args, encoder_file="hip_encoder_fast")
    except Exception as e:
        res["eval_loop_fast_encoder"] = {"error": repr(e)[:300]}
    return res


def renderer_api_wall(args, wl, survey=False):
    """Wall time of `build_render(cfg).render(batch)` (the reference's entry point, tools/inference.py:61 + BaseTrainer.py:267) on
    the bench frame: with feature maps and volumes handed in (frame build + render + dict), and with the per-frame producers
    (image encoder, vertex attention, sparse volume builder) running too.  survey=True: the same call, producers running, on the
    frame a real evaluation loop renders (512x512 sources, the full-size SMPL box seen through SURVEY.md 8d's f = 1.05 W camera:
    ~74 k rays x 64 samples), where the producers are a third of the call instead of a sixth."""
    import torch
    from types import SimpleNamespace as NS
    if survey:
        syn = importlib.import_module("gp-nerf_amd.synthetic")
        dev0 = torch.device("cuda", torch.cuda.current_device())
        sc = syn.make_scene(H=512, W=512, seed=args.seed, fill="survey", pose="identity", make_volumes=False)
        wl = NS(S=64, sc=sc, rays=torch.empty((0, 8), device=dev0), vols_dev=None)
    p = os.path.join(ROOT, "gp-nerf_amd", "plugins")
    if p not in sys.path:
        sys.path.insert(0, p)
    hip_render = importlib.import_module("hip_render")
    cfg = NS(encoder=NS(file="hip_encoder", name="resnet34", out_ch=32),
             head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32, 32, 32, 32])),
             dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3),
             train=NS(n_rays=1024, n_samples=wl.S), test=NS(mesh_th=50))
    dev = wl.rays.device
    r = hip_render.build_render(cfg).to(dev).eval()
    sd = r.state_dict()
    for k, v in wl.sc["head"].items():
        sd["nerfhead." + k] = torch.from_numpy(v.copy())
    r.load_state_dict(sd, strict=True)
    sc = wl.sc
    keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk",
            "mask_at_box")
    b = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in keys}
    res = {}
    with torch.no_grad():
        legs = (("with_producers", {}),) if survey else (("products_in_batch", {"featmaps": torch.from_numpy(sc["featmaps"]).to(dev), "volumes": wl.vols_dev}),
                                                          ("with_producers", {}))
        for name, extra in legs:
            bb = dict(b, **extra)
            for _ in range(2):
                r.render(bb)
            ts, et, rt = [], [], []
            for _ in range(9):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ret = r.render(bb)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
                et.append(ret["etime"] * 1e3)
                rt.append(ret["rtime"] * 1e3)
            res[name] = {"wall_ms": float(np.median(ts)), "wall_ms_min_max": [float(np.min(ts)), float(np.max(ts))],
                         "etime_ms": float(np.median(et)), "rtime_ms": float(np.median(rt)), "rtime_ms_min_max": [float(np.min(rt)), float(np.max(rt))],
                         "calls": len(ts), "rays": int(ret["rgb_map"].shape[1]), "returns": sorted(k for k in ret if k not in ("etime", "rtime"))}
    res["note"] = ("products_in_batch: batch carries featmaps + the 4 dense levels; with_producers: hip_encoder + vertex attention + sparse "
                   "volume builder run per frame (their volumes are sparse, the per-ray kernel's work is the same)")
    return res


def demo_render_body_frame(args):
    """The README command's renderer (`render.file hip_demo_render` = libs/renders/demo_render.py: pixels selected from the occupied
    voxels, samples occupancy-culled) on a PERSON-SHAPED frame: synthetic.body_vertices' capsule-limbed figure in the full-size SMPL
    box, the dense levels non-negative on the voxels the sparse pyramid writes for those vertices, SURVEY.md 8d's f = 1.05 W camera at
    512x512, 64 samples (the scene of tests/golden/demo_body_s64.npz with pose = identity).  Wall time of render(batch) with the
    products in the batch, the per-ray kernel alone, the fraction of the samples it evaluates, and the MFMA roofline on those."""
    import torch
    from types import SimpleNamespace as NS
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    fm = importlib.import_module("gp-nerf_amd.frame")
    dev = torch.device("cuda", torch.cuda.current_device())
    S = 64
    sc = syn.make_scene(H=512, W=512, seed=args.seed, focal_mul=1.05, body="capsules", sigma_bias=0.5, bias_std=0.1, pose="identity")
    p = os.path.join(ROOT, "gp-nerf_amd", "plugins")
    if p not in sys.path:
        sys.path.insert(0, p)
    demo = importlib.import_module("hip_demo_render")
    cfg = NS(encoder=NS(file="hip_encoder", name="resnet34", out_ch=32),
             head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32, 32, 32, 32])),
             dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3),
             train=NS(n_rays=1024, n_samples=S), test=NS(mesh_th=50))
    torch.manual_seed(args.seed)
    r = demo.build_render(cfg).to(dev).eval()
    sd = r.state_dict()
    for k, v in sc["head"].items():
        sd["nerfhead." + k] = torch.from_numpy(v.copy())
    r.load_state_dict(sd, strict=True)
    keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk",
            "mask_at_box", "target_K", "target_pose", "target_K_inv")
    b = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in keys}
    b["featmaps"] = torch.from_numpy(sc["featmaps"]).to(dev)
    b["volumes"] = [torch.from_numpy(v).to(dev) for v in sc["volumes"]]
    with torch.no_grad():
        for _ in range(2):
            ret = r.render(b)
        ts, rt, kr = [], [], []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = r.render(b)
            ts.append((time.perf_counter() - t0) * 1e3)
            rt.append(ret["rtime"] * 1e3)
            kr.append(ret["time_slots"]["render"] * 1e3)
    n_sel = int(ret["mask_at_box"].sum())
    # the per-ray launch alone + how many samples it evaluates (samples_done counts the steps a ray's wavefront ran for it)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    frame = fm.Frame(t(sc["src_imgs"][0]), t(sc["featmaps"]), [t(v) for v in sc["volumes"]], t(sc["src_Ks"][0]), t(sc["src_poses"][0]),
                     sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0], fm.pack_head(sc["head"], dev))
    frame.build_occupancy()
    rays, mask = fm.select_rays(frame, b["target_K"][0], b["target_pose"][0], 512, 512, sc["voxel_size"], b["bounds"][0, 0], b["Rh"][0], b["Th"][0],
                                neg_ray=False, target_K_inv=b["target_K_inv"][0], compact=False)
    idx = torch.nonzero(mask).squeeze(1)
    order = fm.patch_order_of(idx, 512)
    k_ms, _ = time_launches(lambda: fm.render_fused(frame, rays, S, occ_cull=True, want=(), ray_order=order, subset=True), 10, 3)
    occ = frame.occ
    pts, _, grid = fm.sample_points(frame, rays.index_select(0, idx.long()), S)
    # evaluated = occupancy interpolates to > 0 at the sample (demo_render.py:270-283), with the renderer's literal 0.005 grid
    import torch.nn.functional as F
    g = grid.view(1, -1, 1, 1, 3)
    ev = float((F.grid_sample(occ.view(1, 1, *occ.shape), g, padding_mode="zeros", align_corners=True).view(-1) > 0).float().mean())
    flops = n_sel * S * ev * FLOP_PER_SAMPLE
    return {"workload": "512x512 target, person-shaped vertices in the full-size SMPL box, f = 1.05 W, 64 samples/ray, progressive renderer",
            "pixels_selected": n_sel, "pyramid_sites_per_level": [int((v.abs().sum(1) > 0).sum()) for v in b["volumes"]],
            "wall_ms": float(np.median(ts)), "rtime_ms": float(np.median(rt)), "per_ray_kernel_ms_in_call": float(np.median(kr)),
            "per_ray_kernel_ms": k_ms, "samples_evaluated_frac": ev,
            "roofline_on_evaluated_samples": {"achieved_tflops": flops / (k_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                                              "frac": flops / (k_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "note": "products (feature maps, dense levels) in the batch; wall = encoder-less render(batch) incl. ray selection, occupancy, "
                    "device-to-host copy of the image"}


def eval_loop_wall(args, frames=12, fill="survey", reserve=(0,), encoder_file="hip_encoder"):
    """The evaluation LOOP (libs/trainers/BaseTrainer.py:255-280 = evaluator.evaluate_loop) over `frames` ZJU-sized frames (SURVEY.md
    8d's f = 1.05 W camera, ~74 k rays x 64 samples, hip_encoder + vertex attention + sparse volume builder per frame, PSNR / MSE /
    SSIM per frame): wall time per frame of the reference's strictly serial loop against the pipelined one (Renderer.prefetch of
    frame t + 1 behind frame t's per-ray kernel).  Same bits per frame (tests/test_gpu_renderer.py).  encoder_file: hip_encoder (the
    default: fp32 operands, the chain inside 1e-4 of the reference) or hip_encoder_fast (f16 hi/lo operands, ~0.9 ms less per frame)."""
    import torch
    from types import SimpleNamespace as NS
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    ev = importlib.import_module("gp-nerf_amd.evaluator")
    dev = torch.device("cuda", torch.cuda.current_device())
    sc = syn.make_scene(H=512, W=512, seed=args.seed, fill=fill, pose="identity", make_volumes=False)
    p = os.path.join(ROOT, "gp-nerf_amd", "plugins")
    if p not in sys.path:
        sys.path.insert(0, p)
    hip_render = importlib.import_module("hip_render")
    cfg = NS(encoder=NS(file=encoder_file, name="resnet34", out_ch=32),
             head=NS(file="hip_head", rgb=NS(use_rgbhead=True), sigma=NS(code_dim=32, n_heads=4, n_layers=4, n_smpl=6890, outdims=[32, 32, 32, 32])),
             dataset=NS(train=NS(name="zju_mocap", chunk=400), test=NS(name="zju_mocap", chunk=2000), voxel_size=[0.005] * 3, H=512, W=512, ratio=1.0),
             train=NS(n_rays=1024, n_samples=64), test=NS(mesh_th=50, test_seq="bench", save_imgs=False))
    torch.manual_seed(args.seed)                 # the encoder's initialisation (the head's comes from the scene)
    r = hip_render.build_render(cfg).to(dev).eval()
    sd = r.state_dict()
    for k, v in sc["head"].items():
        sd["nerfhead." + k] = torch.from_numpy(v.copy())
    r.load_state_dict(sd, strict=True)
    keys = ("ray_o", "ray_d", "near", "far", "src_imgs", "src_Ks", "src_poses", "feature", "coord", "out_sh", "bounds", "Rh", "R", "Th", "body_msk",
            "mask_at_box")
    b = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in keys}
    n = int(b["ray_o"].shape[1])
    b["rgb"] = torch.rand((1, n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    loader = [dict(b) for _ in range(frames)]
    res = {"frames": frames, "rays_per_frame": n, "encoder": encoder_file + " (" + r.encoder.precision + ")"}
    modes = [("serial", False, 0), ("pipelined", True, 0)] + [(f"pipelined_reserve_{n}_cus", True, n) for n in reserve if n]
    for name, mode, res_cus in modes:
        r.reserve_cus = res_cus
        ev.evaluate_loop(r, loader[:3], cfg, pipeline=mode, quiet=True)               # warm-up: graph capture, allocator
        walls, rts = [], []
        for _ in range(3):
            torch.cuda.synchronize()
            out = ev.evaluate_loop(r, loader, cfg, pipeline=mode, quiet=True)
            walls.append(out["wall_time"] / frames * 1e3)
            rts.append(out["avg_time"] * 1e3)
        res[name] = {"wall_ms_per_frame": float(np.median(walls)), "wall_ms_per_frame_min_max": [float(np.min(walls)), float(np.max(walls))],
                     "avg_rtime_ms": float(np.median(rts)), "psnr_first_frame": float(out["psnr"][0])}
    res["note"] = ("wall = the loop's own clock / frames, evaluator (PSNR, MSE, SSIM on the device) included; pipelined: frame t+1's encoder graph, "
                   "builder and frame glue are enqueued on a second stream behind frame t's per-ray kernel, the host evaluates frame t meanwhile")
    return res


def measured_traffic(args, world):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes of the same configuration
    (bench.py cannot run the profiler on itself); returns (bytes or None, where the number comes from)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if (world == 1 and args.size == 512 and args.samples == 64 and args.fill == "full" and not args.early_term and not args.occ_cull
            and not args.split_f16 and args.occupancy is None and os.path.exists(p)):
        j = json.load(open(p))
        return j["hbm_bytes_per_launch"], j.get("source", "profiles/pmc_traffic.json (rocprofv3 --pmc passes, profile-derived, not measured in this run)")
    return None, None


def _timed_sample(render, rays_h, target_s):
    """Time `render(rays)` on a bounded, evenly spaced sample of the rays that costs about target_s; returns (rays, seconds)."""
    n = rays_h.shape[0]
    probe = rays_h[:: max(1, n // 256)][:256]
    render(probe)                                     # thread start-up, page faults
    t0 = time.perf_counter()
    render(probe)
    per_ray = (time.perf_counter() - t0) / probe.shape[0]
    m = int(max(256, min(n, target_s / max(per_ray, 1e-9))))
    for _ in range(3):                                # re-size until the sample costs about target_s
        sample = rays_h[:: max(1, n // m)][:m]
        t0 = time.perf_counter()
        render(sample)
        dt = time.perf_counter() - t0
        if dt >= 0.6 * target_s or sample.shape[0] >= n:
            break
        m = int(min(n, m * target_s / max(dt, 1e-9)))
    if sample.shape[0] >= n and dt < 0.5 * target_s:  # the whole frame is cheaper than the target: time several passes of it
        reps = int(min(50, max(2, target_s / max(dt, 1e-9))))
        t0 = time.perf_counter()
        for _ in range(reps):
            render(sample)
        return sample.shape[0] * reps, time.perf_counter() - t0
    return sample.shape[0], dt


def cpu_model():
    """The host CPU's model string (/proc/cpuinfo) and socket count: what `cores` are cores OF."""
    try:
        names, sockets = [], set()
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                names.append(l.split(":", 1)[1].strip())
            elif l.startswith("physical id"):
                sockets.add(l.split(":", 1)[1].strip())
        return f"{names[0]} ({len(names)} hardware threads, {max(1, len(sockets))} socket(s))" if names else None
    except OSError:
        return None


def cpu_baseline(sc, rays_h, S, target_s):
    """The same workload on this host's cores (the Python reference cannot travel), on bounded, evenly spaced samples of the same
    rays, by two CPU programs:
      value / kind "port-blocked": oracle/gpnerf_cpu_blocked.c, the throughput twin -- blocks of 64+ samples through every layer as
        [out x in] x [in x block] products on 16-float vectors (AVX-512 where the host has it), channels-last gathers, -O3
        -march=native, OpenMP over all cores; checked against the oracle to 1e-5 (tests/test_cpu_blocked.py).  This is the number a
        GPU/CPU ratio should be read against;
      scalar_oracle / kind "port": oracle/gpnerf_oracle.c, the op-for-op scalar restatement the parity tests use (built -O2, no
        contraction: right for a checker, slow by construction)."""
    from oracle import blocked, oracle
    half = 0.5 * target_s
    fr = blocked.Frame(sc)                            # per-frame preparation (channels-last copies), not timed -- as the GPU Frame
    hw_threads = blocked.max_threads()

    def best_threads(render_t, hi):
        """Thread count with the highest rate on a short probe: the GPU boxes show more hardware threads than the job may use at
        once (rates FALL from 32 to 128 threads there), so the count is measured, not assumed."""
        best, best_rate, th = hi, 0.0, hi
        while th >= 1:
            probe = rays_h[:: max(1, rays_h.shape[0] // (512 * th))][: 512 * th]
            render_t(probe[: 64 * th], th)
            t0 = time.perf_counter()
            render_t(probe, th)
            rate = probe.shape[0] / (time.perf_counter() - t0)
            if rate > best_rate:
                best, best_rate = th, rate
            if th == 1 or rate < 0.7 * best_rate:
                break
            th //= 2
        return best

    threads = best_threads(lambda r, t: blocked.render(fr, r, S, want=(), n_threads=t), hw_threads)
    o_threads = best_threads(lambda r, t: oracle.render(sc, S, rays=r, want_weights=False, n_threads=t), hw_threads)
    nb, tb = _timed_sample(lambda r: blocked.render(fr, r, S, want=(), n_threads=threads), rays_h, half)
    no, to = _timed_sample(lambda r: oracle.render(sc, S, rays=r, want_weights=False, n_threads=o_threads), rays_h, half)
    check = rays_h[:: max(1, rays_h.shape[0] // 512)][:512]
    a, b = blocked.render(fr, check, S, want=()), oracle.render(sc, S, rays=check, want_weights=False)
    err = {k: float(np.abs(a[k] - b[k]).max()) for k in ("rgb_map", "depth_map", "acc_map")}
    n_frame = rays_h.shape[0]
    return {"value": nb / tb, "unit": "rays/s", "cores": threads, "kind": "port-blocked",
            "ms_per_frame": n_frame / (nb / tb) * 1e3,
            "gflops": nb / tb * S * FLOP_PER_SAMPLE / 1e9,
            "sample": f"{nb} evenly spaced rays of the same frame x {S} samples, {tb:.1f} s on {threads} OpenMP threads "
                      f"(oracle/gpnerf_cpu_blocked.c, gcc -O3 -march=native, channels-last frame prepared outside the timed call)",
            "max_abs_vs_scalar_oracle": err,
            "threads_available": hw_threads, "cpu_model": cpu_model(),
            "scalar_oracle": {"value": no / to, "unit": "rays/s", "cores": o_threads, "kind": "port",
                              "ms_per_frame": n_frame / (no / to) * 1e3,
                              "sample": f"{no} evenly spaced rays x {S} samples, {to:.1f} s on {o_threads} OpenMP threads "
                                        f"(oracle/gpnerf_oracle.c, the parity checker: scalar, -O2, no FMA contraction)"}}


def cpu_config1(seed):
    """BASELINE.json configs[0] on the CPU: every ray of a 64x64 crop, 32 samples per ray, median of 3 runs after a warm-up, by both
    CPU programs (see cpu_baseline)."""
    from oracle import blocked, oracle
    syn = importlib.import_module("gp-nerf_amd.synthetic")
    sc = syn.make_scene(H=64, W=64, seed=seed, fill="full", pose="identity")
    rays = np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32)
    fr = blocked.Frame(sc)

    def med(fn):
        fn()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    tb = min(blocked.max_threads(), 32)               # 4 096 rays: more threads than that only add start-up time
    db = med(lambda: blocked.render(fr, rays, 32, want=(), n_threads=tb))
    do = med(lambda: oracle.render(sc, 32, rays=rays, want_weights=False, n_threads=tb))
    return {"value": rays.shape[0] / db, "unit": "rays/s", "ms_per_frame": db * 1e3, "rays": int(rays.shape[0]), "cores": tb,
            "kind": "port-blocked", "scalar_oracle": {"value": rays.shape[0] / do, "ms_per_frame": do * 1e3, "kind": "port"}}


if __name__ == "__main__":
    main()
