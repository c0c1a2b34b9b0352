"""The driver's contract with bench.py: one JSON line with the fields it parses (metric, value, unit, n_gpus, steps, warmup,
ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline{bound, achieved, peak, unit, frac,
traffic}, cpu_baseline{value, unit, cores, kind, sample})."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints exactly one line"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_has_the_contracts_fields():
    j = run_bench("--steps", "2", "--warmup", "1", "--size", "64", "--samples", "16", "--cpu-seconds", "0.5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"] == "rays_per_sec" and j["unit"] == "rays/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - j["config"]["rays_total"] / (j["ms_per_step"] * 1e-3)) < 1e-3 * j["value"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    # a roofline FRACTION: the work the matrix pipe did over its peak; the rate of answers, which the bit-exact exits
    # can push beyond the peak, sits beside it, and so does the data-independent launch that evaluates every layer
    assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["dense_frac"] <= 1.0 and r["dense_ms"] > 0 and r["dense_same_bits"] is True
    assert r["algorithmic_rate"]["tflops"] >= r["achieved"] > 0 and r["flop_per_launch"] <= r["algorithmic_rate"]["flop_per_launch"]
    ex = r["exits"]
    assert 0.0 <= ex["flop_not_done_frac"] < 1.0 and 0.0 <= ex["colour_branch_not_run_frac"] <= 1.0 and 0.0 <= ex["opaque_tail_frac"] <= 1.0
    t = j["trained_like"]
    assert "error" not in t, t
    assert 0.0 < t["frac"] <= 1.0 and 0.0 < t["dense_frac"] <= 1.0 and t["dense_same_bits"] is True and t["kernel_ms"] <= t["dense_ms"] * 1.05
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port", "port-blocked") and c["scalar_oracle"]["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "rays/s" and c["sample"]
    assert "beside_headline" in j and "split_f16_api_outputs_patch_order" in j["beside_headline"]


@pytest.mark.gpu
def test_bench_early_termination_line():
    j = run_bench("--steps", "2", "--warmup", "1", "--size", "256", "--samples", "48", "--early-term", "--no-cpu-baseline", "--no-extras")
    assert j["config"]["early_term"] is True and 0.0 < j["early_term"]["samples_evaluated_frac"] <= 1.0
    assert j["roofline"]["samples_evaluated_frac"] == j["early_term"]["samples_evaluated_frac"]


def _env_without_ranks(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how a driver might issue the SCALE run) launches its two
    ranks itself -- RCCL when two devices are visible, the gloo dry run with both ranks on cuda:0 otherwise -- and relays rank 0's
    single line: the strong-scaling flow of one frame (libs/renders/BaseRender.py:160-184 as N ranks x one launch) plus the
    1024x1024 frame of BASELINE.json configs[3]."""
    import torch
    extra = {} if torch.cuda.device_count() >= 2 else {"GPNERF_BENCH_BACKEND": "gloo"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "128"],
                         capture_output=True, text=True, timeout=900, env=_env_without_ranks(**extra))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 2
    assert j["config"]["rays_total"] == 128 * 128 and 0 < j["config"]["rays_per_gpu"] < j["config"]["rays_total"]
    assert j["value"] > 0 and abs(j["value"] - j["config"]["rays_total"] / (j["ms_per_step"] * 1e-3)) < 1e-3 * j["value"]
    c4 = j["config4_1024"]
    assert c4["rays_total"] == 1024 * 1024 and c4["rays_per_rank"] * 2 >= c4["rays_total"] and c4["value"] > 0
    assert j["maps_finite"] is True and c4["maps_finite"] is True
    # who ran where: one report per rank, the device each sat on, its own kernel time
    assert [r["rank"] for r in j["ranks"]] == [0, 1] and all(r["device_name"] and r["kernel_ms"] > 0 for r in j["ranks"])
    assert j["distinct_devices"] == (2 if torch.cuda.device_count() >= 2 else 1)
    assert j["kernel_ms_per_rank"]["max"] >= j["kernel_ms_per_rank"]["min"] > 0


def test_bench_launch_line_is_the_drivers():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(4, 29511, ["--gpus", "4", "--steps", "3"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_bench_refuses_a_world_size_mismatch():
    """Under a launcher (WORLD_SIZE set) the process is a rank and must not start ranks of its own; a mismatch is an error."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env=_env_without_ranks(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_bench_self_launch_needs_a_gpu_per_rank_and_says_so():
    """No GPU here: the parent counts devices before starting anything and refuses a measured (nccl) run with fewer GPUs than ranks."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env=_env_without_ranks(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert out.returncode == 2 and "device(s) visible" in out.stderr and not out.stdout.strip()


def test_bench_self_launch_relays_one_line_and_the_childs_status(tmp_path, monkeypatch, capsys):
    """The parent half of `python bench.py --gpus N` without a launcher (no GPU needed): whatever it starts as its ranks, it passes
    rank 0's JSON line through on stdout, everything else to stderr, and returns the child's exit status."""
    sys.path.insert(0, ROOT)
    import bench
    child = tmp_path / "child.py"
    child.write_text("import sys\nprint('NOTE: some launcher chatter')\nprint('{\"metric\": \"rays_per_sec\", \"n_gpus\": 2}')\nsys.exit(7)\n")
    monkeypatch.setattr(bench, "launch_command", lambda gpus, port, argv: [sys.executable, str(child)])
    monkeypatch.setenv("GPNERF_BENCH_BACKEND", "gloo")          # (skips the device count: this is the dry-run backend)
    from types import SimpleNamespace as NS
    rc = bench.self_launch(NS(gpus=2))
    out = capsys.readouterr()
    assert rc == 7
    assert [l for l in out.out.splitlines() if l.strip()] == ['{"metric": "rays_per_sec", "n_gpus": 2}']
    assert "launcher chatter" in out.err


def _rank_report_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rep = dict(bench.device_identity(torch.device("cpu")), rank=rank, local_rank=rank, kernel_ms=1.0 + rank, exchange_ms=0.25 * (rank + 1), rays=1000 + rank)
    out = bench.summarize_ranks(bench.gather_rank_reports(rep, world))
    if rank == 0:
        q.put(out)
    dist.destroy_process_group()


def test_rank_reports_are_gathered_into_the_line():
    """N > 1: every rank's device identity, its own kernel time and the exchange's time reach rank 0 through ONE all_gather_object
    (gloo, world 2): `ranks` in rank order, `distinct_devices`, per-rank kernel spread, exchange time."""
    import socket

    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank_report_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = q.get(timeout=120)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and out["distinct_devices"] == 2          # two processes = two "devices" on the CPU
    assert out["kernel_ms_per_rank"] == {"min": 1.0, "max": 2.0, "mean": 1.5} and out["exchange_ms"]["max"] == 0.5
    assert all(k in out["ranks"][0] for k in ("device_index", "device_name", "pci_bus_id", "uuid", "rays"))
    sys.path.insert(0, ROOT)
    import bench
    one = bench.summarize_ranks([dict(rank=0, device_index=0, device_name="x", pci_bus_id="0000:05:00.0", uuid=None, kernel_ms=3.0, exchange_ms=None)])
    assert one["distinct_devices"] == 1 and one["exchange_ms"] is None
    two_on_one = bench.summarize_ranks([dict(rank=r, device_index=0, device_name="x", pci_bus_id="0000:05:00.0", uuid=None, kernel_ms=3.0, exchange_ms=None) for r in (0, 1)])
    assert two_on_one["distinct_devices"] == 1          # the gloo dry run's two ranks on cuda:0 show up as what they are
