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
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port", "port-blocked") and c["scalar_oracle"]["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "rays/s" and c["sample"]
    assert "beside_headline" in j and "split_f16_api_outputs_patch_order" in j["beside_headline"]


@pytest.mark.gpu
def test_bench_early_termination_line():
    j = run_bench("--steps", "2", "--warmup", "1", "--size", "256", "--samples", "48", "--early-term", "--no-cpu-baseline", "--no-extras")
    assert j["config"]["early_term"] is True and 0.0 < j["early_term"]["samples_evaluated_frac"] <= 1.0
    assert j["roofline"]["samples_evaluated_frac"] == j["early_term"]["samples_evaluated_frac"]


def test_bench_refuses_a_world_size_mismatch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
