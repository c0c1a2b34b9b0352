cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6r; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_bench_contract.py -m gpu -q -x 2>&1 | tail -8 | tee $o/gpu_tests.txt
timeout 600 python bench.py --no-cpu-baseline > $o/bench_default.json 2> $o/bench_default.err
python - <<'PY' | tee $o/summary.txt
import json
d = json.load(open("gpurun_out/r6r/bench_default.json"))
r = d["roofline"]
print("ms", d["ms_per_step"], "rays/s", d["value"], "frac", r["frac"], "kernel_ms", r.get("kernel_ms"), "dense", r.get("dense_ms"), r.get("dense_frac"))
print("trained_like", json.dumps(d.get("trained_like"))[:400])
print(json.dumps(d.get("beside_headline"))[:1500])
PY
timeout 300 python tools/defer_sweep.py 30 2>&1 | tail -4 | tee $o/defer_sweep.txt
