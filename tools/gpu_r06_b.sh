# round 6, step b: the new step_stats accounting + the quarantined build: parity of the exits, the contract test, the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6b; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -5 | tee $o/gpu_tests.txt
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err
tail -c 400 $o/bench_default.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r6b/bench_default.json"))
r = j["roofline"]
print(round(j["value"]), j["ms_per_step"], "frac", r["frac"], "alg", r["algorithmic_rate"]["over_peak"], "dense", r.get("dense_ms"), r.get("dense_frac"))
print(r["exits"])
print(j["trained_like"])
PY
