# round-6 baseline on HEAD: GPU suite + the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6a; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $o/gpu_tests.txt
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err
tail -c 600 $o/bench_default.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r6a/bench_default.json"))
print(round(j["value"]), j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["frac_of_work_done"])
print({k: (v.get("kernel_ms") if isinstance(v, dict) else None) for k, v in j["beside_headline"].items()})
PY
