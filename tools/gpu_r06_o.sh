cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6o; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_configs.py tests/test_gpu_sparse_conv.py tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -5 | tee $o/gpu_tests.txt
timeout 300 python bench.py --steps 10 --warmup 3 --split-f16 --no-cpu-baseline --no-extras > $o/bench_split_guarded.json 2> $o/err1.txt
timeout 300 python bench.py --steps 10 --warmup 3 --split-f16 --no-guard --no-cpu-baseline --no-extras > $o/bench_split_unguarded.json 2> $o/err2.txt
timeout 300 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --split-f16 --no-cpu-baseline --no-extras > $o/bench_c3_split.json 2> $o/err3.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $o/bench_default.json 2> $o/err4.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6o/bench_*.json")):
    try:
        j = json.load(open(f)); print(f.split("/")[-1], round(j["value"]), round(j["ms_per_step"], 3), round(j["roofline"]["frac"], 4))
    except Exception as e:
        print(f, "ERR", e)
PY
