cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6v; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_bench_contract.py -m gpu -q -x 2>&1 | tail -8 | tee $o/gpu_tests.txt
for a in "--samples 128 --early-term" "--fill survey" "--size 576" "" "--samples 128 --early-term --fold"; do
  timeout 300 python bench.py --steps 10 --warmup 3 $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$a', round(j['ms_per_step'],3), 'frac', round(r['frac'],4), 'dense', r.get('dense_ms'), r.get('dense_frac'))" | tee -a $o/benches.txt
done
timeout 600 python tools/et_sweep.py 30 2>&1 | tail -3 | tee $o/et_sweep.txt
timeout 600 python tools/defer_sweep.py 40 2>&1 | tail -3 | tee $o/defer_sweep.txt
