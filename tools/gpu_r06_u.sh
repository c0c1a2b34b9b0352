cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6u; rm -rf $o; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $o/gpu_tests.txt
for a in "--size 1024" "--fill survey" "--size 576" "--fold" "--size 320"; do
  timeout 300 python bench.py --steps 10 --warmup 3 $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$a', round(j['ms_per_step'],3), 'frac', round(r['frac'],4), 'dense', r.get('dense_ms'), r.get('dense_frac'))" | tee -a $o/benches.txt
done
