cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6q; rm -rf $o; mkdir -p $o
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -6 | cut -c1-250 | tee $o/gpu_tests.txt
for a in "" "--size 1024" "--fold"; do
timeout 300 python bench.py --steps 10 --warmup 3 $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$a', round(j['ms_per_step'],3), 'frac', round(r['frac'],4), 'dense', r.get('dense_ms'), r.get('dense_same_bits'))"
done | tee $o/bench.txt
timeout 600 python tools/defer_sweep.py 30 2>&1 | tail -2
