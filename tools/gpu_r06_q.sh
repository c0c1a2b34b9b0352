cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6q; rm -rf $o; mkdir -p $o
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "frame_level" 2>&1 | tail -4 | cut -c1-250
for a in "--samples 128 --early-term" "--fill survey" "" "--samples 128 --early-term --fold"; do
timeout 120 python bench.py --steps 10 --warmup 3 $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$a', round(j['ms_per_step'],3), 'frac', round(r['frac'],4))"
done | tee $o/bench.txt
timeout 600 python tools/et_sweep.py 30 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_renderer.py -m gpu -q 2>&1 | tail -4 | cut -c1-250
