cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6q; rm -rf $o; mkdir -p $o
source tools/diag_env.sh
run() { python bench.py --steps 10 --warmup 3 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', '$2', round(j['ms_per_step'],3))"; }
for sz in 272 300 320 384 448 200; do
  run default "--size $sz"; GPNERF_SPLIT=1 run split1 "--size $sz"
done 2>&1 | tee $o/midsize.txt
