cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6s; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $o/bench.json 2> $o/err.txt
f=$(find $o/stats -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-220 | tee $o/kernel_stats_head.txt
timeout 300 python tools/wave_times.py 512 64 2>&1 | grep -v amdgpu | tee $o/wave_times_512.txt
