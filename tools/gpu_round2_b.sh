# round-2 profiles: rocprofv3 kernel stats of the bench (headline + producers via the Renderer.render leg), configs[2] stats,
# the image encoder alone, and PMC passes (separate runs, counters only) for headline and configs[2]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r2b; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $o/bench_default.json 2> $o/stats_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_c3 -- python3 bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/bench_c3.json 2> $o/stats_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_encoder -- python3 tools/encoder_probe.py > $o/encoder.log 2>&1
for f in $(find $o -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-200; done
rm -rf gpurun_out/pmc_r02_default gpurun_out/pmc_r02_c3
bash tools/pmc_passes.sh r02_default --no-extras | tail -3
bash tools/pmc_passes.sh r02_c3 --samples 128 --early-term --no-extras | tail -3
