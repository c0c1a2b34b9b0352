# round-2 final measurements: GPU suite, benches, rocprofv3 kernel stats, PMC passes (separate runs, counters only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r2f; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $o/gpu_tests.txt
timeout 400 python bench.py > $o/bench_default.json 2> $o/bench_default.err
timeout 300 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/bench_c3.json 2> $o/bench_c3.err
timeout 300 python bench.py --steps 10 --warmup 3 --split-f16 --no-cpu-baseline --no-extras > $o/bench_split_guarded.json 2> $o/bench_split.err
timeout 300 python bench.py --steps 10 --warmup 3 --size 1024 --no-cpu-baseline --no-extras > $o/bench_1024.json 2> $o/bench_1024.err
timeout 300 python bench.py --steps 10 --warmup 3 --size 64 --samples 32 --no-cpu-baseline --no-extras > $o/bench_64.json 2> $o/bench_64.err
timeout 300 python bench.py --steps 10 --warmup 3 --occ-cull --occupancy 0.1 --outputs light --no-cpu-baseline --no-extras > $o/bench_cull10.json 2> $o/bench_cull.err
timeout 200 python tools/time_render_api.py > $o/render_api.txt 2>&1
timeout 200 python tools/guard_probe.py > $o/guard_probe.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2f/bench_*.json")):
    try:
        j = json.load(open(f)); print(f.split("/")[-1], round(j["value"]), round(j["ms_per_step"], 3), round(j["roofline"]["frac"], 4), j.get("early_term", {}).get("samples_evaluated_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $o/render_api.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $o/prof_bench_default.json 2> $o/stats_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_c3 -- python3 bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/prof_bench_c3.json 2> $o/stats_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_encoder -- python3 tools/encoder_probe.py > $o/encoder.log 2>&1
for f in $(find $o -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-160; done
rm -rf gpurun_out/pmc_r02_default gpurun_out/pmc_r02_c3 gpurun_out/pmc_r02_split
bash tools/pmc_passes.sh r02_default --no-extras | tail -2
bash tools/pmc_passes.sh r02_c3 --samples 128 --early-term --no-extras | tail -2
bash tools/pmc_passes.sh r02_split --split-f16 --no-extras | tail -2
