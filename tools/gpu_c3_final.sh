# configs[2] (512x512x128 + early termination): bench line, rocprofv3 kernel stats, PMC passes -> gpurun_out/r2f_c3/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r2f_c3; rm -rf $o gpurun_out/pmc_r02_c3; mkdir -p $o
timeout 300 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/bench_c3.json 2> $o/bench_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_c3 -- python3 bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/prof_bench_c3.json 2> $o/stats_c3.err
bash tools/pmc_passes.sh r02_c3 --samples 128 --early-term --no-extras | tail -2
python3 -c "
import json; j=json.load(open('$o/bench_c3.json')); print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['early_term']['samples_evaluated_frac'])"
