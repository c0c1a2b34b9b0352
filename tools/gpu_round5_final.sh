# round-5 measurements: GPU suite, benches (reference-order default + the folded fast form), rocprofv3 kernel stats, PMC passes
# (separate runs, counters only), probes.  -> gpurun_out/r5f, copied to profiles/r05 by tools/collect_round5.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r5f; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $o/gpu_tests.txt
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 3 --fold --no-cpu-baseline --no-extras > $o/bench_folded.json 2> $o/bench_folded.err
timeout 300 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/bench_c3.json 2> $o/bench_c3.err
timeout 300 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --fold --no-cpu-baseline --no-extras > $o/bench_c3_folded.json 2> $o/bench_c3f.err
timeout 300 python bench.py --steps 10 --warmup 3 --split-f16 --no-cpu-baseline --no-extras > $o/bench_split_guarded.json 2> $o/bench_split.err
timeout 300 python bench.py --steps 10 --warmup 3 --size 1024 --no-cpu-baseline --no-extras > $o/bench_1024.json 2> $o/bench_1024.err
timeout 300 python bench.py --steps 10 --warmup 3 --size 64 --samples 32 --no-cpu-baseline --no-extras > $o/bench_64.json 2> $o/bench_64.err
timeout 300 python bench.py --steps 10 --warmup 3 --fill survey --no-cpu-baseline --no-extras > $o/bench_survey.json 2> $o/bench_survey.err
timeout 300 python bench.py --steps 10 --warmup 3 --fill survey --fold --no-cpu-baseline --no-extras > $o/bench_survey_folded.json 2> $o/bench_surveyf.err
timeout 300 python bench.py --steps 10 --warmup 3 --occ-cull --occupancy 0.1 --outputs light --no-cpu-baseline --no-extras > $o/bench_cull10.json 2> $o/bench_cull.err
GPNERF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $o/bench_2ranks_gloo_dry_run.json 2> $o/bench_2ranks.err
timeout 300 python tools/trained_like_report.py > $o/trained_like.txt 2>&1
timeout 200 python tools/e2e512_probe.py > $o/e2e512_probe.txt 2>&1
timeout 200 python tools/probes/exact_encoder_time.py > $o/exact_encoder_time.txt 2>&1
GPNERF_DEBUG=1 GPNERF_EXACT_UNTILED=1 timeout 200 python tools/probes/exact_encoder_time.py 2>&1 | grep forward_exact >> $o/exact_encoder_time.txt
timeout 200 python tools/probes/overlap_probe.py > $o/overlap_probe.txt 2>&1
timeout 200 python tools/probes/eval_loop_time.py 2>&1 | grep -v "^ssim\|^mse\|^psnr" > $o/eval_loop.txt
timeout 200 python tools/probes/exchange_local_cost.py > $o/exchange_local_cost.txt 2>&1
timeout 200 python tools/probes/demo_body_time.py > $o/demo_body.txt 2>&1
timeout 200 python tools/time_render_api.py > $o/render_api.txt 2>&1
timeout 200 python tools/time_survey_api.py 20 > $o/render_api_survey.txt 2>&1
timeout 200 python tools/probes/encoder_time.py > $o/encoder_time.txt 2>&1
timeout 200 python tools/stamps.py 512 > $o/stamps_default.txt 2>&1
timeout 300 python tools/probes/skip_probe.py > $o/skip_probe.txt 2>&1
timeout 200 python tools/probes/defer_debug.py > $o/defer_debug.txt 2>&1
timeout 600 python tools/parity_sweep.py 300 > $o/parity_sweep.txt 2>&1
timeout 600 python tools/et_sweep.py 50 > $o/et_sweep.txt 2>&1
timeout 600 python tools/defer_sweep.py 150 > $o/defer_sweep.txt 2>&1
timeout 600 python tools/producers_sweep.py 40 > $o/producers_sweep.txt 2>&1
(cd tools/micro && hipcc -O2 --offload-arch=gfx950 -o /tmp/mdo mfma_dst_overlap.hip 2>/dev/null && /tmp/mdo; hipcc -O2 --offload-arch=gfx950 -o /tmp/pswap permlane32_swap.hip 2>/dev/null && /tmp/pswap) > $o/micro.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5f/bench_*.json")):
    try:
        j = json.load(open(f)); print(f.split("/")[-1], round(j["value"]), round(j["ms_per_step"], 3), round(j["roofline"]["frac"], 4), j.get("early_term", {}).get("samples_evaluated_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -2 $o/render_api.txt; tail -1 $o/render_api_survey.txt; tail -1 $o/encoder_time.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_headline -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $o/prof_bench_headline.json 2> $o/stats_headline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_folded -- python3 bench.py --steps 20 --warmup 3 --fold --no-cpu-baseline --no-extras > $o/prof_bench_folded.json 2> $o/stats_folded.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $o/prof_bench_default.json 2> $o/stats_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_c3 -- python3 bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/prof_bench_c3.json 2> $o/stats_c3.err
rocprofv3 --kernel-trace --output-format csv -d $o/trace_eval_loop -- python3 tools/probes/eval_loop_time.py 4 > $o/trace_eval_loop.log 2>&1
for f in $(find $o -name "*kernel_stats.csv"); do echo "== $f"; head -4 $f | cut -c1-160; done
rm -rf gpurun_out/pmc_r05_default gpurun_out/pmc_r05_folded gpurun_out/pmc_r05_c3 gpurun_out/pmc_r05_survey
bash tools/pmc_passes.sh r05_default --no-extras | tail -2
bash tools/pmc_passes.sh r05_folded --fold --no-extras | tail -2
bash tools/pmc_passes.sh r05_c3 --samples 128 --early-term --no-extras | tail -2
bash tools/pmc_passes.sh r05_survey --fill survey --no-extras | tail -2
