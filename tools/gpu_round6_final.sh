# round-6 measurements: GPU suite, benches, rocprofv3 kernel stats, PMC passes (separate runs, counters only), probes.
# -> gpurun_out/r6f, copied to profiles/r06 by tools/collect_round6.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6f; rm -rf $o; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $o/gpu_tests.txt
# the headline's PMC passes first: roofline.traffic of the bench lines below is read from profiles/pmc_traffic.json, which is derived here
# from THIS collection's counters (tools/collect_round6.sh derives the committed copy from the same summary)
rm -rf gpurun_out/pmc_r06_default
PMC_KERNEL="render_fused_kernel<0, false, false, true, true, true>+colour_accumulate_kernel" bash tools/pmc_passes.sh r06_default --no-extras | tail -2
python3 tools/pmc_summary.py gpurun_out/pmc_r06_default 'render_fused_kernel<0, false, false, false, false, false>' summary_dense.json > /dev/null
python3 tools/pmc_derive.py gpurun_out/pmc_r06_default/summary.json "512x512x64 full fill, API output set, patch order, reference-order form, bit-exact exits on (default)" \
    "render_fused_kernel<0, false, false, true, true, true> + colour_accumulate_kernel (one call)" "$(cat gpurun_out/HEAD_COMMIT 2>/dev/null || echo HEAD)" $o/pmc_headline_on_box.json --traffic profiles/pmc_traffic.json > /dev/null
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 3 --fold --no-cpu-baseline --no-extras > $o/bench_folded.json 2> $o/bench_folded.err
timeout 300 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/bench_c3.json 2> $o/bench_c3.err
timeout 300 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --split-f16 --no-cpu-baseline --no-extras > $o/bench_c3_split.json 2> $o/bench_c3s.err
timeout 300 python bench.py --steps 10 --warmup 3 --split-f16 --no-cpu-baseline --no-extras > $o/bench_split_guarded.json 2> $o/bench_split.err
timeout 300 python bench.py --steps 10 --warmup 3 --size 1024 --no-cpu-baseline --no-extras > $o/bench_1024.json 2> $o/bench_1024.err
timeout 300 python bench.py --steps 10 --warmup 3 --size 64 --samples 32 --no-cpu-baseline --no-extras > $o/bench_64.json 2> $o/bench_64.err
timeout 300 python bench.py --steps 10 --warmup 3 --fill survey --no-cpu-baseline --no-extras > $o/bench_survey.json 2> $o/bench_survey.err
timeout 300 python bench.py --steps 10 --warmup 3 --occ-cull --occupancy 0.1 --outputs light --no-cpu-baseline --no-extras > $o/bench_cull10.json 2> $o/bench_cull.err
GPNERF_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 > $o/bench_8ranks_gloo_dry_run.json 2> $o/bench_8ranks.err
timeout 300 python tools/trained_like_report.py > $o/trained_like.txt 2>&1
timeout 200 python tools/e2e512_probe.py > $o/e2e512_probe.txt 2>&1
timeout 200 python tools/probes/exact_encoder_time.py > $o/encoder_forms.txt 2>&1
timeout 200 python tools/probes/render_glue.py > $o/render_glue.txt 2>&1
timeout 200 python tools/probes/eval_loop_time.py 2>&1 | grep -v "^ssim\|^mse\|^psnr" > $o/eval_loop.txt
timeout 200 python tools/probes/demo_body_time.py > $o/demo_body.txt 2>&1
timeout 200 python tools/time_render_api.py > $o/render_api.txt 2>&1
timeout 200 python tools/time_survey_api.py 20 > $o/render_api_survey.txt 2>&1
timeout 300 python tools/probes/skip_probe.py > $o/skip_probe.txt 2>&1
timeout 600 python tools/parity_sweep.py 300 > $o/parity_sweep.txt 2>&1
timeout 600 python tools/et_sweep.py 50 > $o/et_sweep.txt 2>&1
timeout 600 python tools/defer_sweep.py 150 > $o/defer_sweep.txt 2>&1
timeout 600 python tools/producers_sweep.py 40 > $o/producers_sweep.txt 2>&1
(cd tools/micro && hipcc -O3 --offload-arch=gfx950 -o /tmp/aph asm_producer_hazards.hip 2>/dev/null && /tmp/aph) > $o/asm_producer_hazards.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6f/bench_*.json")):
    try:
        j = json.load(open(f)); r = j["roofline"]
        print(f.split("/")[-1], round(j["value"]), round(j["ms_per_step"], 3), "frac", round(r["frac"], 4), "dense", r.get("dense_ms"), r.get("dense_frac"), j.get("early_term", {}).get("samples_evaluated_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_headline -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $o/prof_bench_headline.json 2> $o/stats_headline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $o/prof_bench_default.json 2> $o/stats_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_c3 -- python3 bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > $o/prof_bench_c3.json 2> $o/stats_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_survey -- python3 bench.py --steps 10 --warmup 3 --fill survey --no-cpu-baseline --no-extras > $o/prof_bench_survey.json 2> $o/stats_survey.err
for f in $(find $o -name "*kernel_stats.csv"); do echo "== $f"; head -4 $f | cut -c1-160; done
rm -rf gpurun_out/pmc_r06_c3 gpurun_out/pmc_r06_survey
# the kernels of one gpnerf_render_fused call (last name: once per call).  Template arguments <form, chained, culled, deferred, listed, unified>
PMC_KERNEL="render_fused_kernel<0, true, false, true, true, false>+colour_units_kernel<0>+colour_accumulate_kernel" bash tools/pmc_passes.sh r06_c3 --samples 128 --early-term --no-extras | tail -2
PMC_KERNEL="render_fused_kernel<0, true, false, true, true, true>+colour_accumulate_kernel" bash tools/pmc_passes.sh r06_survey --fill survey --no-extras | tail -2
find gpurun_out/pmc_r06_* -name "*.csv" -size +2M -delete
# the three ways of running the colour branch of the samples that need it, on ONE box (lab library knobs): the launch's own wavefronts
# evaluate the launch's list once they have no tile left (default where the launch shape allows), a second kernel evaluates it
# (GPNERF_UNIFIED=0), every wavefront evaluates its own samples in passes of 32 inside the tile loop (GPNERF_FRAME_DEFER=0, round 5)
(
source tools/diag_env.sh
run() { python bench.py --steps 10 --warmup 3 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 |', '$2', '|', round(j['ms_per_step'],3), 'ms')"; }
for a in "" "--samples 128 --early-term" "--fill survey" "--size 1024" "--size 384" "--size 320" "--size 576" "--fold"; do
  run "default" "$a"; GPNERF_UNIFIED=0 run "list + second kernel" "$a"; GPNERF_FRAME_DEFER=0 run "passes per wavefront" "$a"
done
) > $o/frame_level_ab.txt 2>&1
timeout 300 python tools/wave_times.py 512 64 2>&1 | grep -v amdgpu > $o/wave_times_listed.txt
GPNERF_DEBUG=1 GPNERF_UNIFIED=0 timeout 300 python tools/wave_times.py 512 64 2>&1 | grep -v amdgpu > $o/wave_times_second_kernel.txt
GPNERF_DEBUG=1 GPNERF_FRAME_DEFER=0 timeout 300 python tools/wave_times.py 512 64 2>&1 | grep -v amdgpu > $o/wave_times_wavefront_passes.txt
