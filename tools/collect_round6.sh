#!/bin/bash
# copy what tools/gpu_round6_final.sh left under gpurun_out/ into profiles/r06/ (run here, after the gpurun call has merged)
set -e
cd "$(dirname "$0")/.."
o=gpurun_out/r6f; d=profiles/r06; c=${1:-$(git rev-parse --short HEAD)}
mkdir -p $d
cp $o/prof_bench_headline.json $d/a_bench_headline.json
cp $(ls -t $o/stats_headline/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_headline.csv
cp $(ls -t $o/stats_default/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_headline_and_producers.csv
cp $o/prof_bench_c3.json $d/a_bench_config3_early_term.json
cp $(ls -t $o/stats_c3/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_config3_early_term.csv
cp $o/prof_bench_survey.json $d/a_bench_survey_frame.json
cp $(ls -t $o/stats_survey/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_survey_frame.csv
for n in default folded c3 c3_split split_guarded 1024 64 survey cull10 8ranks_gloo_dry_run; do [ -s $o/bench_$n.json ] && cp $o/bench_$n.json $d/d_bench_$n.json; done
cp $o/gpu_tests.txt $d/e_gpu_tests.txt
grep -v amdgpu $o/trained_like.txt > $d/b_trained_like.txt
grep -v amdgpu $o/e2e512_probe.txt > $d/e_e2e512_probe.txt
grep -v amdgpu $o/encoder_forms.txt > $d/e_encoder_forms.txt
grep -v amdgpu $o/render_glue.txt > $d/g_render_glue.txt
grep -v amdgpu $o/render_api.txt > $d/e_render_api.txt
grep -v amdgpu $o/render_api_survey.txt > $d/e_render_api_survey_frame.txt
grep -v amdgpu $o/eval_loop.txt > $d/d_eval_loop.txt
grep -v amdgpu $o/demo_body.txt > $d/g_demo_render_body_frame.json
grep -v amdgpu $o/skip_probe.txt > $d/c_exits_on_off.txt
grep -v amdgpu $o/parity_sweep.txt > $d/f_parity_sweep.txt
grep -v amdgpu $o/et_sweep.txt > $d/f_et_sweep.txt
grep -v amdgpu $o/defer_sweep.txt > $d/f_defer_sweep.txt
grep -v amdgpu $o/producers_sweep.txt > $d/f_producers_sweep.txt
cp $o/asm_producer_hazards.txt $d/i_asm_producer_hazards.txt
cp $o/frame_level_ab.txt $d/c_frame_level_deferral_ab.txt
{ echo '== default: colour work listed per launch, evaluated by the launch (the time a wavefront finished its LAST TILE; it then evaluates list units until the launch ends)'; cat $o/wave_times_listed.txt; echo; echo '== list + second kernel (GPNERF_UNIFIED=0): exits of the sample-loop kernel'; cat $o/wave_times_second_kernel.txt; echo; echo '== colour passes per wavefront (GPNERF_FRAME_DEFER=0, round 5): exits of the one kernel'; cat $o/wave_times_wavefront_passes.txt; } > $d/c_wave_exit_times.txt
python tools/pmc_derive.py gpurun_out/pmc_r06_default/summary.json "512x512x64 full fill, API output set, patch order, reference-order form, bit-exact exits on (default)" \
    "render_fused_kernel<0, false, false, true, true, true> (sample loop, then the launch's own wavefronts evaluate its colour list) + colour_accumulate_kernel (one call)" $c $d/b_pmc_summary_headline.json --traffic profiles/pmc_traffic.json | grep -E "busy|hbm_bytes|l2_hit|valu_active"
python tools/pmc_derive.py gpurun_out/pmc_r06_default/summary_dense.json "512x512x64 full fill, API output set, patch order, reference-order form, GPNERF_FLAG_NO_EXITS (every layer of every sample: roofline.dense_*)" \
    "render_fused_kernel<0, false, false, false, false, false>" $c $d/b_pmc_summary_headline_dense.json | grep -E "busy|hbm_bytes|l2_hit|valu_active"
python tools/pmc_derive.py gpurun_out/pmc_r06_c3/summary.json "512x512x128 early termination (configs[2]), reference-order form, per call (six segment launches + the colour list)" \
    "render_fused_kernel<0, true, false, true, true, false> x 6 + colour_units_kernel<0> + colour_accumulate_kernel" $c $d/b_pmc_summary_config3_early_term.json | grep -E "busy|hbm_bytes|l2_hit"
python tools/pmc_derive.py gpurun_out/pmc_r06_survey/summary.json "512x512x64 survey fill (73 689 rays), API output set, patch order, reference-order form" \
    "render_fused_kernel<0, true, false, true, true, true> (one launch: 2 048 whole tiles + 255 tiles as eight-samples-per-step units, then its colour list) + colour_accumulate_kernel" $c $d/b_pmc_summary_survey_frame.json | grep -E "busy|hbm_bytes|l2_hit|valu_active"
python tools/resource_table.py $d > /dev/null 2>&1 || true
rm -f $d/h_kernel_resources_wip.md
ls $d
