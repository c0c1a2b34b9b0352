#!/bin/bash
# copy what tools/gpu_round5_final.sh left under gpurun_out/ into profiles/r05/ (run here, after the gpurun call has merged)
set -e
cd "$(dirname "$0")/.."
o=gpurun_out/r5f; d=profiles/r05; c=${1:-$(git rev-parse --short HEAD)}
mkdir -p $d
cp $o/prof_bench_headline.json $d/a_bench_headline.json
cp $(ls -t $o/stats_headline/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_headline.csv
cp $o/prof_bench_folded.json $d/a_bench_headline_folded_form.json
cp $(ls -t $o/stats_folded/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_headline_folded_form.csv
cp $(ls -t $o/stats_default/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_headline_and_producers.csv
cp $o/prof_bench_c3.json $d/a_bench_config3_early_term.json
cp $(ls -t $o/stats_c3/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_config3_early_term.csv
for n in default folded c3 c3_folded split_guarded 1024 64 survey survey_folded cull10 2ranks_gloo_dry_run; do [ -s $o/bench_$n.json ] && cp $o/bench_$n.json $d/d_bench_$n.json; done
cp $o/gpu_tests.txt $d/e_gpu_tests.txt
cp $o/trained_like.txt $d/b_trained_like.txt
cp $o/e2e512_probe.txt $d/e_e2e512_probe.txt
cp $o/exact_encoder_time.txt $d/e_exact_encoder_time.txt
cp $o/render_api.txt $d/e_render_api.txt
cp $o/render_api_survey.txt $d/e_render_api_survey_frame.txt
cp $o/encoder_time.txt $d/e_encoder_time.txt
cp $o/demo_body.txt $d/g_demo_render_body_frame.json
cp $o/exchange_local_cost.txt $d/m_exchange_local_cost.txt
cp $o/parity_sweep.txt $d/f_parity_sweep.txt
cp $o/et_sweep.txt $d/f_et_sweep.txt
cp $o/producers_sweep.txt $d/f_producers_sweep.txt
[ -s $o/defer_sweep.txt ] && grep -v amdgpu $o/defer_sweep.txt > $d/f_defer_sweep.txt
cp $o/micro.txt $d/i_micro_mfma_overlap_permlane_swap.txt
[ -s $o/stamps_default.txt ] && grep -v amdgpu $o/stamps_default.txt > $d/c_stamps_headline.txt
[ -s $o/skip_probe.txt ] && grep -v amdgpu $o/skip_probe.txt > $d/c_exits_on_off.txt
[ -s $o/defer_debug.txt ] && grep -v amdgpu $o/defer_debug.txt > $d/c_deferred_vs_in_step_colour.txt
{ echo "# pipelined evaluation loop (VERDICT r4 next #2): measurements of $c"; echo;
  echo "## tools/probes/overlap_probe.py -- per-ray kernel of the ZJU-sized frame + the NEXT frame's encoder graph on a second stream, by reserved CUs"; cat $o/overlap_probe.txt | grep -v amdgpu;
  echo; echo "## tools/probes/eval_loop_time.py -- evaluator.evaluate_loop over 12 such frames, serial against pipelined (bench.py beside_headline.eval_loop)"; cat $o/eval_loop.txt | grep -v amdgpu; } > $d/d_pipeline.txt
python tools/trace_frames.py $(ls -t $o/trace_eval_loop/runc/*kernel_trace.csv | head -1) 3 > $d/d_pipeline_kernel_timeline.txt 2>&1 || true
python tools/pmc_derive.py gpurun_out/pmc_r05_default/summary.json "512x512x64 full fill, API output set, patch order, reference-order form, colour branch deferred (default)" \
    "render_fused_kernel<0, false, false, true>" $c $d/b_pmc_summary_headline.json --traffic profiles/pmc_traffic.json | grep -E "busy|hbm_bytes|l2_hit|valu_active"
python tools/pmc_derive.py gpurun_out/pmc_r05_folded/summary.json "512x512x64 full fill, API output set, patch order, folded fast form (--fold)" \
    "render_fused_kernel<4, false, false, true>" $c $d/b_pmc_summary_headline_folded_form.json | grep -E "busy|hbm_bytes|l2_hit|valu_active"
python tools/pmc_derive.py gpurun_out/pmc_r05_c3/summary.json "512x512x128 early termination (configs[2]), reference-order form, mean over the segment launches of a frame" \
    "render_fused_kernel<0, true, false, true>" $c $d/b_pmc_summary_config3_early_term.json | grep -E "busy|hbm_bytes|l2_hit"
python tools/pmc_derive.py gpurun_out/pmc_r05_survey/summary.json "512x512x64 survey fill (73 689 rays), API output set, patch order, reference-order form" \
    "render_fused_kernel<0, true, false, true> (one launch: 2 048 whole tiles + 255 tiles as eight-samples-per-step units)" $c $d/b_pmc_summary_survey_frame.json | grep -E "busy|hbm_bytes|l2_hit|valu_active"
python tools/resource_table.py $d > /dev/null 2>&1 || true
rm -f $d/h_kernel_resources_wip.md
ls $d
