#!/bin/bash
# copy what tools/gpu_round4_final.sh left under gpurun_out/ into profiles/r04/ (run here, after the gpurun call has merged)
set -e
cd "$(dirname "$0")/.."
o=gpurun_out/r4f; d=profiles/r04; c=${1:-$(git rev-parse --short HEAD)}
mkdir -p $d
cp $o/prof_bench_headline.json $d/a_bench_headline.json
cp $(ls -t $o/stats_headline/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_headline.csv
cp $o/prof_bench_default.json $d/a_bench_headline_with_extras.json
cp $(ls -t $o/stats_default/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_headline_and_producers.csv
cp $o/prof_bench_c3.json $d/a_bench_config3_early_term.json
cp $(ls -t $o/stats_c3/runc/*kernel_stats.csv | head -1) $d/a_kernel_stats_config3_early_term.csv
cp $(ls -t $o/stats_encoder/runc/*kernel_stats.csv | head -1) $d/c_kernel_stats_encoder.csv
cp $(ls -t $o/stats_survey_api/runc/*kernel_stats.csv | head -1) $d/c_kernel_stats_renderer_api_survey_frame.csv
for n in default c3 split_guarded 1024 64 survey cull10 2ranks_gloo_dry_run; do cp $o/bench_$n.json $d/d_bench_$n.json; done
cp $o/gpu_tests.txt $d/e_gpu_tests.txt
cp $o/render_api.txt $d/e_render_api.txt
cp $o/render_api_survey.txt $d/e_render_api_survey_frame.txt
cp $o/render_phases_survey.txt $d/e_render_phases_survey_frame.txt
cp $o/encoder_time.txt $d/e_encoder_time.txt
cp $o/encoder_views_time.txt $d/m_encoder_views_time.txt
cp $o/encoder_error.txt $d/e_encoder_error.txt
cp $o/e2e512_probe.txt $d/e_e2e512_probe.txt
cp $o/trained_like.txt $d/j_trained_like.txt
cp $o/parity_sweep.txt $d/f_parity_sweep.txt
cp $o/et_sweep.txt $d/f_et_sweep.txt
cp $o/producers_sweep.txt $d/f_producers_sweep.txt
cp $o/view_fold_proxy.txt $d/f_view_fold_proxy.txt
cp $o/c3_traffic_per_level.txt $d/f_config3_traffic_per_level.txt
cp $o/config_sweep.txt $d/g_config_sweep.txt
python tools/pmc_derive.py gpurun_out/pmc_r04_default/summary.json "512x512x64 full fill, API output set, patch order" \
    "render_fused_kernel<4, false, false>" $c $d/b_pmc_summary_headline.json --traffic profiles/pmc_traffic.json | grep -E "busy|hbm_bytes|l2_hit|valu_active"
python tools/pmc_derive.py gpurun_out/pmc_r04_c3/summary.json "512x512x128 early termination (configs[2]), mean over the 8 segment launches of a frame" \
    "render_fused_kernel<4, true, false>" $c $d/b_pmc_summary_config3_early_term.json | grep -E "busy|hbm_bytes|l2_hit"
python tools/pmc_derive.py gpurun_out/pmc_r04_split/summary.json "512x512x64 split-precision form with range guard" \
    "render_fused_kernel<2, false, false>" $c $d/b_pmc_summary_split_guarded.json | grep -E "busy|hbm_bytes"
python tools/pmc_derive.py gpurun_out/pmc_r04_survey/summary.json "512x512x64 survey fill (73 689 rays), API output set, patch order" \
    "render_fused_kernel<4, true, false> (one launch: 2 048 whole tiles + 255 tiles as eight-samples-per-step units)" $c $d/b_pmc_summary_survey_frame.json | grep -E "busy|hbm_bytes|l2_hit|valu_active"
python tools/resource_table.py $d > /dev/null 2>&1 || true
tail -7 $d/e_render_phases_survey_frame.txt
tail -1 $d/e_render_api_survey_frame.txt
