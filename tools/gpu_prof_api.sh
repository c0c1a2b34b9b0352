# rocprofv3 kernel stats of tools/time_render_api.py (dense, products-given and progressive Renderer.render legs): glue kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/api_prof; mkdir -p gpurun_out/api_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/api_prof -- python3 tools/time_render_api.py > gpurun_out/api_prof/run.log 2>&1
f=$(find gpurun_out/api_prof -name "*kernel_stats.csv" | head -1); grep -E "select_pixels|make_rays|occupancy|nonzero|index|sort|Sort|cfirst|images_to|vertex_att|stage_project|assign|mark_|dense_kernel|index_kernel|count_dup|merge_dup|conv_mfma|radix" $f | cut -c1-170
