# kernel trace of Renderer.render with producers on the ZJU-sized (survey) frame: per-phase kernel time and idle gaps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/time_survey_api.py 10 2>&1 | tail -1
rm -rf gpurun_out/sva; mkdir -p gpurun_out/sva
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sva -- python3 tools/time_survey_api.py 4 > gpurun_out/sva/run.log 2>&1
f=$(find gpurun_out/sva -name "*kernel_trace.csv" | head -1)
python3 tools/trace_frames.py $f 6 | tail -32
