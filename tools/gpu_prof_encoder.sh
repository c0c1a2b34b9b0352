# rocprofv3 kernel stats of the image encoder at 512x512 (tools/encoder_probe.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/enc_prof; mkdir -p gpurun_out/enc_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/enc_prof -- python3 tools/encoder_probe.py > gpurun_out/enc_prof/run.log 2>&1
tail -2 gpurun_out/enc_prof/run.log
f=$(find gpurun_out/enc_prof -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-230
