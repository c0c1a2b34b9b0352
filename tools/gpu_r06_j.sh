cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6j; rm -rf $o; mkdir -p $o
timeout 300 python tools/probes/exact_encoder_time.py 2>&1 | grep precision | tee $o/encoder_forms.txt
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_encoder.py tests/test_gpu_renderer.py -m gpu -q -s 2>&1 | grep -v "^$" | grep "e2e\|FAILED\|passed\|failed\|Error\|assert\|one-hot" | tail -20 | tee $o/gpu_tests.txt
