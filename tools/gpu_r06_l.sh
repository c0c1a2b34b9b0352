cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6l; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_sparse_conv.py tests/test_gpu_renderer.py tests/test_gpu_multi.py -m gpu -q -s -x 2>&1 | grep -v "^$" | grep "vertex sets\|FAILED\|passed\|failed\|Error\|assert" | tail -12 | tee $o/gpu_tests.txt
