cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6q; rm -rf $o; mkdir -p $o
timeout 900 python tools/defer_sweep.py 80 2>&1 | tail -2
timeout 600 python tools/parity_sweep.py 60 2>&1 | tail -2
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -3 | cut -c1-250 | tee $o/gpu_tests.txt
