cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6t; rm -rf $o; mkdir -p $o
timeout 300 python tools/wave_times.py 512 64 2>&1 | grep -v amdgpu | tee $o/wave_times_512.txt
timeout 300 python tools/stamps.py 512 2>&1 | grep -v amdgpu | tee $o/stamps_512.txt
