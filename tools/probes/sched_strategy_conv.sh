# LLVM AMDGPU scheduler strategies on gpnerf_conv.hip / gpnerf_volume.hip (builds on the GPU box): encoder time, frame phase
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
FL="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function"
build() { # name, extra flags
  ( hipcc $FL $2 -c -o /tmp/ab/c_$1.o $C/gpnerf_conv.hip 2> /tmp/ab/c_$1.err && hipcc $FL $2 -c -o /tmp/ab/v_$1.o $C/gpnerf_volume.hip 2>> /tmp/ab/c_$1.err &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/ab/libc_$1.so $C/gpnerf_kernels.o /tmp/ab/v_$1.o /tmp/ab/c_$1.o ) || echo "build $1 failed: $(tail -2 /tmp/ab/c_$1.err)"; }
build base "" & build maxilp "-mllvm -amdgpu-sched-strategy=max-ilp" & build maxmem "-mllvm -amdgpu-sched-strategy=max-memory-clause" &
build minreg "-mllvm -amdgpu-sched-strategy=iterative-minreg" & wait
run() { [ -f /tmp/ab/libc_$1.so ] || return; echo "== $1"; for i in 1 2; do GPNERF_DEBUG=1 GPNERF_LIB_PATH=/tmp/ab/libc_$1.so python tools/probes/encoder_time.py 2>&1 | tail -1 | cut -c1-42; done
  GPNERF_DEBUG=1 GPNERF_LIB_PATH=/tmp/ab/libc_$1.so python tools/probes/render_phases.py 2>&1 | grep "frame:" | tr '\n' ' '; echo; }
for v in base maxilp maxmem minreg base; do run $v; done
