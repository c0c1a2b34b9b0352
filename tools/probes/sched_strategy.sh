# LLVM AMDGPU scheduler strategies on gpnerf_kernels.hip (builds on the GPU box), headline + survey + configs[2] per build
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
FL="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function"
build() { # name, extra flags
  hipcc $FL $2 -c -o /tmp/ab/k_$1.o $C/gpnerf_kernels.hip 2> /tmp/ab/k_$1.err &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/ab/lib_$1.so /tmp/ab/k_$1.o $C/gpnerf_volume.o $C/gpnerf_conv.o || echo "build $1 failed: $(tail -2 /tmp/ab/k_$1.err)"; }
build base "" & build maxilp "-mllvm -amdgpu-sched-strategy=max-ilp" & build maxmem "-mllvm -amdgpu-sched-strategy=max-memory-clause" &
build minreg "-mllvm -amdgpu-sched-strategy=iterative-minreg" & build itilp "-mllvm -amdgpu-sched-strategy=iterative-ilp" & wait
run() { [ -f /tmp/ab/lib_$1.so ] || return; for args in "" "--fill survey" "--samples 128 --early-term"; do
  GPNERF_DEBUG=1 GPNERF_LIB_PATH=/tmp/ab/lib_$1.so python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras $args 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 | $args |', round(j['ms_per_step'],3), 'ms')"; done; }
for v in base maxilp maxmem minreg itilp base; do run $v; done
