# code objects built for gfx950:xnack- (no page-fault replay: loads may overwrite their own address registers) against the default
# (xnack "any"): register counts, spills and time of the fused kernel and the encoder (builds on the GPU box)
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
FL="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
for f in kernels volume conv; do hipcc $FL --offload-arch=gfx950:xnack- -Rpass-analysis=kernel-resource-usage -c -o /tmp/ab/x_$f.o $C/gpnerf_$f.hip 2> /tmp/ab/x_$f.err & done; wait
hipcc -shared -fPIC --offload-arch=gfx950:xnack- -o /tmp/ab/lib_xnackoff.so /tmp/ab/x_kernels.o /tmp/ab/x_volume.o /tmp/ab/x_conv.o || { tail -3 /tmp/ab/x_kernels.err; exit 1; }
grep -A12 "render_fused_kernelILi4ELb0ELb0E" /tmp/ab/x_kernels.err | grep -E "VGPRs:|Spill|ScratchSize" | head -4
run() { for args in "" "--fill survey" "--samples 128 --early-term"; do
  GPNERF_DEBUG=1 GPNERF_LIB_PATH=$2 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras $args 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 | $args |', round(j['ms_per_step'],3), 'ms')"; done
  GPNERF_DEBUG=1 GPNERF_LIB_PATH=$2 python tools/probes/encoder_time.py 2>&1 | tail -1 | cut -c1-42; }
run default $PWD/$C/libgpnerf_hip.so; run xnack- /tmp/ab/lib_xnackoff.so; run default $PWD/$C/libgpnerf_hip.so; run xnack- /tmp/ab/lib_xnackoff.so
