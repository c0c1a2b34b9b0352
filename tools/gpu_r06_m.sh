cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6m; rm -rf $o; mkdir -p $o
(cd tools/micro && hipcc -O3 --offload-arch=gfx950 -o /tmp/aph asm_producer_hazards.hip 2>/dev/null && timeout 60 /tmp/aph) 2>&1 | tee $o/asm_producer_hazards.txt
