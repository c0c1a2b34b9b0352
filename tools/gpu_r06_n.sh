cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT/build/rc5"
o=$GRAFT_REPO_ROOT/gpurun_out/r6n; rm -rf $o; mkdir -p $o
for v in A B; do
  echo "== lib$v (round-5 revision c9a579c, -DGPNERF_X_SPLIT_DEFER, no opaque copy of the regathered inputs$( [ $v = B ] && echo '; s_nop 1 behind every v_fma_mixhi_f16 of lo_pair'))" | tee -a $o/split_defer_root_cause.txt
  GPNERF_DIAG_LIB=lib$v.so timeout 300 python tools/probes/defer_debug.py 2>&1 | grep "^split\|^guard" | tee -a $o/split_defer_root_cause.txt
done
