#!/bin/bash
# Collect rocprofv3 PMC counters for the fused kernel in separate passes (never combined with tracing).
# usage: [PMC_KERNEL='render_fused_kernel<0, false, false, true>'] tools/pmc_passes.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/passN/, summary.json
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $counters --output-format csv -d $out/pass$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/pass$i.log 2>&1
  echo "pass $i ($counters): rc=$?"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
LIST
python3 tools/pmc_summary.py $out "${PMC_KERNEL:-render_fused}"
