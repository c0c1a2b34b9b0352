#!/bin/bash
# rocprofv3 PMC passes on the vector-memory path of the fused kernel (TA / TCP / SQ wait counters); counters only, no tracing.
# Few counters per pass and a timeout on every pass: a pass that asks for more than the hardware can collect aborts and then
# hangs in rocprofv3's finalisation.
# usage: tools/pmc_gather.sh <tag> [bench args...]   -> gpurun_out/pmcg_<tag>/
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcg_$tag
mkdir -p $out
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $counters --output-format csv -d $out/pass$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/pass$i.log 2>&1
  echo "pass $i ($counters): rc=$?"
done <<'LIST'
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD
LIST
python3 tools/pmc_summary.py $out "render_fused_kernel<false," > /dev/null
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d.items(): print(f"{k:40s} {v['mean_per_dispatch']:.4g}")
PY
