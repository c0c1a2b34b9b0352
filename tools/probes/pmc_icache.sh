cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in defer nodefer; do
  if [ $tag = nodefer ]; then export GPNERF_DEBUG=1 GPNERF_DEFER=0; fi
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_ic_$tag/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d gpurun_out/pmc_ic_$tag/p2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_ic_$tag > /dev/null 2>&1
  python3 - <<PY
import json
j=json.load(open("gpurun_out/pmc_ic_$tag/summary.json"))
print("$tag", {k: round(v["mean_per_dispatch"]) for k,v in j.items() if isinstance(v, dict) and "mean_per_dispatch" in v})
PY
done
