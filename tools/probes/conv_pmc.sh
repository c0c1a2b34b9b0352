# rocprofv3 PMC counters of the 3x3 convolution kernel (separate passes, no tracing), summarised per wave
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_conv; rm -rf $out; mkdir -p $out
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $counters --output-format csv -d $out/pass$i -- python3 tools/probes/conv_pmc_target.py > $out/pass$i.log 2>&1
  echo "pass $i ($counters): rc=$?"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16
LIST
python3 - <<'PY'
import csv, glob, collections
tot = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_conv/pass*/**/*counter_collection.csv", recursive=True)):
    n = collections.Counter(); s = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "conv3x3" not in r["Kernel_Name"]: continue
        s[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in s: tot[k] = s[k] / n[k]
waves = 192 * 4 * (2 if tot.get("SQ_WAVES", 768) > 1000 else 1)
print("per dispatch mean; per-wave = value / SQ_WAVES")
for k, v in tot.items(): print(f"{k:34s} {v:16.0f}  per wave {v / tot.get('SQ_WAVES', waves):12.1f}")
PY
