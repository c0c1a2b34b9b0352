# kernel trace of the headline bench with the folded volumes: fold kernels and the fused kernel of the last frames
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fold_prof; mkdir -p gpurun_out/fold_prof
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fold_prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${ARGS:-} > gpurun_out/fold_prof/bench.json 2> gpurun_out/fold_prof/err.txt
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/fold_prof/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
i = max(j for j, r in enumerate(rows) if "render_fused_kernel" in r["Kernel_Name"])
agg = collections.defaultdict(list)
for r in rows[max(0, i - 24): i + 1]:
    agg[r["Kernel_Name"][:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items(): print(f"{sum(v) / len(v):9.1f} us x{len(v):3d} (min {min(v):8.1f} max {max(v):8.1f})  {k}")
PY
