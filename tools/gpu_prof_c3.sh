# rocprofv3 kernel trace of the configs[2] bench: durations of the per-segment launches of one frame
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/c3_prof; mkdir -p gpurun_out/c3_prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c3_prof -- python3 bench.py --steps 4 --warmup 2 ${ARGS:---samples 128 --early-term} --no-cpu-baseline --no-extras > gpurun_out/c3_prof/bench.json 2> gpurun_out/c3_prof/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c3_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "render_fused_kernel" in r["Kernel_Name"]]
import os
n = int(os.environ.get("NSEG", "8"))
last = ks[-2 * n - 1:-1] if len(ks) > 2 * n else ks     # the last two frames' launches (the samples_done probe's follow: dropped with its last)
t0 = int(last[0]["Start_Timestamp"])
prev = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f} us")
    prev = e
print("span", (prev - t0) / 1e3, "us")
PY
