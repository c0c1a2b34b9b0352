# kernel trace of the 73 689-ray frame (bench --fill survey) with the sample-split geometry off: bulk launch + remainder launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sv_prof; mkdir -p gpurun_out/sv_prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sv_prof -- python3 bench.py --steps 3 --warmup 1 --fill survey --no-cpu-baseline --no-extras > gpurun_out/sv_prof/bench.json 2> gpurun_out/sv_prof/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/sv_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "render_fused_kernel" in r["Kernel_Name"]][-6:]
t0 = int(ks[0]["Start_Timestamp"]); prev = t0
for r in ks:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][30:70]} grid {r['Grid_Size_X']} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f} us")
    prev = e
PY
