cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sva2; mkdir -p gpurun_out/sva2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sva2 -- python3 tools/time_survey_api.py 4 > gpurun_out/sva2/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/sva2/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if "render_fused_kernel<4" in r["Kernel_Name"]]
lo = max(i for i, r in enumerate(rows[:fused[-1]]) if "conv2d_nhwc_kernel<1, 1, 1" in r["Kernel_Name"])    # last encoder kernel of the last frame
t0 = int(rows[lo]["End_Timestamp"]); prev = t0
for r in rows[lo + 1:fused[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:58]
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {nm}")
    prev = max(prev, e)
PY
