# rocprofv3 kernel stats of the image encoder at 3x512x512 (tools/probes/encoder_graph_only.py: graph replays, so the gaps are the device's, not the host's)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/enc_prof; mkdir -p gpurun_out/enc_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/enc_prof -- python3 tools/probes/encoder_graph_only.py > gpurun_out/enc_prof/run.log 2>&1
tail -1 gpurun_out/enc_prof/run.log
f=$(find gpurun_out/enc_prof -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-200
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/enc_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last call of the encoder: from the last conv2d<7,2,..> (stem) to the end
stems = [i for i, r in enumerate(rows) if "conv7x7_s2_stem_kernel" in r["Kernel_Name"]]
lo = stems[-1]
print("last call: per-launch list (name, grid, dur us, gap us)")
prev = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print(f"  {nm:44s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{int(r['Grid_Size_Y']):2d}x{int(r['Grid_Size_Z']):2d}  {(e-s)/1e3:7.1f}  gap {(s-prev)/1e3:6.1f}")
    prev = e
PY
