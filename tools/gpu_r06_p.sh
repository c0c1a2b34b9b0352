cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6p; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o/trace -- python3 tools/probes/exact_encoder_time.py > $o/run.txt 2>&1
f=$(find $o/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $o/encoder_fp32_graph_timeline.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the LAST fp32 graph replay: find the last run of kernels that starts with a stem EXACT kernel (ILi2ELb1E) before the split section
names = [r["Kernel_Name"] for r in rows]
stems = [i for i, n in enumerate(names) if "conv7x7_s2_stem_kernel<2, true>" in n]
i0 = stems[-1]
i1 = next(i for i in range(i0 + 1, len(rows)) if "conv7x7" in names[i] or i == len(rows) - 1)
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
tot = {}
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:58]
    g = r.get("Grid_Size", "?"); 
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {n}  grid {r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}x{r.get('Grid_Size_Z','')} wg {r.get('Workgroup_Size_X','')}")
    a = tot.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    prev = max(prev, e)
print(f"span {(prev - t0) / 1e3:.1f} us, kernels {sum(v[1] for v in tot.values()):.1f} us, {i1 - i0} launches")
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {d:8.1f} us x{c:3d}  {n}")
PY
rm -rf $o/trace
