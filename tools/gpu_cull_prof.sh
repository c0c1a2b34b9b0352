# kernel trace of the culled frame (bench --occ-cull --occupancy $OCC, light outputs) under two builds of the library:
source tools/diag_env.sh   # the lab library: launcher experiment knobs exist only there (csrc/diag/)
# per-kernel mean durations of the last frames (mask pre-pass, tile order, fused kernel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OCC=${OCC:-1.0}
for l in ${LIBS:-preP cullmask}; do
rm -rf gpurun_out/cull_$l; mkdir -p gpurun_out/cull_$l
GPNERF_LIB_PATH=$PWD/build/ab/$l.so timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cull_$l -- python3 bench.py --steps 5 --warmup 2 --occ-cull --occupancy $OCC --outputs light --no-cpu-baseline --no-extras > gpurun_out/cull_$l/bench.json 2> gpurun_out/cull_$l/err.txt
echo "== $l rc $?"
python3 - $l <<'PY'
import csv, glob, sys, collections
fs = glob.glob(f"gpurun_out/cull_{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)
if not fs: print("no trace"); sys.exit(0)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
i = max(j for j, r in enumerate(rows) if "render_fused_kernel" in r["Kernel_Name"])
agg = collections.defaultdict(list)
for r in rows[max(0, i - 40): i + 1]:
    agg[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items(): print(f"{sum(v) / len(v):9.1f} us x{len(v):3d}  {k}")
PY
done
