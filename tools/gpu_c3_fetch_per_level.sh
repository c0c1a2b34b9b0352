# configs[2]: HBM read / write bytes and L2 hits of EVERY segment launch of one frame (rocprofv3 --pmc, counters only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/c3_fetch; rm -rf $out; mkdir -p $out
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$out/$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --samples 128 --early-term --no-cpu-baseline --no-extras > $d.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
res = collections.OrderedDict()
for name in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    f = glob.glob(f"gpurun_out/c3_fetch/{name}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "render_fused_kernel" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)[-17:-1] if len(by) > 17 else sorted(by)      # the last two frames' 8 launches each (the probe launch follows)
    res[name] = [by[i] for i in ids]
n = len(res["FETCH_SIZE"])
print("launch  read MB  write MB  L2 hit")
for i in range(n):
    rd = res["FETCH_SIZE"][i].get("FETCH_SIZE", 0) * 1024 * 2 / 1e6          # KB; x2 on gfx950 (MI355X_MICROARCH.md, tools/pmc_derive.py)
    wr = res["WRITE_SIZE"][i].get("WRITE_SIZE", 0) * 1024 / 1e6
    h = res["TCC_HIT_sum"][i]; hit = h.get("TCC_HIT_sum", 0) / max(1.0, h.get("TCC_HIT_sum", 0) + h.get("TCC_MISS_sum", 0))
    print(f"{i % 8:5d}  {rd:8.1f}  {wr:8.1f}  {hit:6.3f}")
PY
