#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per-counter mean over the dispatches of one kernel, or per CALL over the kernels of one call.
usage: pmc_summary.py gpurun_out/pmc_<tag> [kernel-name substring[+substring...], default "render_fused"] [output file name, default summary.json]
(bench.py launches several instantiations in one process -- the headline, the diagnostic step_stats launch of the same kernels, and
the dense GPNERF_FLAG_NO_EXITS launch `render_fused_kernel<0, false, false, false, false>` -- so the caller names the one it wants.
Since round 6 one gpnerf_render_fused call of the default path is several kernels -- the sample loop `render_fused_kernel<0, false,
false, true, true>` (one launch per sample segment in the chained form), `colour_units_kernel<0>` and `colour_accumulate_kernel`.
"a+b+c" gives counters per CALL: every dispatch of the named kernels added up, divided by the dispatches of the LAST name, which
runs once per call; each part's own per-dispatch means are kept under "parts")"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
kernels = (sys.argv[2] if len(sys.argv) > 2 else "render_fused").split("+")
res, parts = {}, {k: {} for k in kernels}
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per_kernel = {}
    for kernel in kernels:
        acc = {}
        for row in rows:
            if kernel not in row.get("Kernel_Name", ""):
                continue
            acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        per_kernel[kernel] = acc
        for name, per in acc.items():
            vals = list(per.values())
            parts[kernel][name] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
    for name in per_kernel[kernels[-1]]:
        calls = len(per_kernel[kernels[-1]][name])
        total = sum(sum(per_kernel[k].get(name, {}).values()) for k in kernels)
        res[name] = {"mean_per_dispatch": total / calls, "dispatches": calls}
print(json.dumps(res, indent=1))
out = dict(res)
if len(kernels) > 1:
    out["parts"] = parts
json.dump(out, open(os.path.join(root, sys.argv[3] if len(sys.argv) > 3 else "summary.json"), "w"), indent=1)
