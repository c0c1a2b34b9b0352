#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per-counter mean over the dispatches of one kernel.
usage: pmc_summary.py gpurun_out/pmc_<tag> [kernel-name substring, default "render_fused"] [output file name, default summary.json]
(bench.py launches several instantiations in one process -- the headline, the diagnostic step_stats launch of the same kernel, and
the dense GPNERF_FLAG_NO_EXITS launch `render_fused_kernel<0, false, false, false>` -- so the caller names the one it wants)"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "render_fused"
res = {}
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    for row in csv.DictReader(open(f)):
        if kernel not in row.get("Kernel_Name", ""):
            continue
        acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, per in acc.items():
        vals = list(per.values())
        res[name] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(root, sys.argv[3] if len(sys.argv) > 3 else "summary.json"), "w"), indent=1)
