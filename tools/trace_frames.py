#!/usr/bin/env python3
"""Timeline of one Renderer.render call from a rocprofv3 kernel trace (diagnostic).
usage: trace_frames.py kernel_trace.csv [n-th fused-kernel dispatch to show, default: last of the first leg]
Splits the trace at render_fused_kernel dispatches; for the chosen frame prints every kernel between the previous fused
kernel's end and this one's end: start offset, duration, gap since the previous kernel's end."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if "render_fused_kernel" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lo, hi = fused[which - 1] + 1, fused[which]
t0 = int(rows[lo - 1]["End_Timestamp"])
prev = t0
busy = gaps = 0
agg = {}
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {name}")
    busy += e - s
    gaps += max(0, s - prev)
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    prev = max(prev, e)
print(f"frame span {(prev - t0) / 1e3:.1f} us: kernels {busy / 1e3:.1f} us, idle gaps {gaps / 1e3:.1f} us, {hi - lo + 1} launches")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {d:9.1f} us  x{c:3d}  {n}")
