"""Timeline of the pipelined evaluation loop from a rocprofv3 kernel trace (tools/probes/eval_loop_time.py under
`rocprofv3 --kernel-trace`): for the LAST loop of the run (the pipelined one), per frame: when the per-ray kernel runs, on which
queue, and when the NEXT frame's producers (encoder graph, volume builder, frame glue: everything on the other queue) run relative
to it.  usage: pipeline_timeline.py kernel_trace.csv [n_frames]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
fused = [r for r in rows if "render_fused_kernel" in r["Kernel_Name"]]
last = fused[-nf:]
t0 = last[0]["s"]
print(f"{len(fused)} per-ray launches in the trace; the last {nf} (pipelined loop, steady state).  Times in ms from the first one's start.")
for i, k in enumerate(last):
    end_prev = last[i - 1]["e"] if i else None
    nxt = last[i + 1]["s"] if i + 1 < len(last) else None
    print(f"\nframe {i}: per-ray kernel on queue {k['Queue_Id']}: {(k['s'] - t0) / 1e6:8.3f} .. {(k['e'] - t0) / 1e6:8.3f}  ({(k['e'] - k['s']) / 1e6:.3f} ms)"
          + (f"   idle since the previous per-ray kernel: {(k['s'] - end_prev) / 1e6:.3f} ms" if end_prev else ""))
    if nxt is None:
        continue
    between = [r for r in rows if r["s"] >= k["s"] and r["s"] < nxt and r is not k]
    byq = {}
    for r in between:
        byq.setdefault(r["Queue_Id"], []).append(r)
    for q, rs in sorted(byq.items()):
        busy = sum(r["e"] - r["s"] for r in rs) / 1e6
        inside = [r for r in rs if r["s"] < k["e"]]
        names = {}
        for r in rs:
            n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
            names[n] = names.get(n, 0) + 1
        top = ", ".join(f"{n} x{c}" for n, c in sorted(names.items(), key=lambda x: -x[1])[:4])
        print(f"   queue {q}: {len(rs):3d} launches, {busy:.3f} ms busy, first at {(rs[0]['s'] - t0) / 1e6:8.3f}, last ends {(max(r['e'] for r in rs) - t0) / 1e6:8.3f}; "
              f"{len(inside)} started while the per-ray kernel ran  [{top}]")
