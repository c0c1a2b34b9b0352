"""MFMA run structure of one kernel in `hipcc -S` output: how often VALU instructions interrupt the MFMA stream (diagnostic).
usage: isa_mfma_runs.py file.s <substring of the kernel symbol>"""
import collections
import re
import sys

rows = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(rows) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l.split(":")[0])
end = next(i for i in range(start, len(rows)) if "s_endpgm" in rows[i])
seq = []
for l in rows[start:end]:
    l = l.strip()
    if not l or l.startswith((";", ".")) or l.split(";")[0].strip().endswith(":"):
        continue
    seq.append(l.split()[0])
cls = lambda op: "M" if op.startswith("v_mfma") else ("V" if op.startswith("v_") else "o")
c = "".join(cls(o) for o in seq)
runs = re.findall(r"M(?:o*M)*", c)
print("MFMA runs:", len(runs), sorted(collections.Counter(len(re.findall("M", r)) for r in runs).items()))
idx = [i for i, ch in enumerate(c) if ch == "M"]
ops, short = collections.Counter(), 0
for a, b in zip(idx[:-1], idx[1:]):
    v = [o for o in seq[a + 1:b] if o.startswith("v_")]
    if 0 < len(v) <= 8:
        short += 1
        ops.update(v)
print("VALU gaps of <= 8 instructions between two MFMAs:", short, ops.most_common(10))
print(dict(collections.Counter(cls(o) for o in seq)))
