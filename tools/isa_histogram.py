"""Instruction histogram of one kernel from `hipcc -S --cuda-device-only` output (diagnostic).
usage: isa_histogram.py file.s <substring of the kernel's symbol> [first_line last_line]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
rows = open(path).read().split("\n")
start = next(i for i, l in enumerate(rows) if re.match(r"^_Z\w*:", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(rows)) if "s_endpgm" in rows[i])
if len(sys.argv) > 4 and int(sys.argv[4]) > 0:
    start, end = int(sys.argv[3]), int(sys.argv[4])
c = collections.Counter()
for l in rows[start:end]:
    l = l.strip()
    if not l or l.startswith((";", ".")) or l.split(";")[0].strip().endswith(":"):
        continue
    c[l.split()[0]] += 1


def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")): return "trans"
    if op.startswith(("v_accvgpr", "v_mov")): return "vmov"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    return "other"


cats = collections.Counter()
for op, n in c.items():
    cats[cat(op)] += n
print(f"lines {start}-{end}: {sum(c.values())} instructions")
print(dict(cats.most_common()))
for op, n in c.most_common(int(sys.argv[5]) if len(sys.argv) > 5 else 40):
    print(f"{n:6d} {op}")
