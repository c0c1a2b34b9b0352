"""Where a kernel's register spills sit (diagnostic): v_readlane / v_writelane (SGPR spills) and scratch_ (VGPR spills) per loop
nesting depth, from `hipcc -S --cuda-device-only` output.  usage: isa_spills.py file.s <substring of the kernel's symbol>"""
import collections
import re
import sys

rows = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(rows) if re.match(r"^_Z\w*:", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(rows)) if rows[i].startswith(".Lfunc_end"))
depth, c, n_by_depth = 0, collections.Counter(), collections.Counter()
for l in rows[start:end]:
    m = re.search(r"Depth=(\d+)", l)
    if re.match(r"^\.LBB\d+_\d+:", l):
        depth = int(m.group(1)) if m else 0
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    n_by_depth[depth] += 1
    if op.startswith(("v_readlane", "v_writelane", "scratch_")):
        c[(depth, op.split("_b32")[0])] += 1
print("instructions per loop depth:", dict(sorted(n_by_depth.items())))
for (d, op), n in sorted(c.items()):
    print(f"  depth {d}: {n:4d} {op}")
