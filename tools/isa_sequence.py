"""Compressed instruction-class sequence of one kernel in `hipcc -S` output (diagnostic): L = global load, Wn = s_waitcnt
vmcnt(n), Mn = run of n MFMAs, vn = n VALU instructions, | = label.  usage: isa_sequence.py file.s <kernel substring> [max chars]"""
import re
import sys

rows = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(rows) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l.split(":")[0])
end = next(i for i in range(start, len(rows)) if "s_endpgm" in rows[i])
out, v, m = [], 0, 0


def flush():
    global v, m
    if v:
        out.append(f"v{v}")
    if m:
        out.append(f"M{m}")
    v = m = 0


for l in rows[start:end]:
    t = l.strip()
    if not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    if op.startswith("global_load"):
        flush(); out.append("L")
    elif op == "s_waitcnt" and "vmcnt" in t:
        flush(); out.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
    elif op.startswith("v_mfma"):
        if v:
            flush()
        m += 1
    elif op.startswith("v_"):
        if m:
            flush()
        v += 1
    elif t.split(";")[0].strip().endswith(":"):
        flush(); out.append("|")
flush()
s = " ".join(out)
s = re.sub(r"(?:L )+L", lambda mo: f"L*{mo.group(0).count('L')}", s)
print(s[: int(sys.argv[3]) if len(sys.argv) > 3 else 8000])
