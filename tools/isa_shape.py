"""Instruction-class shape of one kernel in a .s file (hipcc -S --cuda-device-only): M mfma, r/w ds_read/ds_write, G/S global load/store,
v VALU, s SALU, n s_nop, |B| barrier, [..] s_waitcnt.  usage: isa_shape.py file.s mangled-name-substring [first-label last-label]"""
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
start = [i for i, l in enumerate(s.split("\n")) if l.startswith("_Z") and key in l.split(":")[0]]
lines = s.split("\n")
i = start[0]
out = []
for l in lines[i + 1:]:
    l = l.strip()
    if l.startswith(".Lfunc_end"): break
    if not l or l.startswith(";"): continue
    if l.startswith(".LBB"): out.append("\n" + l.split(":")[0] + ":"); continue
    if l.startswith("."): continue
    op = l.split()[0]
    if "mfma" in op: out.append("M")
    elif op.startswith("ds_read"): out.append("r")
    elif op.startswith("ds_write"): out.append("w")
    elif op.startswith(("global_load", "buffer_load")): out.append("G")
    elif op.startswith(("global_store", "buffer_store")): out.append("S")
    elif op.startswith("s_waitcnt"): out.append("[" + l.split(None, 1)[1].replace(" ", "") + "]")
    elif op.startswith("s_barrier"): out.append("|B|")
    elif op.startswith("s_nop"): out.append("n")
    elif op.startswith("v_"): out.append("v")
    elif op.startswith(("s_cbranch", "s_branch")): out.append("J")
    else: out.append("s")
txt = "".join(out)
if len(sys.argv) > 3:
    a = txt.index(sys.argv[3] + ":"); b = txt.index(sys.argv[4] + ":") if len(sys.argv) > 4 else len(txt)
    txt = txt[a:b]
print(txt)
