#!/bin/bash
# isa_diff.sh a.so b.so : do two builds of the library carry the same device code?  Disassembles every gfx950 code object of
# both (llvm-objdump --offloading, -d), strips addresses, and diffs per kernel.  Prints the kernels that differ (none = identical ISA).
set -e
OBJDUMP=${OBJDUMP:-/opt/rocm/lib/llvm/bin/llvm-objdump}
dis() {
  d=$(mktemp -d); cp "$1" $d/lib.so; (cd $d && $OBJDUMP --offloading lib.so >/dev/null 2>&1 || true)
  for co in $d/lib.so*gfx950*; do $OBJDUMP -d "$co"; done | sed -E 's/^\s*//; s/\s*\/\/ [0-9A-Fa-f]+:.*$//; s/<[^>]*\+0x[0-9a-f]+>//' > $2
  rm -rf $d
}
dis "$1" /tmp/isa_a.txt; dis "$2" /tmp/isa_b.txt
python3 - <<'PY'
import re
def kernels(p):
    out, cur = {}, None
    for l in open(p):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m: cur = m.group(1); out[cur] = []
        elif cur is not None and l.strip(): out[cur].append(re.sub(r"^[0-9a-f]+:?\s*", "", l.strip()))
    return out
a, b = kernels("/tmp/isa_a.txt"), kernels("/tmp/isa_b.txt")
diff = [k for k in sorted(set(a) | set(b)) if a.get(k) != b.get(k)]
print(f"{len(a)} / {len(b)} kernels; differing: {len(diff)}")
for k in diff[:40]:
    print("  ", k[:150], len(a.get(k, [])), len(b.get(k, [])))
PY
