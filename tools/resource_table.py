#!/usr/bin/env python3
"""Register / spill table of every kernel of the library, plus where the spill instructions of the fused kernel's instantiations
sit (inside or outside the sample loops).  Compiles the three sources with -Rpass-analysis=kernel-resource-usage and -S
(~2 min); writes profiles/<round>/h_kernel_resources.md.   usage: resource_table.py profiles/r03"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gp-nerf_amd", "csrc")
FLAGS = ["-I" + os.path.join(CSRC, "nodiag"), "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function"]


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("void ", "")


def resources(src, tmp):
    err = subprocess.run(["hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.path.join(tmp, "x.o"), src],
                         capture_output=True, text=True).stderr
    out, cur = [], None
    for l in err.split("\n"):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r" AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("sspill", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, l)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    return out


def loops(asm_rows, key):
    """per innermost (depth >= 2) loop of one kernel: instructions, MFMAs, scratch loads / stores, v_readlane / v_writelane"""
    start = next(i for i, l in enumerate(asm_rows) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(asm_rows)) if asm_rows[i].startswith(".Lfunc_end"))
    stats, depth, hdr = collections.OrderedDict(), 0, None
    outside = collections.Counter()
    for l in asm_rows[start:end]:
        m2 = re.match(r"^\.L(BB\d+_\d+):.*Loop Header: Depth=(\d+)", l)
        m = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", l)
        if m2:
            depth, hdr = int(m2.group(2)), m2.group(1)
        elif m:
            depth, hdr = int(m.group(2)), m.group(1)
        elif (re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb.")) and "Depth" not in l:
            depth, hdr = 0, None
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        c = stats.setdefault(hdr, collections.Counter()) if depth >= 2 else outside
        c["n"] += 1
        for k, pre in (("mfma", "v_mfma"), ("scratch_load", "scratch_load"), ("scratch_store", "scratch_store"), ("v_readlane", "v_readlane"),
                       ("v_writelane", "v_writelane")):
            if op.startswith(pre):
                c[k] += 1
    return {k: v for k, v in stats.items() if v["mfma"] >= 100}, outside


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03")
    os.makedirs(dst, exist_ok=True)
    lines = ["# Kernel resources (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) -- `python tools/resource_table.py`", "",
             "| kernel | VGPR | AGPR | VGPR spills | SGPR spills | scratch B/lane | waves/SIMD |", "|---|---|---|---|---|---|---|"]
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("gpnerf_kernels.hip", "gpnerf_conv.hip", "gpnerf_volume.hip"):
            for c in resources(os.path.join(CSRC, f), tmp):
                lines.append(f"| `{demangle(c['name'])[:100]}` | {c.get('vgpr')} | {c.get('agpr', 0)} | {c.get('vspill')} | {c.get('sspill')} | {c.get('scratch')} | {c.get('occ')} |")
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "gpnerf_kernels.hip")], capture_output=True)
        rows = open(asm).read().split("\n")
        lines += ["", "## Spill instructions inside the sample loops of `render_fused_kernel` (from `hipcc -S`)", "",
                  "A spilled VGPR comes back as `scratch_load`, a spilled SGPR as `v_readlane`.  Per sample loop (the chained form has one copy of the loop",
                  "per samples-per-step variant P = 1, 2, 4, 8): instructions of one iteration, MFMAs, and the spill instructions among them; `outside`: the rest of the kernel.", "",
                  "| instantiation | loop | instructions | MFMAs | scratch_load | scratch_store | v_readlane | v_writelane |", "|---|---|---|---|---|---|---|---|"]
        for inst, key in (("<FORM_F32_FOLD, plain> (headline)", "render_fused_kernelILi4ELb0ELb0E"), ("<FORM_F32, plain>", "render_fused_kernelILi0ELb0ELb0E"),
                          ("<FORM_F32_FOLD, chained> (configs[2])", "render_fused_kernelILi4ELb1ELb0E"), ("<FORM_F32_FOLD, culled>", "render_fused_kernelILi4ELb0ELb1E"),
                          ("<FORM_SPLIT_GUARD, plain>", "render_fused_kernelILi2ELb0ELb0E")):
            st, outside = loops(rows, key)
            for h, c in st.items():
                lines.append(f"| {inst} | {h} | {c['n']} | {c['mfma']} | {c['scratch_load']} | {c['scratch_store']} | {c['v_readlane']} | {c['v_writelane']} |")
            lines.append(f"| {inst} | outside | {outside['n']} | {outside['mfma']} | {outside['scratch_load']} | {outside['scratch_store']} | {outside['v_readlane']} | {outside['v_writelane']} |")
    open(os.path.join(dst, "h_kernel_resources.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-14:]))


if __name__ == "__main__":
    main()
