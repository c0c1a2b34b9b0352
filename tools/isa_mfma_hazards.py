"""Static check of one `hipcc -S` listing for the software-managed MFMA hazard of gfx940/gfx950: a VGPR written by an XDL
MFMA must not be read or written by a VALU / VMEM / LDS instruction until (passes + 2) wait states later (16-pass: LLVM pads to exactly 18) (LLVM
GCNHazardRecognizer::checkMAIVALUHazards, GFX940_XDL_N_PassWriteVgprVALU*WaitStates); the hardware does not interlock.
LLVM inserts the s_nops for instructions it knows -- but an `asm()` statement is opaque to it: a v_fma_mix* / v_cvt written as
inline assembly that consumes a fresh MFMA result gets no wait states.  The listing is walked linearly per kernel (fall-through
order, labels ignored: a branch target is checked as if entered from the instruction above it), one wait state per issued
instruction, `s_nop n` = n + 1.

usage: isa_mfma_hazards.py file.s|lib.so [kernel substring]      exit code 1 if a violation is found
A .so is taken apart with llvm-objdump (--offloading, then -d on every gfx950 code object); tests/test_abi.py runs this on the
built library, so an inline-asm consumer that lands inside an MFMA's shadow fails the CPU suite."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

PASSES = {"v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_32x32x2f32": 16, "v_mfma_f32_32x32x16_f16": 10, "v_mfma_f32_32x32x8_f16": 16,
          "v_mfma_f32_16x16x4_f32": 8, "v_mfma_f32_16x16x32_f16": 4, "v_mfma_f32_32x32x16_bf16": 10}


def regs(tok):
    """VGPRs as ints, AGPRs as 1000 + n"""
    out = set()
    for k, a, b in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out.update(range(int(a) + (1000 if k == "a" else 0), int(b) + 1 + (1000 if k == "a" else 0)))
    for k, a in re.findall(r"\b([va])(\d+)\b", tok):
        out.add(int(a) + (1000 if k == "a" else 0))
    return out


def check(rows, name):
    recent, bad, n_mfma, n_asm = [], [], 0, 0          # recent: [dst regs, wait states since issue, needed, text]
    in_asm = False
    for ln, l in rows:
        t = l.split("//")[0].split(";")[0].strip()
        if "#ASMSTART" in l or "#ASMEND" in l:
            in_asm = "#ASMSTART" in l
            continue
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        ws = 1
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            # nothing falls through an unconditional jump: what follows is another block's start, reached only by jumps (the walk
            # is linear; the inline-asm consumers this check is for sit in straight-line code behind their MFMAs)
            recent = []
            continue
        if op == "s_nop":
            ws = int(t.split()[1], 0) + 1
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            n_mfma += 1
            dst = regs(t.split(",")[0])
            for r in recent:
                r[1] += 1
            recent.append([dst, 0, PASSES.get(op, 16) + 2, f"{ln}: {t}"])
        else:
            touched = regs(t.split(None, 1)[1]) if " " in t else set()
            is_consumer = op.startswith(("v_", "global_", "ds_", "buffer_", "flat_", "scratch_")) and not op.startswith("v_mfma")
            if is_consumer and touched:
                for dst, el, need, txt in recent:
                    # this instruction issues `el` wait states after the MFMA (el counts the instructions / nops in between)
                    if el < need and dst & touched:
                        bad.append((ln, t, txt, el, need, in_asm))
            n_asm += in_asm
            for r in recent:
                r[1] += ws
        recent = [r for r in recent if r[1] < r[2]]
    return bad, n_mfma, n_asm


def listing_of(path):
    """rows of an assembly listing; a shared library / object is disassembled first"""
    if path.endswith((".s", ".S", ".asm")):
        return open(path).read().split("\n")
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    tmp = tempfile.mkdtemp(prefix="gpnerf_isa_")
    try:
        lib = os.path.join(tmp, "x.so")
        shutil.copy(path, lib)
        subprocess.run([objdump, "--offloading", lib], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        rows = []
        for co in sorted(glob.glob(lib + ".*gfx950*")):
            rows += subprocess.run([objdump, "-d", co], capture_output=True, text=True, check=True).stdout.split("\n")
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scan(path, want=""):
    """[(kernel, n_mfma, violations)] for every kernel with MFMAs whose name contains `want`"""
    rows = listing_of(path)
    hdr = re.compile(r"^(?:[0-9a-f]{8,16} <(_Z\w+)>:|(_Z\w*):)")
    starts = [(i, (m.group(1) or m.group(2))) for i, l in enumerate(rows) for m in [hdr.match(l)] if m]
    res = []
    for j, (s, name) in enumerate(starts):
        if want not in name:
            continue
        e = starts[j + 1][0] if j + 1 < len(starts) else len(rows)
        e = next((i + 1 for i in range(s, e) if "s_endpgm" in rows[i]), e)
        bad, n_mfma, n_asm = check([(i + 1, rows[i]) for i in range(s + 1, e)], name)
        if n_mfma:
            res.append((name, n_mfma, n_asm, bad))
    return res


def main():
    total = 0
    for name, n_mfma, n_asm, bad in scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        print(f"{name[:110]}: {n_mfma} MFMAs, {n_asm} inline-asm instructions, {len(bad)} hazard(s)")
        for ln, t, txt, el, need, ia in bad[:12]:
            print(f"    line {ln}: `{t}`{' [inline asm]' if ia else ''} touches the result of `{txt}` after {el} of {need} wait states")
        total += len(bad)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
