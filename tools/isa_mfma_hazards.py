"""Static check of one `hipcc -S` listing for the software-managed MFMA hazard of gfx940/gfx950: a VGPR written by an XDL
MFMA must not be read or written by a VALU / VMEM / LDS instruction until (passes + 2) wait states later (16-pass: LLVM pads to exactly 18) (LLVM
GCNHazardRecognizer::checkMAIVALUHazards, GFX940_XDL_N_PassWriteVgprVALU*WaitStates); the hardware does not interlock.
LLVM inserts the s_nops for instructions it knows -- but an `asm()` statement is opaque to it: a v_fma_mix* / v_cvt written as
inline assembly that consumes a fresh MFMA result gets no wait states.  The listing is walked linearly per kernel (fall-through
order, labels ignored: a branch target is checked as if entered from the instruction above it), one wait state per issued
instruction, `s_nop n` = n + 1.

The converse pair (round 6): a VGPR WRITTEN by a VALU instruction must not be read as an MFMA source operand (A, B or C) before
two wait states have passed -- gfx950 does not interlock that either (tools/micro/asm_producer_hazards.hip: a v_fma_mixhi_f16, a
v_permlane32_swap_b32, even a plain v_mov_b32 / v_add_f32 directly in front of the MFMA that reads its result hands the MFMA the
register's PREVIOUS contents; one wait state cures three of the four, v_add_f32 -> fp32 MFMA needs two).  LLVM pads the
producers it can see; the kernels' inline-asm producers (lo_pair's v_fma_mixlo/hi_f16, interleave16's v_permlane32_swap_b32) it
cannot.  This is the mechanism behind round 5's "split-precision deferred colour branch comes out 10-30 % wrong in one build":
the regathered inputs' lo halves were converted by asm statements the scheduler had placed directly in front of their MFMA
(DESIGN.md 4.1).  Checked here for every opaque producer: inside an `asm` block of a listing, or one of those opcodes in a
disassembled library.

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


OPAQUE_OPS = ("v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_permlane32_swap_b32")      # only ever emitted by inline asm in this code base
# wait states between an opaque VALU write and an MFMA read of it, by the consumer (measured, tools/micro/asm_producer_hazards.hip:
# v_fma_mixhi_f16 / v_mov_b32 -> v_mfma_f32_32x32x16_f16: 1 cures; v_add_f32 -> v_mfma_f32_32x32x2_f32: 2; v_permlane32_swap -> the
# same: 1 -- the fp32 MFMA is held to 2)
PRODUCER_WAIT = 2
def producer_wait(mfma_op):
    return 1 if "f16" in mfma_op.split("x")[-1] or "bf16" in mfma_op else 2


def check(rows, name):
    recent, bad, n_mfma, n_asm = [], [], 0, 0          # recent: [dst regs, wait states since issue, needed, text]
    producers = []                                     # opaque VALU writes: [dst regs, wait states since issue, text]
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
            producers = []
            continue
        if op == "s_nop":
            ws = int(t.split()[1], 0) + 1
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            n_mfma += 1
            dst = regs(t.split(",")[0])
            srcs = regs(",".join(t.split(None, 1)[1].split(",")[1:])) if " " in t else set()
            for pdst, el, txt in producers:
                if el < producer_wait(op) and pdst & srcs:
                    bad.append((ln, t, txt, el, producer_wait(op), "producer"))
            for r in recent:
                r[1] += 1
            for q in producers:
                q[1] += 1
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
            for q in producers:
                q[1] += ws
            if op.startswith("v_") and (in_asm or op.split("_e32")[0].split("_e64")[0] in OPAQUE_OPS) and " " in t:
                ops = t.split(None, 1)[1].split(",")
                # (v_permlane32_swap writes both of its operands)
                pdst = regs(ops[0]) | (regs(ops[1]) if op.startswith("v_permlane32_swap") and len(ops) > 1 else set())
                producers.append([pdst, 0, f"{ln}: {t}"])
        recent = [r for r in recent if r[1] < r[2]]
        producers = [q for q in producers if q[1] < PRODUCER_WAIT]
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
            if ia == "producer":
                print(f"    line {ln}: `{t}` reads the result of the inline-asm `{txt}` after {el} of {need} wait states")
            else:
                print(f"    line {ln}: `{t}`{' [inline asm]' if ia else ''} touches the result of `{txt}` after {el} of {need} wait states")
        total += len(bad)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
