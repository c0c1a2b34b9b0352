"""Static check of one `hipcc -S` listing for the software-managed MFMA hazard of gfx940/gfx950: a VGPR written by an XDL
MFMA must not be read or written by a VALU / VMEM / LDS instruction until (passes + 2) wait states later (16-pass: LLVM pads to exactly 18) (LLVM
GCNHazardRecognizer::checkMAIVALUHazards, GFX940_XDL_N_PassWriteVgprVALU*WaitStates); the hardware does not interlock.
LLVM inserts the s_nops for instructions it knows -- but an `asm()` statement is opaque to it: a v_fma_mix* / v_cvt written as
inline assembly that consumes a fresh MFMA result gets no wait states.  The listing is walked linearly per kernel (fall-through
order, labels ignored: a branch target is checked as if entered from the instruction above it), one wait state per issued
instruction, `s_nop n` = n + 1.

usage: isa_mfma_hazards.py file.s [kernel substring]      exit code 1 if a violation is found"""
import re
import sys

PASSES = {"v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_32x32x2f32": 16, "v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_32x32x8_f16": 16,
          "v_mfma_f32_16x16x4_f32": 8, "v_mfma_f32_16x16x32_f16": 4, "v_mfma_f32_32x32x16_bf16": 8}


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def check(rows, name):
    recent, bad, n_mfma, n_asm = [], [], 0, 0          # recent: [dst regs, wait states since issue, needed, text]
    in_asm = False
    for ln, l in rows:
        t = l.split(";")[0].strip()
        if "#ASMSTART" in l or "#ASMEND" in l:
            in_asm = "#ASMSTART" in l
            continue
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        ws = 1
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


def main():
    rows = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [i for i, l in enumerate(rows) if re.match(r"^_Z\w*:", l)]
    total = 0
    for s in starts:
        name = rows[s].split(":")[0]
        if want not in name:
            continue
        e = next(i for i in range(s, len(rows)) if "s_endpgm" in rows[i])
        bad, n_mfma, n_asm = check([(i + 1, rows[i]) for i in range(s, e)], name)
        if n_mfma == 0:
            continue
        print(f"{name[:110]}: {n_mfma} MFMAs, {n_asm} inline-asm instructions, {len(bad)} hazard(s)")
        for ln, t, txt, el, need, ia in bad[:12]:
            print(f"    line {ln}: `{t}`{' [inline asm]' if ia else ''} touches the result of `{txt}` after {el} of {need} wait states")
        total += len(bad)
    sys.exit(1 if total else 0)


main()
