"""Static check of the built library's ISA for the "last workgroup finishes the InstanceNorm table" protocol of
gp-nerf_amd/csrc/gpnerf_conv.hip (finalize_if_last): every wave must have its write-through tile-sum stores
(`global_store_dword ... sc1`) ACKNOWLEDGED -- an `s_waitcnt vmcnt(0)` -- before the workgroup barrier that precedes the
ticket (`global_atomic_add <ret>, ... sc0`).  A workgroup-scope release fence does not emit that wait on gfx9 (it waits on
lgkmcnt only), so without the explicit wait the ticket can become visible while another wave's stores are still on their way
to their L2 channel, and the last workgroup -- possibly on another XCD -- sums stale tiles.  Timing-dependent, invisible to
golden vectors; hence a gate on the code the compiler actually emitted.

The listing is walked linearly per kernel (fall-through order): a sc1 store makes the wave "dirty", `s_waitcnt` with
vmcnt(0) makes it clean, `s_barrier` latches the state, and a returning sc0 atomic add behind a barrier that was crossed
dirty (or with no barrier at all since the last sc1 store) is a violation.

usage: isa_ticket_release.py lib.so|file.s [kernel substring]     exit code 1 if a violation is found"""
import importlib.util
import os
import re
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _hazards():
    spec = importlib.util.spec_from_file_location("isa_mfma_hazards", os.path.join(_HERE, "isa_mfma_hazards.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def check(rows):
    """-> (tickets found, [(line, text, why)])"""
    dirty, barrier_since_store, latched_dirty = False, False, False
    tickets, bad = 0, []
    for ln, l in rows:
        t = l.split("//")[0].split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("global_store") and re.search(r"\bsc1\b", t):
            dirty, barrier_since_store = True, False
        elif op == "s_waitcnt" and re.search(r"vmcnt\(0\)", t):
            dirty = False
        elif op == "s_barrier":
            latched_dirty, barrier_since_store = dirty, True
        elif op.startswith("global_atomic_add") and re.search(r"\bsc0\b", t):
            tickets += 1
            if not barrier_since_store:
                bad.append((ln, t, "no s_barrier between the last write-through store and the ticket"))
            elif latched_dirty:
                bad.append((ln, t, "the barrier in front of the ticket was entered with write-through stores not waited for (no s_waitcnt vmcnt(0))"))
    return tickets, bad


def scan(path, want="conv"):
    """[(kernel, tickets, violations)] for every kernel whose name contains `want` and that has write-through stores + a ticket"""
    hz = _hazards()
    rows = hz.listing_of(path)
    hdr = re.compile(r"^(?:[0-9a-f]{8,16} <(_Z\w+)>:|(_Z\w*):)")
    starts = [(i, (m.group(1) or m.group(2))) for i, l in enumerate(rows) for m in [hdr.match(l)] if m]
    res = []
    for j, (s, name) in enumerate(starts):
        if want not in name:
            continue
        e = starts[j + 1][0] if j + 1 < len(starts) else len(rows)
        body = [(i + 1, rows[i]) for i in range(s + 1, e)]
        if not any("global_store" in l and " sc1" in l for _, l in body):
            continue
        tickets, bad = check(body)
        if tickets:
            res.append((name, tickets, bad))
    return res


def main():
    total = 0
    for name, tickets, bad in scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "conv"):
        print(f"{name[:110]}: {tickets} ticket(s), {len(bad)} violation(s)")
        for ln, t, why in bad:
            print(f"    line {ln}: `{t}`: {why}")
        total += len(bad)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
