#!/usr/bin/env python3
"""profiles/ summary of a tools/pmc_passes.sh run: raw counter means per dispatch of the fused kernel + the derived figures
DESIGN.md quotes.
usage: pmc_derive.py gpurun_out/pmc_<tag>/summary.json "<workload>" "<kernel>" <commit> out.json [--traffic profiles/pmc_traffic.json]

Derivations (MI355X_MICROARCH.md, HBM / rocprofv3 section): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the launch's shader
cycles are a eighth of it; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, so the matrix pipe's busy fraction is
value / (1024 x cycles); SQ_ACTIVE_INST_VALU counts quad-cycles; FETCH_SIZE / WRITE_SIZE are in KB, FETCH_SIZE x 2 on gfx950."""
import json
import sys

src, workload, kernel, commit, out = sys.argv[1:6]
loaded = json.load(open(src))
parts = loaded.pop("parts", None)          # (pmc_summary.py "a+b+c": the kernels of one call, summed; each part's own means)
raw = {k: v["mean_per_dispatch"] for k, v in loaded.items()}
n = min(v["dispatches"] for v in loaded.values())
cyc = raw["GRBM_GUI_ACTIVE"] / 8.0
d = {"shader_cycles_per_launch": cyc}
if "SQ_VALU_MFMA_BUSY_CYCLES" in raw:
    d["mfma_pipe_busy_frac"] = raw["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
if "SQ_ACTIVE_INST_VALU" in raw:
    d["valu_active_frac"] = raw["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0)
d["valu_instructions_per_launch"] = raw.get("SQ_INSTS_VALU")
d["mfma_instructions_per_launch"] = raw.get("SQ_INSTS_MFMA")
d["hbm_read_bytes"] = raw["FETCH_SIZE"] * 1024.0 * 2.0
d["hbm_write_bytes"] = raw["WRITE_SIZE"] * 1024.0
d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
d["fetch_correction"] = "x2: gfx950 FETCH_SIZE tallies 128-B requests of 16-B-per-lane loads at 64 B (MI355X_MICROARCH.md, HBM)"
d["write_note"] = ("WRITE_SIZE counts 64-B write requests; the per-lane 4-byte stores of weights[N,S] are partial-line requests, so this is "
                   "above the payload")
d["l2_hit_rate"] = raw["TCC_HIT_sum"] / (raw["TCC_HIT_sum"] + raw["TCC_MISS_sum"])
d["lds_bank_conflict_cycles"] = raw.get("SQ_LDS_BANK_CONFLICT")
res = {"workload": workload, "kernel": kernel, "commit": commit, "dispatches_averaged": n,
       "collection": "rocprofv3 --pmc, 4 separate passes (tools/pmc_passes.sh), counters only", "raw_mean_per_dispatch": raw, "derived": d}
if parts:
    per = {}
    for name, p in parts.items():
        r = {k: v["mean_per_dispatch"] for k, v in p.items()}
        c = r["GRBM_GUI_ACTIVE"] / 8.0
        per[name] = {"dispatches_per_call": p["GRBM_GUI_ACTIVE"]["dispatches"] / float(n), "shader_cycles_per_dispatch": c,
                     "mfma_pipe_busy_frac": r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c * 1024.0),
                     "valu_active_frac": r.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (c * 1024.0),
                     "hbm_bytes_per_dispatch": r["FETCH_SIZE"] * 2048.0 + r["WRITE_SIZE"] * 1024.0,
                     "l2_hit_rate": r["TCC_HIT_sum"] / max(1.0, r["TCC_HIT_sum"] + r["TCC_MISS_sum"])}
    res["per_kernel"] = per
    res["note"] = ("one gpnerf_render_fused call = the kernels under per_kernel; raw_mean_per_dispatch / derived are per CALL: every dispatch of "
                   "these kernels added up (busy fractions over the summed shader cycles)")
json.dump(res, open(out, "w"), indent=1)
if "--traffic" in sys.argv:
    t = sys.argv[sys.argv.index("--traffic") + 1]
    json.dump({"workload": workload + ", 1 launch", "FETCH_SIZE_KB": raw["FETCH_SIZE"], "WRITE_SIZE_KB": raw["WRITE_SIZE"],
               "fetch_correction": d["fetch_correction"], "hbm_bytes_per_launch": d["hbm_bytes_per_launch"], "l2_hit_rate": d["l2_hit_rate"],
               "mfma_pipe_busy_frac": d.get("mfma_pipe_busy_frac"),
               "source": f"{out} (rocprofv3 --pmc, separate passes, tools/pmc_passes.sh, commit {commit}; profile-derived, not measured inside this bench run)"},
              open(t, "w"), indent=1)
print(json.dumps(d, indent=1))
