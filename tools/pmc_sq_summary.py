#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ passes into one JSON: per kernel (the jet forward / reverse kernels and the generic set's maps),
mean of every counter over dispatches, plus derived shares:
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES)      (cycles the matrix pipe is busy, per SIMD-cycle of busy CUs)
  mfma_cyc_per_inst= SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA                      (32 = back-to-back v_mfma_f32_16x16x4_f32)
  valu_per_mfma    = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA
  wait shares      = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycle units, disjoint; guide: PMC slots)
  lds_conflict_frac= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
usage: pmc_sq_summary.py out.json "<command>" counter_collection.csv..."""
import collections
import csv
import json
import sys

out_path, cmd, files = sys.argv[1], sys.argv[2], sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"command": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> --output-format csv -- " + cmd, "kernels": {}}
for k, d in agg.items():
    if not any(s in k for s in ("f_forward", "f_backward", "g_fwd", "g_bwd", "w_forward", "w_bwd")):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = max(len(v) for v in d.values())
    der = {}
    g = m.get
    if g("SQ_BUSY_CU_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        der["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CU_CYCLES"))
    if g("SQ_INSTS_MFMA"):
        der["mfma_cyc_per_inst"] = g("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / g("SQ_INSTS_MFMA")
        if g("SQ_INSTS_VALU") is not None:
            der["valu_per_mfma"] = (g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA")
    if g("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if g(c) is not None:
                der[c.lower() + "_over_wave_cycles"] = g(c) / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE"):
        der["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_VALU_MFMA_BUSY_CYCLES"):
        der["mfma_valu_coexec_frac"] = g("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / g("SQ_VALU_MFMA_BUSY_CYCLES")
    res["kernels"][k] = {"dispatches": n, "counters": m, "derived": der}
json.dump(res, open(out_path, "w"), indent=1)
for k, v in res["kernels"].items():
    if v["counters"].get("SQ_INSTS_MFMA", 0) > 0:
        print(k[:70], json.dumps({a: round(b, 4) for a, b in v["derived"].items()}))
