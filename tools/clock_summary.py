#!/usr/bin/env python3
"""Effective shader clock of the dominant kernels under rocprofv3: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration of the SAME
dispatch (MI355X_MICROARCH.md, DVFS give-back).  usage: clock_summary.py out.json counter_collection.csv [kernel_trace.csv]"""
import collections, csv, json, sys
out, cc = sys.argv[1], sys.argv[2]
kt = sys.argv[3] if len(sys.argv) > 3 else None
dur = {}
if kt:
    for r in csv.DictReader(open(kt)):
        dur[r.get("Dispatch_Id")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    else:
        d = dur.get(r.get("Dispatch_Id"))
    if not d:
        continue
    agg[r["Kernel_Name"].split("(")[0].strip()].append((float(r["Counter_Value"]) / 8.0, d))
res = {}
for k, v in agg.items():
    if not any(s in k for s in ("f_forward", "f_backward", "w_forward", "w_bwd", "g_fwd", "g_bwd")):
        continue
    v = v[len(v) // 4:]                         # drop the first quarter (warm-up steps: clock not settled)
    cyc = sum(a for a, _ in v) / len(v)
    ns = sum(b for _, b in v) / len(v)
    res[k] = dict(dispatches=len(v), cycles_per_xcd=cyc, duration_us=ns / 1e3, clock_ghz=cyc / ns)
    print(f"{k[:60]:60s} n={len(v):4d}  {ns / 1e3:10.1f} us  {cyc / ns:.3f} GHz")
json.dump(res, open(out, "w"), indent=1)
