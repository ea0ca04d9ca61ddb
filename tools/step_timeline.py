#!/usr/bin/env python3
"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace csv (start offset, duration, gap)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
upd = [i for i, n in enumerate(names) if "k_update" in n and "part" not in n]
a, b = upd[len(upd) // 2], upd[len(upd) // 2 + 1]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f us  (gap %5.1f)  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:70]))
    prev_end = max(prev_end, e)
print("step: %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
