#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter over dispatches."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if not any(s in k for s in (sys.argv[2:] or ['f_forward', 'f_backward'])):
        continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%d mean=%.4g' % (c, len(v), sum(v) / len(v)))
