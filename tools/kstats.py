#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: name (40 chars), calls, average us."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['AverageNs']) * int(r['Calls']) < 2e4: continue
    print("%-58s %4s calls  %10.1f us avg" % (r['Name'].split('(')[0][:58], r['Calls'], float(r['AverageNs']) / 1e3))
