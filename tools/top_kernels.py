#!/usr/bin/env python
"""top kernels of a rocprofv3 kernel_stats.csv: python tools/top_kernels.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%9.2f ms %6s calls avg %10.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
