#!/usr/bin/env python3
"""kstat.py <run_kernel_stats.csv> <substring> [...]: calls / average / min / max (us) of the kernels whose name contains a substring."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for pat in sys.argv[2:]:
    for r in rows:
        if pat in r["Name"]:
            print(f"{pat:28s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}  | {r['Name'][:60]}")
