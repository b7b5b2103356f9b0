#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace run, from its rocpd database or
its *_kernel_trace.csv:  python3 tools/kernel_stats.py <run_results.db | kernel_trace.csv> > profiles/rNN/..._kernel_stats.csv"""
import csv
import sqlite3
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
path = sys.argv[1]
if path.endswith(".db"):
    cur = sqlite3.connect(path).cursor()
    for name, start, end in cur.execute("select name, start, end from kernels"):
        acc[name][0] += 1
        acc[name][1] += end - start
else:
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"]][0] += 1
        acc[row["Kernel_Name"]][1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
tot = sum(v[1] for v in acc.values())
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage"')
for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f'"{name.replace("(anonymous namespace)::", "")}",{n},{int(t)},{t / n:.1f},{100.0 * t / tot:.2f}')
