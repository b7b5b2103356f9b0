#!/usr/bin/env python3
"""Sum of each counter of a rocprofv3 counter_collection.csv per kernel name (and dispatch count):
    python3 tools/pmc_rows.py <counter_collection.csv> [name substring]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
dur = defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in cnt[name]:
        cnt[name].add(r["Dispatch_Id"])
        dur[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for name, d in acc.items():
    n = len(cnt[name])
    print(f"{name}  dispatches {n}  avg {dur[name]/n/1e3:.1f} us")
    for k, v in sorted(d.items()):
        print(f"    {k:32s} {v/n:16.0f} per dispatch")
