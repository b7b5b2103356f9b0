#!/usr/bin/env python3
"""Dispatch order of one stacked sigma build in a rocprofv3 kernel trace of tools/eom_prof_many.py: a build starts with
k t2_layouts launches in a row (the pair layouts of the k trial vectors) and ends before the next such group.  The LAST build of
that script runs under per-GEMM events (phase launches off); argv[2] = 1 (default) takes the one before it — a timed build, as
the solver runs it —, 0 the last."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
lay = [i for i, n in enumerate(names) if 't2_layouts' in n]
# (the k layout launches of a build follow each other within 0.1 ms, a copy between them; builds are milliseconds apart)
starts = [i for j, i in enumerate(lay) if j == 0 or int(rows[i]['Start_Timestamp']) - int(rows[lay[j - 1]]['Start_Timestamp']) > 1000000]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a = starts[-1 - back]
b = starts[-back] if back > 0 else len(rows)
# (memcpy nodes are not kernels: the U1 copies do not show)
t0 = int(rows[a]['Start_Timestamp'])
prev = t0
tot = {}
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    print(f"{(s - t0) / 1e6:9.3f} ms  dur {(e - s) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f} us  grid {r.get('Grid_Size_X', '?'):>9s}  {name}")
    key = name.split('(')[0]
    tot[key] = tot.get(key, 0.0) + (e - s) / 1e3
    prev = e
print("# total span ms", (prev - t0) / 1e6)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"# {v:10.1f} us  {k}")
