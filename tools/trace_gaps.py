#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 kernel trace: the largest gaps and the busy fraction over the last N ms."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e9
t_end = int(rows[-1]['End_Timestamp'])
sel = [r for r in rows if int(r['Start_Timestamp']) >= t_end - win]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel)
span = int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])
print(f"window {span / 1e6:.2f} ms, kernels {len(sel)}, busy {busy / 1e6:.2f} ms = {100.0 * busy / span:.2f} %")
gaps = []
for a, b in zip(sel, sel[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    gaps.append((g, a['Kernel_Name'][:50], b['Kernel_Name'][:50]))
gaps.sort(reverse=True)
for g, a, b in gaps[:12]:
    print(f"  gap {g / 1e3:8.1f} us  after {a}  before {b}")
print(f"  gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000)}, sum of all gaps {sum(g[0] for g in gaps) / 1e6:.3f} ms")
