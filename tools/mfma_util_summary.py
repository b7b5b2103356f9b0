#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from three separate rocprofv3 --pmc passes of the same command
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_MFMA):

    python3 tools/mfma_util_summary.py <busy csv> <gui_active csv> <insts csv> > profiles/rNN/..._pmc_mfma_util.csv

MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (cycles * 1024 SIMDs) (rocprofv3's derived-metric definition), where
cycles = GRBM_GUI_ACTIVE / 8: the per-dispatch value in the csv is the SUM over the 8 XCDs (it equals 8 x duration x
clock).  The average shader clock during the kernel follows as cycles / duration.
Dispatches are matched across the passes by their order of launch."""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].replace("(anonymous namespace)::", ""), float(r["Counter_Value"]),
             int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]


busy = load(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES")
act = load(sys.argv[2], "GRBM_GUI_ACTIVE")
ins = load(sys.argv[3], "SQ_INSTS_MFMA")
assert len(busy) == len(act) == len(ins), (len(busy), len(act), len(ins))
XCDS = 8
acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for (n1, b, _), (n2, a, ns), (n3, i, _) in zip(busy, act, ins):
    assert n1 == n2 == n3
    e = acc[n1]
    e[0] += 1; e[1] += b; e[2] += a / XCDS; e[3] += i; e[4] += ns
print("# rocprofv3 --pmc <counter> in SEPARATE passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`")
print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs), cycles = GRBM_GUI_ACTIVE / 8 XCDs; "
      "busy_cycles_per_mfma = BUSY / SQ_INSTS_MFMA; clock = cycles / kernel duration (of the GRBM pass)")
print("kernel,launches,mfma_util_percent,busy_cycles_per_mfma,avg_clock_GHz,cycles_total")
for name, (n, b, a, i, ns) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
    if i <= 0:
        continue
    print(f'"{name}",{n},{100.0 * b / (a * 1024.0):.2f},{b / i:.2f},{a / ns:.3f},{a:.0f}')
