#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic per launch.

    python3 tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN/..._pmc_hbm_traffic.csv

Counters are in KiB.  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md,
HBM section); WRITE_SIZE is taken as is."""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
        acc[name][0] += 1
        acc[name][1] += float(row["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`")
print("# counters are in KiB; FETCH_SIZE is DOUBLED below (gfx950 reports 1/2 of a wide coalesced read, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is")
print("kernel,launches,fetch_GB_per_launch_corrected,write_GB_per_launch")
rows = []
for name, (n, kib) in fetch.items():
    wn, wkib = write.get(name, (0, 0.0))
    rows.append((2.0 * kib * 1024 / 1e9, name, n, 2.0 * kib * 1024 / 1e9 / n, (wkib * 1024 / 1e9 / wn) if wn else 0.0))
for _, name, n, f, w in sorted(rows, reverse=True):
    print(f'"{name}",{n},{f:.4f},{w:.4f}')
