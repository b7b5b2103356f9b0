#!/usr/bin/env python3
"""Shader clock per dispatch from one rocprofv3 --pmc GRBM_GUI_ACTIVE pass (counter summed over the 8 XCDs):
    python3 tools/dispatch_clock.py <counter_collection.csv> [min_us]
prints, in launch order, every dispatch longer than min_us (default 200) with duration and cycles / duration."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
for r in rows:
    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if ns < 1e3 * min_us:
        continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50]
    print(f"{int(r['Dispatch_Id']):6d} {ns/1e3:10.1f} us  {float(r['Counter_Value'])/8/ns:6.3f} GHz  grid {r.get('Grid_Size','?'):>9s}  {name}")
