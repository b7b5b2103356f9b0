import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find last t2_layouts and print from the previous one
# (anchor of an iteration: the pair-layout kernel; with phase launches on it runs inside phase_kernel and the grouped
# LDS-DMA launch of the ladder halves — one per iteration at the sizes this tool is used for — takes its place; argv[3] overrides)
anchor = sys.argv[3] if len(sys.argv) > 3 else ('dgemm_glds_group_kernel' if any('phase_kernel' in r['Kernel_Name'] for r in rows) else 't2_layouts')
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
# argv[2]: which iteration, counted from the end in t2_layouts launches (default 1: the last complete one; bench.py's
# default run ends with an eager pass under per-GEMM events, its timed, graph-replayed steps come before that)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = idx[-1 - back], idx[-back]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    print(f"{(s-t0)/1e6:9.3f} ms  dur {(e-s)/1e3:10.1f} us  gap {(s-prev_end)/1e3:7.1f} us  grid {r.get('Grid_Size_X','?'):>8s} {name}")
    prev_end = e
print("iteration", (int(rows[b]['Start_Timestamp']) - t0) / 1e6)
