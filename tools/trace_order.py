import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find last t2_layouts and print from the previous one
idx = [i for i, r in enumerate(rows) if 't2_layouts' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    print(f"{(s-t0)/1e6:9.3f} ms  dur {(e-s)/1e3:10.1f} us  gap {(s-prev_end)/1e3:7.1f} us  grid {r.get('Grid_Size_X','?'):>8s} {name}")
    prev_end = e
print("iteration", (int(rows[b]['Start_Timestamp']) - t0) / 1e6)
