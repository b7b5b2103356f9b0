import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last build: from the last-but-(k) t2_layouts ... simply print the last N dispatches where N = count between the 4th-from-last group
names = [r['Kernel_Name'] for r in rows]
# find indices of 'ladder_sym'-like first kernel of a build: use the stacked ring GEMM '<true, true>' pairs: last two mark last build
idx = [i for i, n in enumerate(names) if 'dgemm_glds_kernel<true, true>' in n]
start = idx[-2]
# walk back to the previous build's last kernel: go back 60 dispatches
start = max(0, start - 40)
t0 = int(rows[start]['Start_Timestamp']); prev = t0
tot = {}
for r in rows[start:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:58]
    print(f"{(s-t0)/1e6:9.3f} ms dur {(e-s)/1e3:9.1f} us gap {(s-prev)/1e3:7.1f} {name}")
    prev = e
