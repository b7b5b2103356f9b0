import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
en=[i for i,r in enumerate(rows) if "energy_norms" in r["Kernel_Name"]]
a,b=en[-2]+1,en[-1]+1
t0=int(rows[a]["Start_Timestamp"])
tot=0
prev_end=t0
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    d=(e-s)/1e3
    tot+=d
    nm=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:50]
    print("%8.1f us  +%8.1f gap %5.1f  %-50s grid %s"%(d,(s-t0)/1e3,(s-prev_end)/1e3,nm,r["Grid_Size_X"]+"x"+r["Grid_Size_Y"]))
    prev_end=e
print("kernels",b-a,"sum us",tot,"span",(int(rows[b-1]["End_Timestamp"])-t0)/1e3)
