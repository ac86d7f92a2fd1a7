"""Raw kernel durations around the first marker of a rocprofv3 --kernel-trace CSV (debug aid for tools/trace_cfgs.py)."""
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])))
rows.sort()
i0=[i for i,r in enumerate(rows) if "FillFunctor" in r[2] and r[3]<=256]
print(len(rows), i0[:8])
for s,e,n,g in rows[i0[1]-1:i0[1]+48]: print(e-s, n[:60], g)
