"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid) -> per-step time table."""
import csv, sys, collections
path, steps = sys.argv[1], float(sys.argv[2])
agg = collections.defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0][-60:]
        key = (name, r["Grid_Size_X"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[key][0] += 1
        agg[key][1] += d
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print(f"total kernel time/step: {tot/steps/1e3:.2f} ms")
for (name, gx, gz, wg), (n, t) in rows[:45]:
    print(f"{t/steps/1e3:8.3f} ms/step  {n/steps:6.1f} calls  avg {t/n:8.1f} us  grid {int(gx)//int(wg):6d} x{gz:>3}  {name}")
