"""Per-step kernel table from a rocprofv3 --kernel-trace CSV of bench.py: drops the warm-up steps (steps are delimited
by the optimizer: Adafactor's af_gnorm kernel, else the sumsq kernel of the AdamW / SGD path) and aggregates by (kernel, grid).   python tools/rocprof_steps.py trace.csv WARMUP"""
import collections, csv, sys
path, warm = sys.argv[1], int(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Grid_Size_Z"]),
                     int(r["Workgroup_Size_X"])))
rows.sort()
idx = [i for i, r in enumerate(rows) if "af_gnorm" in r[2]] or [i for i, r in enumerate(rows) if "sumsq_kernel" in r[2]]
sel = rows[idx[warm - 1]:idx[-1]] if warm > 0 else rows[:idx[-1]]
steps = len(idx) - warm
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, name, gx, gz, wg in sel:
    n = name.split("(")[0][-64:]
    agg[(n, gx // wg, gz)][0] += 1
    agg[(n, gx // wg, gz)][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
wall = (sel[-1][1] - sel[0][0]) / 1e6
print(f"# {steps} timed steps; kernel time/step {tot / steps / 1e3:.2f} ms; wall/step {wall / steps:.2f} ms")
print("# ms/step  calls/step  avg_us  grid(x*z)  kernel")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / steps / 1e3:8.3f} {n / steps:7.1f} {t / n:9.1f}  {k[1]:6d}x{k[2]:<3} {k[0]}")
