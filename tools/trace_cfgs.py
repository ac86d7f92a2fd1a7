"""Kernel time per configuration from a rocprofv3 --kernel-trace CSV of a tool run in trace mode (a 64-element fill
marker, then 10 launches per configuration) joined with the tool's "CFG i name" lines.
   python tools/trace_cfgs.py trace.csv tool.log"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])))
rows.sort()
names = {}
for line in open(sys.argv[2]):
    if line.startswith("CFG "):
        _, i, name = line.rstrip().split(" ", 2)
        names[int(i)] = name
groups, cur = [], None
for s, e, name, gx in rows:
    if "FillFunctor" in name and gx <= 256:
        cur = []
        groups.append(cur)
    elif cur is not None:
        cur.append((s, e, name))
# a configuration's group ends where the next marker starts; the checks between configurations launch kernels too, so
# only the first 10 x (kernels per call) dispatches after a marker are counted: calls are identical, so the group's
# leading run of GEMM-family kernels is cut at a multiple of 10
out = {}
groups = [g for g in groups if g and any(k in g[0][2] for k in ("gemm_bf16", "splitk_epilogue", "reduce_slabs"))]
for i, g in enumerate(groups):
    fam = []
    for s, e, name in g:
        if any(k in name for k in ("gemm_bf16", "splitk_epilogue", "reduce_slabs")):
            fam.append((s, e))
        else:
            break
    n = len(fam) // 10 * 10
    fam = fam[:n]
    if not fam:
        continue
    busy = sum(e - s for s, e in fam) / 10 / 1e3
    span = (fam[-1][1] - fam[0][0]) / 10 / 1e3
    gem = [e - s for (s, e), (_, _, nm) in zip(fam, g) if "gemm_bf16" in nm]
    out[i] = (busy, span, n // 10, sum(gem) / max(len(gem), 1) / 1e3)
last = None
for i in sorted(out):
    name = names.get(i, "?")
    key = name.rsplit(" ", 1)[0]
    if key != last:
        print()
        print(key, end=": ")
        last = key
    print(f"{name.rsplit(' ', 1)[-1]}={out[i][0]:.1f}({out[i][3]:.1f})", end="  ")
print()
