"""GPU busy / idle accounting of ONE training step from a rocprofv3 --kernel-trace CSV of bench.py, all queues together
(a step = from one af_stats_kernel start to the next): union busy time, idle time split by gap length, the largest gaps with
the kernels on either side, and the idle time by the kernel family that FOLLOWS the gap (who was late).
    python tools/step_gaps.py trace.csv [STEP_INDEX_FROM_END=2] [--list]"""
import collections, csv, sys
path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60],
                     int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r.get("Queue_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "af_stats" in r[2]]
a, b = marks[-back - 1], marks[-back]
sel = rows[a:b]
t0 = sel[0][0]
end = t0
busy = 0
gaps = []
for i, (s, e, n, g, q) in enumerate(sel):
    if s > end:
        gaps.append((s - end, i))
        busy += e - s
    elif e > end:
        busy += e - end
    end = max(end, e)
wall = end - t0
idle = sum(g for g, _ in gaps)
print(f"# step wall {wall / 1e6:.3f} ms; busy (union over queues) {busy / 1e6:.3f} ms; idle {idle / 1e6:.3f} ms in {len(gaps)} gaps; "
      f"{len(sel)} kernels, sum of durations {sum(e - s for s, e, *_ in sel) / 1e6:.3f} ms")
hist = collections.OrderedDict((k, [0, 0]) for k in ("<1us", "1-2us", "2-4us", "4-8us", "8-20us", ">20us"))
for g, _ in gaps:
    k = "<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else "8-20us" if g < 20000 else ">20us"
    hist[k][0] += 1
    hist[k][1] += g
print("# gaps: " + "; ".join(f"{k}: {n} = {t / 1e3:.0f} us" for k, (n, t) in hist.items()))
late = collections.defaultdict(lambda: [0, 0])
for g, i in gaps:
    late[sel[i][2]][0] += 1
    late[sel[i][2]][1] += g
print("# idle time by the kernel that follows the gap:")
for n, (c, t) in sorted(late.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 1e3:8.1f} us  {c:4d} gaps  avg {t / c / 1e3:5.1f}  {n}")
if "--list" in sys.argv:
    for s, e, n, g, q in sel:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} q{q} {g:6d} {n}")
