"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV of bench.py (steps are delimited by the optimizer's
sumsq kernel): per kernel start (us from the step's start), duration, idle gap before it, stream / queue, grid, name; then
the step's busy / idle split and the time between marker kernels.
    python tools/step_timeline.py trace.csv [STEP_INDEX] [--full]"""
import csv, sys
path = sys.argv[1]
step = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else -2
full = "--full" in sys.argv
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                     int(r["Grid_Size_Z"]), r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
# a step ends with the optimizer's last kernel: the last of a run of af_* (Adafactor) / opt_kernel launches
is_opt = [("af_" in r[2] or "opt_kernel" in r[2] or "sumsq" in r[2]) for r in rows]
idx = [i for i in range(len(rows)) if is_opt[i] and (i + 1 == len(rows) or not is_opt[i + 1])]
a, b = idx[step - 1] if step != 0 else -1, idx[step]
sel = rows[a + 1:b + 1]
t0 = sel[0][0]
busy_end, busy, gaps = t0, 0.0, 0.0
out = []
for s, e, name, gx, gz, q in sel:
    gap = max(0, s - busy_end) / 1e3
    if s >= busy_end:
        busy += (e - s) / 1e3
    elif e > busy_end:
        busy += (e - busy_end) / 1e3
    gaps += gap
    busy_end = max(busy_end, e)
    short = name.split("(")[0]
    short = short.replace("void ", "")[:70]
    out.append(((s - t0) / 1e3, (e - s) / 1e3, gap, q, gx, gz, short))
wall = (busy_end - t0) / 1e3
print(f"# step wall {wall / 1e3:.3f} ms, busy (union) {busy / 1e3:.3f} ms, idle gaps {gaps / 1e3:.3f} ms, kernels {len(out)}, "
      f"sum of durations {sum(o[1] for o in out) / 1e3:.3f} ms")
if full:
    for t, d, g, q, gx, gz, n in out:
        print(f"{t:10.1f} {d:8.1f} {g:6.1f} q{q} {gx:6d}x{gz:<3d} {n}")
# coarse phases by marker kernels
marks = [("conv0", "cnn fwd start"), ("group_pack", "posconv"), ("attn_fwd", "first attention"), ("ce_", "loss"),
         ("attn2_dq", "first attention bwd"), ("c0m_bwd", "conv0 bwd"), ("af_stats", "optimizer")]
last = {}
for t, d, g, q, gx, gz, n in out:
    for key, label in marks:
        if key in n and label not in last:
            last[label] = t
for label, t in sorted(last.items(), key=lambda kv: kv[1]):
    print(f"# {t / 1e3:8.3f} ms  first {label}")
