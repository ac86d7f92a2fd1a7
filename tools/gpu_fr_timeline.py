"""Where a work item's time goes in the free-running 256-wide GEMM kernel, per encoder shape (config 2: 15 968 rows, d 768, F 3072), with the
epilogues the step uses (bias, residual, dropout, GELU with the saved derivative).  Needs the trace build of gemm_fr.hip:

    tools/lab/build_variant.sh frtrace gemm_fr.hip "-DSMX_FR_TRACE=1 -DSMX_TU=gemm_fr"
    SMX_LIB=tools/lab/libsmx_frtrace.so python tools/gpu_fr_timeline.py

Wave 0 of every workgroup stamps the 100-MHz clock at kernel entry, after the prologue, and per item at its start / after its K loop / after its
epilogue (stores issued, not acknowledged).  Printed: medians over workgroups, microseconds."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speechmix_amd import ops, _lib as L
from speechmix_amd.ops import view

dev = torch.device("cuda:0")
torch.manual_seed(0)
M, d, F = 15968, 768, 3072
lib = L.lib()
if not hasattr(lib, "smx_fr_trace_read"):
    sys.exit("needs the SMX_FR_TRACE build (see the docstring)")


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).bfloat16()


def trace():
    buf = np.zeros(256 * 64, dtype=np.uint64)
    torch.cuda.synchronize()
    rc = lib.smx_fr_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_ulonglong(buf.nbytes))
    assert rc == 0
    return buf.reshape(256, 64).astype(np.int64)


def clear():
    pass          # (stale stamps are recognised by the item count of the launch)


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]


def run(name, fn, items_hint):
    if ONLY and not any(o in name for o in ONLY):
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    t = trace() * 0.01          # us
    W = items_hint
    grid = min(W, 256)
    rows = []
    t0 = t[:grid, 0].min()
    for b in range(grid):
        n = (W - b + 255) // 256 if grid == 256 else 1
        rows.append((n, t[b]))
    nmax = max(n for n, _ in rows)
    out = [f"{name}: launch {us:.1f} us, {W} items on {grid} workgroups"]
    ent = np.array([r[0] - t0 for _, r in rows])
    pro = np.array([r[1] - r[0] for _, r in rows])
    out.append(f"   entry spread (first..last workgroup) {ent.min():.2f} .. {np.median(ent):.2f} .. {ent.max():.2f} | prologue median {np.median(pro):.2f} max {pro.max():.2f}")
    for j in range(1, nmax + 1):
        sel = [r for n, r in rows if n >= j]
        st = np.array([r[3 * j - 1] for r in sel]); kl = np.array([r[3 * j] for r in sel]); ep = np.array([r[3 * j + 1] for r in sel])
        prev = np.array([r[3 * j - 2] if j > 1 else r[1] for r in sel])
        out.append(f"   item {j} ({len(sel)} wgs): start-gap {np.median(st - prev):.2f} | K loop {np.median(kl - st):.2f} (max {np.max(kl - st):.2f}) | epilogue {np.median(ep - kl):.2f} (max {np.max(ep - kl):.2f}) | ends at {np.median(ep - t0):.1f} (max {np.max(ep - t0):.1f})")
    # epilogue of the LAST item of each workgroup, first group of row blocks: entry | side inputs ready | arithmetic done | lanes exchanged | stores issued
    e = np.array([r[40:45] for _, r in rows])
    if e[:, 4].min() > 0:
        d = np.diff(e, axis=1)
        out.append("   last item's epilogue, first group: side inputs %.2f | arithmetic %.2f | lane exchange %.2f | stores %.2f  (medians, us)" % tuple(np.median(d, axis=0)))
    # K tiles of one item (the second one where every workgroup has two, else the first): time of each tile, wave 0
    j = 2 if W >= 2 * grid else 1
    kt = np.array([np.concatenate(([r[3 * j - 1]], r[45:61])) for _, r in rows])
    d = np.diff(kt, axis=1)
    nk = int((d[0] > 0).sum())
    if nk:
        out.append(f"   item {j}, K tiles (medians, us): " + " ".join(f"{np.median(d[:, i]):.2f}" for i in range(nk)))
    last = max(r[3 * n + 1] - t0 for n, r in rows)
    out.append(f"   last stamp at {last:.1f} us after the first entry (launch {us:.1f}: drain + launch overhead {us - last:.1f})")
    print("\n".join(out), flush=True)


x = rnd(M, d); wqkv = rnd(3 * d, d, scale=0.03); bqkv = torch.randn(3 * d, device=dev) * 0.1
y3 = torch.zeros(M, 3 * d, dtype=torch.bfloat16, device=dev)
wo = rnd(d, d, scale=0.03); bo = torch.randn(d, device=dev) * 0.1; y1 = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
w1 = rnd(F, d, scale=0.03); b1 = torch.randn(F, device=dev) * 0.1; pre = torch.zeros(M, F, dtype=torch.bfloat16, device=dev); f = torch.zeros(M, F, dtype=torch.bfloat16, device=dev)
w2 = rnd(d, F, scale=0.03); b2 = torch.randn(d, device=dev) * 0.1
dyF = rnd(M, F); dy3 = rnd(M, 3 * d)
GELU_SAVE = ops.ACT_GELU | ops.ACT_SAVE_GRAD


def items(N, mt):
    return ((M + mt - 1) // mt) * ((N + 255) // 256)


MODES = [tuple(int(v) for v in m.split(":")) for m in os.environ.get("SMX_TL_MODES", "13:192,12:256").split(",")]
for mode, mt in MODES:
    print(f"==== tr_mode {mode} ({mt} x 256 tiles)   {os.path.basename(os.environ.get('SMX_LIB', ''))}")
    run("QKV fwd      N 2304 K 768  bias", lambda: ops.gemm(x, wqkv, y3, M, 3 * d, d, ops.BF16, bias=bqkv, tr_mode=mode), items(3 * d, mt))
    run("QKV fwd      N 2304 K 768  plain", lambda: ops.gemm(x, wqkv, y3, M, 3 * d, d, ops.BF16, tr_mode=mode), items(3 * d, mt))
    run("out-proj fwd N 768  K 768  bias+drop+resid", lambda: ops.gemm(x, wo, y1, M, d, d, ops.BF16, bias=bo, resid=x, drop=(0.1, 1234), tr_mode=mode), items(d, mt))
    run("FFN1 fwd     N 3072 K 768  bias+gelu(saved)+drop", lambda: ops.gemm(x, w1, f, M, F, d, ops.BF16, bias=b1, act=GELU_SAVE, aux_out=pre, drop=(0.1, 77), tr_mode=mode), items(F, mt))
    run("FFN1 fwd     N 3072 K 768  plain", lambda: ops.gemm(x, w1, f, M, F, d, ops.BF16, tr_mode=mode), items(F, mt))
    run("FFN2 fwd     N 768  K 3072 bias+drop+resid", lambda: ops.gemm(f, w2, y1, M, d, F, ops.BF16, bias=b2, resid=x, drop=(0.1, 99), tr_mode=mode), items(d, mt))
    run("FFN2 dgrad   N 3072 K 768  actgrad(saved)", lambda: ops.gemm(x, w2, f, M, F, d, ops.BF16, b_rc=True, bv=view(F), aux_in=pre, act=GELU_SAVE, drop=(0.1, 77), tr_mode=mode), items(F, mt))
    run("FFN1 dgrad   N 768  K 3072 resid", lambda: ops.gemm(dyF, w1, y1, M, d, F, ops.BF16, b_rc=True, bv=view(d), resid=x, tr_mode=mode), items(d, mt))
    run("out dgrad    N 768  K 768", lambda: ops.gemm(x, wo, y1, M, d, d, ops.BF16, b_rc=True, bv=view(d), tr_mode=mode), items(d, mt))
    run("QKV dgrad    N 768  K 2304 resid", lambda: ops.gemm(dy3, wqkv, y1, M, d, 3 * d, ops.BF16, b_rc=True, bv=view(d), resid=x, tr_mode=mode), items(d, mt))
