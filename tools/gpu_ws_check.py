"""Wave-specialised GEMM kernel (tr_mode 14, csrc/gemm_ws.hip) against the 128 x 128 kernel (bit-identity on ragged shapes, every epilogue
class the step uses) and against the free-running kernels (tr_mode 12 / 13) on the encoder shapes of config 2 (time per launch).
    python tools/gpu_ws_check.py [check] [time]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import ACT_GELU, view

dev = torch.device("cuda:0")
what = sys.argv[1:] or ["check", "time"]
WS = int(os.environ.get("SMX_WS_MODE", "14"))


def check(ncase=42, seed=11):
    rng = random.Random(seed)
    torch.manual_seed(seed)
    kinds = ["fwd", "fwd_act", "fwd_saved", "dgrad", "dgrad_actgrad", "dgrad_saved", "wgrad"]
    bad = compared = 0
    for case in range(ncase):
        M = rng.choice([264, 1000, 4000, 7968, 15968]) + 8 * rng.randrange(0, 4)
        N = 8 * rng.randrange(8, 400)
        K = 8 * rng.randrange(16, 200)
        kind = kinds[case % len(kinds)]
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = W.t().contiguous()
        bias = torch.randn(N, device=dev) * 0.1
        S = torch.randn(M, N, device=dev).bfloat16()
        ref, split = None, rng.choice([1, 3])
        for mode in (1, WS, WS):
            try:
                if kind == "wgrad":
                    kst = (M + 63) // 64
                    per = (kst + split - 1) // split
                    sp = (kst + per - 1) // per
                    G = torch.zeros(sp, N, K, dtype=torch.float32, device=dev)
                    ops.gemm(S, A, G, N, K, M, ops.BF16, a_rc=True, b_rc=True, av=view(N), bv=view(K), out_f32=True, split_k=sp,
                             split_stride=N * K if sp > 1 else 0, tr_mode=mode)
                    res = (G,)
                else:
                    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                    aux = torch.zeros_like(Y)
                    kw = {"fwd": dict(bias=bias, resid=S, drop=(0.1, 4)), "fwd_act": dict(bias=bias, act=ACT_GELU, aux_out=aux, drop=(0.1, 5)),
                          "fwd_saved": dict(bias=bias, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=aux, drop=(0.1, 6)),
                          "dgrad": dict(b_rc=True, bv=view(N), resid=S), "dgrad_actgrad": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU),
                          "dgrad_saved": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU | ops.ACT_SAVE_GRAD)}[kind]
                    ops.gemm(A, Wt if kw.get("b_rc") else W, Y, M, N, K, ops.BF16, tr_mode=mode, **kw)
                    res = (Y, aux)
            except RuntimeError as e:
                print("refused", case, kind, M, N, K, mode, str(e)[:60])
                continue
            torch.cuda.synchronize()
            if mode == 1:
                ref = res
            else:
                compared += 1
                for a_, b_ in zip(ref, res):
                    if not torch.equal(a_, b_):
                        d = (a_.float() - b_.float()).abs()
                        bad += 1
                        print("MISMATCH", case, kind, (M, N, K), "split", split, "max", d.max().item(), "count", int((d > 0).sum()), "of", d.numel(),
                              "first", torch.nonzero(d > 0)[:3].tolist(), flush=True)
    print(f"check: {compared} comparisons, {bad} mismatches", flush=True)
    return bad


REPS = int(os.environ.get("SMX_WS_REPS", "10"))          # back-to-back launches per timed interval


def timeit(fn, n=9):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1000.0 / REPS for e0, e1 in ev)
    return ts[len(ts) // 2]


def times():
    torch.manual_seed(0)
    M, d, F = 15968, 768, 3072
    rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).bfloat16()
    x = rnd(M, d); wqkv = rnd(3 * d, d, scale=0.03); bqkv = torch.randn(3 * d, device=dev) * 0.1
    y3 = torch.zeros(M, 3 * d, dtype=torch.bfloat16, device=dev)
    wo = rnd(d, d, scale=0.03); bo = torch.randn(d, device=dev) * 0.1; y1 = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
    w1 = rnd(F, d, scale=0.03); b1 = torch.randn(F, device=dev) * 0.1; pre = torch.zeros(M, F, dtype=torch.bfloat16, device=dev); f = torch.zeros(M, F, dtype=torch.bfloat16, device=dev)
    w2 = rnd(d, F, scale=0.03); b2 = torch.randn(d, device=dev) * 0.1
    dyF = rnd(M, F); dy3 = rnd(M, 3 * d)
    GSL = torch.zeros(3, d, F, dtype=torch.float32, device=dev)
    GS = ops.ACT_GELU | ops.ACT_SAVE_GRAD
    rows = [
        ("QKV fwd      2304x768  bias", 2.0 * M * 3 * d * d, lambda m: ops.gemm(x, wqkv, y3, M, 3 * d, d, ops.BF16, bias=bqkv, tr_mode=m)),
        ("out-proj fwd  768x768  bias+drop+resid", 2.0 * M * d * d, lambda m: ops.gemm(x, wo, y1, M, d, d, ops.BF16, bias=bo, resid=x, drop=(0.1, 1234), tr_mode=m)),
        ("FFN1 fwd     3072x768  bias+gelu(saved)+drop", 2.0 * M * F * d, lambda m: ops.gemm(x, w1, f, M, F, d, ops.BF16, bias=b1, act=GS, aux_out=pre, drop=(0.1, 77), tr_mode=m)),
        ("FFN1 fwd     3072x768  plain", 2.0 * M * F * d, lambda m: ops.gemm(x, w1, f, M, F, d, ops.BF16, tr_mode=m)),
        ("FFN2 fwd      768x3072 bias+drop+resid", 2.0 * M * F * d, lambda m: ops.gemm(f, w2, y1, M, d, F, ops.BF16, bias=b2, resid=x, drop=(0.1, 99), tr_mode=m)),
        ("FFN2 dgrad   3072x768  actgrad(saved)", 2.0 * M * F * d, lambda m: ops.gemm(x, w2, f, M, F, d, ops.BF16, b_rc=True, bv=view(F), aux_in=pre, act=GS, drop=(0.1, 77), tr_mode=m)),
        ("FFN1 dgrad    768x3072 resid", 2.0 * M * F * d, lambda m: ops.gemm(dyF, w1, y1, M, d, F, ops.BF16, b_rc=True, bv=view(d), resid=x, tr_mode=m)),
        ("out dgrad     768x768", 2.0 * M * d * d, lambda m: ops.gemm(x, wo, y1, M, d, d, ops.BF16, b_rc=True, bv=view(d), tr_mode=m)),
        ("QKV dgrad     768x2304 resid", 2.0 * M * 3 * d * d, lambda m: ops.gemm(dy3, wqkv, y1, M, d, 3 * d, ops.BF16, b_rc=True, bv=view(d), resid=x, tr_mode=m)),
        ("FFN2 wgrad    768x3072 K=15968 (one problem, 3 slices)", 2.0 * M * F * d,
         lambda m: ops.gemm(x, dyF, GSL, d, F, M, ops.BF16, a_rc=True, b_rc=True, av=view(d), bv=view(F), out_f32=True, split_k=3, split_stride=d * F, tr_mode=m)),
    ]
    print(f"{'shape':52s} " + " ".join(f"{'mode ' + str(m):>18s}" for m in (1, 12, 13, WS)))
    for name, fl, fn in rows:
        cells = []
        for m in (1, 12, 13, WS):
            try:
                us = timeit(lambda: fn(m))
                cells.append(f"{us:7.1f} us {fl / us * 1e-6:5.0f} TF")
            except RuntimeError:
                cells.append(f"{'-':>18s}")
        print(f"{name:52s} " + " ".join(f"{c:>18s}" for c in cells), flush=True)


rc = 0
if "check" in what:
    rc = check()
if "time" in what:
    times()
sys.exit(1 if rc else 0)
