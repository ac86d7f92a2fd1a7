"""GPU check of the 256-wide ping-pong GEMM (tr_mode 8 / 9) against the 128x128 production kernel (tr_mode 1) on the
model's shapes, all operand layouts and epilogues, repeated to screen for races; then timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import ACT_GELU, view
dev = torch.device("cuda:0")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
notime = "notime" in sys.argv


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def cmp(name, got, ref, tol=2e-2):
    err = (got.float() - ref.float()).abs().max().item()
    sc = ref.float().abs().max().item()
    ok = err <= tol * max(sc, 1.0) and torch.isfinite(got.float()).all().item()
    print(f"{'OK  ' if ok else 'FAIL'} {name}: err {err:.3e} / {sc:.3e}", flush=True)
    return ok


TRS = tuple(int(x) for x in os.environ.get("SMX_CHECK_TRS", "8").split(","))
KC_ONLY = tuple(int(x) for x in os.environ.get("SMX_CHECK_KC_ONLY", "").split(",") if x)   # kernels that only take K-contiguous operands


def main():
    torch.manual_seed(0)
    ok = True
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    shapes = [(300, 200, 136), (15968, 768, 768), (7968, 3072, 768), (15968, 768, 3072), (1000, 512, 1536), (4096, 2304, 768)]
    if quick:
        shapes = shapes[:3]
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device=dev).bfloat16()
        Wm = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = Wm.t().contiguous()
        bias = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev).bfloat16()
        P = torch.randn(M, N, device=dev).bfloat16()
        for tr in TRS:
            for rep in range(3 if not quick else 2):
                # fwd: bias + gelu + aux + resid + dropout
                Y1 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); X1 = torch.zeros_like(Y1)
                Y2 = torch.zeros_like(Y1); X2 = torch.zeros_like(Y1)
                kw = dict(bias=bias, act=ACT_GELU, resid=R, drop=(0.1, 11))
                ops.gemm(A, Wm, Y1, M, N, K, ops.BF16, aux_out=X1, tr_mode=1, **kw)
                ops.gemm(A, Wm, Y2, M, N, K, ops.BF16, aux_out=X2, tr_mode=tr, **kw)
                ok &= cmp(f"fwd {M}x{N}x{K} tr{tr} rep{rep} out", Y2, Y1) and cmp("    aux", X2, X1)
                if tr in KC_ONLY:
                    continue
                # dgrad: W rows-contiguous, aux_in gelu'
                Y1.zero_(); Y2.zero_()
                ops.gemm(A, Wt, Y1, M, N, K, ops.BF16, b_rc=True, bv=view(N), aux_in=P, act=ACT_GELU, tr_mode=1)
                ops.gemm(A, Wt, Y2, M, N, K, ops.BF16, b_rc=True, bv=view(N), aux_in=P, act=ACT_GELU, tr_mode=tr)
                ok &= cmp(f"dgrad {M}x{N}x{K} tr{tr} rep{rep}", Y2, Y1)
            if tr in KC_ONLY:
                continue
            # wgrad-like: C[N, K] = Y^T[M,N] A[M,K] reduction over M, split-K slabs
            Yb = torch.randn(M, N, device=dev).bfloat16()
            kst = (M + 63) // 64
            per = (kst + 4) // 5
            for split in (1, (kst + per - 1) // per):       # no empty K slice
                S1 = torch.zeros(split, N, K, dtype=torch.float32, device=dev); S2 = torch.zeros_like(S1)
                for S, t in ((S1, 1), (S2, tr)):
                    ops.gemm(Yb, A, S, N, K, M, ops.BF16, a_rc=True, b_rc=True, av=view(N), bv=view(K), out_f32=True,
                             split_k=split, split_stride=N * K if split > 1 else 0, tr_mode=t)
                ok &= cmp(f"wgrad {N}x{K}x{M} split{split} tr{tr}", S2.sum(0), S1.sum(0), 1e-3)
    # conv view (overlapping rows) + batched
    Bz, T, Cin, Cout, k, s = 4, 999, 64, 512, 3, 2
    To = (T - k) // s + 1
    x = torch.randn(Bz, T, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, k * Cin, device=dev) * 0.1).bfloat16()
    for tr in TRS:
        y1 = torch.zeros(Bz * To, Cout, dtype=torch.bfloat16, device=dev); y2 = torch.zeros_like(y1)
        ops.gemm(x, w, y1, Bz * To, Cout, k * Cin, ops.BF16, av=view(s * Cin, To, T * Cin), tr_mode=1)
        ops.gemm(x, w, y2, Bz * To, Cout, k * Cin, ops.BF16, av=view(s * Cin, To, T * Cin), tr_mode=tr)
        ok &= cmp(f"conv view tr{tr}", y2, y1)
        dy = torch.randn(Bz * To, Cout, device=dev).bfloat16()
        d1 = torch.zeros(3, Cout, k * Cin, dtype=torch.float32, device=dev); d2 = torch.zeros_like(d1)
        for d, t in ((d1, 1), (d2, 8 if tr in KC_ONLY else tr)):
            ops.gemm(dy, x, d, Cout, k * Cin, Bz * To, ops.BF16, a_rc=True, b_rc=True, av=view(Cout), bv=view(s * Cin, To, T * Cin),
                     out_f32=True, split_k=3, split_stride=Cout * k * Cin, tr_mode=t)
        ok &= cmp(f"conv wgrad view tr{tr}", d2.sum(0), d1.sum(0), 1e-3)
        G, M, N, K = 4, 700, 48, 192
        A = torch.randn(G, M, K, device=dev).bfloat16(); B = torch.randn(G, N, K, device=dev).bfloat16()
        c1 = torch.zeros(M, G * N, dtype=torch.bfloat16, device=dev); c2 = torch.zeros_like(c1)
        for c, t in ((c1, 1), (c2, tr)):
            ops.gemm(A, B, c, M, N, K, ops.BF16, cv=view(G * N), nbatch=G, batch_a=M * K, batch_b=N * K, batch_c=N, tr_mode=t)
        ok &= cmp(f"batched tr{tr}", c2, c1)
    print("ALL OK" if ok else "SOME FAILED", flush=True)
    if "notime" in sys.argv:
        return 0 if ok else 1

    # timings
    for (M, N, K) in [(15968, 3072, 768), (15968, 768, 3072), (15968, 2304, 768), (15968, 768, 768), (511968, 512, 1536),
                      (7968, 3072, 768)]:
        A = torch.randn(M, K, device=dev).bfloat16()
        Wm = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = Wm.t().contiguous()
        Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        P = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        R = torch.randn(M, N, device=dev).bfloat16()
        bias = torch.randn(N, device=dev)
        fl = 2.0 * M * N * K
        line = []
        for tr in (1,) + TRS:
            t0 = bench(lambda: ops.gemm(A, Wm, Y, M, N, K, ops.BF16, tr_mode=tr))
            t1 = bench(lambda: ops.gemm(A, Wm, Y, M, N, K, ops.BF16, bias=bias, act=ACT_GELU, aux_out=P, tr_mode=tr))
            t2 = bench(lambda: ops.gemm(A, Wm, Y, M, N, K, ops.BF16, bias=bias, resid=R, drop=(0.1, 7), tr_mode=tr))
            t3 = bench(lambda: ops.gemm(A, Wt, Y, M, N, K, ops.BF16, b_rc=True, bv=view(N), aux_in=P, act=ACT_GELU,
                                        tr_mode=8 if tr in KC_ONLY else tr))
            line.append(f"tr{tr}: plain {t0:.0f}us {fl/t0/1e6:.0f}TF | gelu+aux {t1:.0f}us {fl/t1/1e6:.0f}TF | "
                        f"resid+drop {t2:.0f}us {fl/t2/1e6:.0f}TF | dgrad gelu' {t3:.0f}us {fl/t3/1e6:.0f}TF")
        print(f"M={M} N={N} K={K}\n  " + "\n  ".join(line), flush=True)
    # wgrad shapes
    for (No, Ko, Mred, split1) in [(768, 3072, 15968, 7), (3072, 768, 15968, 7), (2304, 768, 15968, 9), (768, 768, 15968, 28)]:
        Yb = torch.randn(Mred, No, device=dev).bfloat16(); Xb = torch.randn(Mred, Ko, device=dev).bfloat16()
        fl = 2.0 * No * Ko * Mred
        line = []
        for tr, splits in ((1, (split1,)), (8, (4, 7, 14, 21))):
            for sp in splits:
                S = torch.empty(sp, No, Ko, dtype=torch.float32, device=dev)
                t = bench(lambda: ops.gemm(Yb, Xb, S, No, Ko, Mred, ops.BF16, a_rc=True, b_rc=True, av=view(No), bv=view(Ko),
                                           out_f32=True, split_k=sp, split_stride=No * Ko, tr_mode=tr))
                line.append(f"tr{tr}/s{sp}: {t:.0f}us {fl/t/1e6:.0f}TF")
        print(f"wgrad {No}x{Ko}x{Mred}: " + "  ".join(line), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
