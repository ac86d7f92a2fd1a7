"""Randomised shape screen of the 256-wide GEMM kernels (tr_mode 8 / 12 / 13) against the 128x128 kernel (tr_mode 1): forward,
data-gradient (rows-contiguous B: the round-3 LDS image) and weight-gradient layouts, linear / activation / saved-derivative
epilogues, ragged M / N / K, repeated launches.  Bit-identical outputs expected (same K order, same epilogue arithmetic).
   python tools/gpu_gemm_fuzz.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import ACT_GELU, view
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(0)
bad = compared = refused = 0
for case in range(cases):
    M = rng.choice([256, 264, 1000, 1024, 4000, 7968, 15968, 20000]) + 8 * rng.randrange(0, 4)
    N = 8 * rng.randrange(8, 400)
    K = 8 * rng.randrange(4, 260)
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Wt = W.t().contiguous()
    bias = torch.randn(N, device=dev) * 0.1
    S = torch.randn(M, N, device=dev).bfloat16()
    kind = rng.choice(["fwd", "fwd_act", "fwd_saved", "dgrad", "dgrad_actgrad", "dgrad_saved", "wgrad"])
    outs = {}
    for mode in (1, 8, 12, 13, 14, 12, 13, 14):
        try:
            if kind == "wgrad":
                dY = S
                kst = (M + 63) // 64
                split = rng.choice([1, 3]) if mode == 1 else outs["split"]
                per = (kst + split - 1) // split
                sp = (kst + per - 1) // per
                outs["split"] = split
                G = torch.zeros(sp, N, K, dtype=torch.float32, device=dev)
                ops.gemm(dY, A, G, N, K, M, ops.BF16, a_rc=True, b_rc=True, av=view(N), bv=view(K), out_f32=True, split_k=sp,
                         split_stride=N * K if sp > 1 else 0, tr_mode=mode)
                res = (G,)
            else:
                Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                aux = torch.zeros_like(Y)
                kw = {"fwd": dict(bias=bias, resid=S), "fwd_act": dict(bias=bias, act=ACT_GELU, aux_out=aux, drop=(0.1, 5)),
                      "fwd_saved": dict(bias=bias, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=aux, drop=(0.1, 6)),
                      "dgrad": dict(b_rc=True, bv=view(N), resid=S), "dgrad_actgrad": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU),
                      "dgrad_saved": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU | ops.ACT_SAVE_GRAD)}[kind]
                ops.gemm(A, Wt if kw.get("b_rc") else W, Y, M, N, K, ops.BF16, tr_mode=mode, **kw)
                res = (Y, aux)
        except RuntimeError:
            refused += 1
            continue
        if mode == 1:
            outs["ref"] = res
        elif (compared := compared + 1) and not all(torch.equal(a, b) for a, b in zip(outs["ref"], res)):
            bad += 1
            d = max((a.float() - b.float()).abs().max().item() for a, b in zip(outs["ref"], res))
            print(f"MISMATCH case {case} {kind} M={M} N={N} K={K} mode {mode}: max diff {d:.3e}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"TOTAL: {compared} comparisons against the 128x128 kernel, {refused} launches refused, {bad} mismatches")
