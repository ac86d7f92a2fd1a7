"""Microbench: cost of the GEMM epilogue features on the in-model shapes (events, warmed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import ACT_GELU, ACT_NONE
dev = torch.device("cuda:0")
tr = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def bench(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in [(15968, 3072, 768), (15968, 768, 3072), (15968, 2304, 768), (15968, 768, 768), (511968, 512, 1536), (1024, 768, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Wt = (torch.randn(K, N, device=dev) * 0.05).bfloat16()
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    P = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    R = torch.randn(M, N, device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    fl = 2.0 * M * N * K
    res = {}
    res["plain"] = bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, tr_mode=tr))
    res["bias"] = bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, tr_mode=tr))
    res["bias+gelu+aux"] = bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, act=ACT_GELU, aux_out=P, tr_mode=tr))
    res["bias+resid"] = bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, resid=R, tr_mode=tr))
    res["bias+resid+drop"] = bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, resid=R, drop=(0.1, 7), tr_mode=tr))
    res["dgrad plain"] = bench(lambda: ops.gemm(A, Wt, Y, M, N, K, ops.BF16, b_rc=True, bv=ops.view(N), tr_mode=tr))
    res["dgrad aux_in gelu"] = bench(lambda: ops.gemm(A, Wt, Y, M, N, K, ops.BF16, b_rc=True, bv=ops.view(N), aux_in=P, act=ACT_GELU, tr_mode=tr))
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{k} {v:.1f}us ({fl / v / 1e6:.0f}TF)" for k, v in res.items()), flush=True)
