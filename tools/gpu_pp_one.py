"""One ping-pong GEMM shape, a few launches (profiling target).  argv: M N K tr [epi]   epi: 0 plain, 1 bias+gelu+aux"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
dev = torch.device("cuda:0")
M, N, K, tr = (int(x) for x in sys.argv[1:5])
epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev); P = torch.empty_like(Y); bias = torch.randn(N, device=dev)
Wt = W.t().contiguous()
for _ in range(5):
    if epi == 2:
        ops.gemm(A, Wt, Y, M, N, K, ops.BF16, b_rc=True, bv=ops.view(N), tr_mode=tr)
    elif epi:
        ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, act=ops.ACT_GELU, aux_out=P, tr_mode=tr)
    else:
        ops.gemm(A, W, Y, M, N, K, ops.BF16, tr_mode=tr)
torch.cuda.synchronize()
