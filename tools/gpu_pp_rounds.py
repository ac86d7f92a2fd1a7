"""Per-item overhead of the ping-pong GEMM: time vs number of rounds (items per workgroup) at fixed K.
argv: K [lab flags...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from tools.gpu_check_pp import bench
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 768
labs = [int(x) for x in sys.argv[2:]] or [0]
N = 1024
for tr in [8 + (l << 8) for l in labs] + [int(x) for x in os.environ.get("SMX_ROUNDS_EXTRA", "").split(",") if x] + [1]:
    ts = []
    for r in (1, 2, 4, 8):
        M = 16128 * r
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ts.append(bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, tr_mode=tr), n=10))
    per = (ts[3] - ts[1]) / 6
    print(f"tr{tr & 255} lab{tr >> 8:3d} K={K}: rounds 1/2/4/8 = {ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f} {ts[3]:.1f} us; per round {per:.2f} us "
          f"({2.0*16128*N*K/per/1e6:.0f} TF marginal), launch+tail {ts[1] - 2 * per:.1f} us", flush=True)
