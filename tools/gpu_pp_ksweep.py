"""time(K) = a + b K for the ping-pong GEMM at one full round of tiles: separates per-item overhead from per-K-tile cost.
LAB flags (tr_mode >> 8): 1 no DMA, 2 no LDS reads, 4 no MFMA, 8 lgkm wait before barrier, 16 no epilogue, 32 one barrier."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from tools.gpu_check_pp import bench
dev = torch.device("cuda:0")
M, N = 16128, 1024
labs = [int(x) for x in sys.argv[1:]] or [0]
for tr in [8 + (l << 8) for l in labs] + [1]:
    ts = {}
    for K in (256, 1024, 4096):
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ts[K] = bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, tr_mode=tr), n=20)
    slope = (ts[4096] - ts[1024]) / 48
    print(f"tr{tr & 255} lab{tr >> 8:3d}: K=256 {ts[256]:6.1f}  K=1024 {ts[1024]:6.1f}  K=4096 {ts[4096]:6.1f} us ({2.0*M*N*4096/ts[4096]/1e6:5.0f} TF)"
          f"  slope {slope:.3f} us/Ktile  fixed {ts[1024] - 16 * slope:5.1f} us", flush=True)
