"""Reference point, not a dependency: what the vendor library (hipBLASLt through torch.matmul) reaches on the model's main
GEMM shapes, next to this repo's kernels (plain epilogue, bf16, fp32 accumulate).  Event timing, 20 launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from tools.gpu_check_pp import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(15968, 3072, 768), (15968, 768, 3072), (15968, 2304, 768), (15968, 768, 768), (7968, 768, 3072),
                  (511968, 512, 1536), (1024, 768, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Wt = W.t().contiguous()
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    t_nt = bench(lambda: torch.matmul(A, W.t(), out=Y), n=20)          # y = x W^T (forward layout)
    t_nn = bench(lambda: torch.matmul(A, Wt, out=Y), n=20)             # W stored [K, N]
    t1 = bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, tr_mode=1), n=20)
    try:
        t8 = bench(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, tr_mode=8), n=20)
    except RuntimeError:
        t8 = float("nan")
    print(f"M={M} N={N} K={K}: vendor NT {t_nt:.0f} us {fl/t_nt/1e6:.0f} TF | vendor NN {t_nn:.0f} us {fl/t_nn/1e6:.0f} TF | "
          f"128x128 {t1:.0f} us {fl/t1/1e6:.0f} TF | ping-pong {t8:.0f} us {fl/t8/1e6:.0f} TF", flush=True)
