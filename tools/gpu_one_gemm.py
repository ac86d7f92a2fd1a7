import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import _lib as L
from tools.gpu_check_gemm import run, dev
M, N, K = 15968, 3072, 768
tr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
A = torch.randn(M, K).bfloat16().to(dev); W = torch.randn(N, K).bfloat16().to(dev)
Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(10):
    run(A, W, Y, M, N, K, 0, 0, L.BF16, tr_mode=tr)
torch.cuda.synchronize()
