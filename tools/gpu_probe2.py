import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import _lib as L
from tools.gpu_check_gemm import run, view, dev
def bench(name, fn, flops, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"TIME {name}: {dt*1e6:.1f} us  {flops/dt/1e12:.1f} TFLOP/s", flush=True)
M = 15968
x = torch.randn(8192, 8192, device=dev)
for _ in range(20): y = x @ x   # warm clocks
for K in (768,):
    for N in (768, 1536, 2304, 3072, 3200, 4096):
        A = torch.randn(M, K).bfloat16().to(dev); Wt = torch.randn(K, N).bfloat16().to(dev)
        Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        bench(f"NN(b_rc) M{M} N{N} K{K} dma", lambda: run(A, Wt, Y, M, N, K, 0, 1, L.BF16), 2*M*N*K)
        bench(f"NN(b_rc) M{M} N{N} K{K} old", lambda: run(A, Wt, Y, M, N, K, 0, 1, L.BF16, tr_mode=2), 2*M*N*K)
