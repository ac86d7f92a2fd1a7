import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import _lib as L
from tools.gpu_check_gemm import run, view, dev
def bench(name, fn, flops, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"TIME {name}: {dt*1e6:.1f} us  {flops/dt/1e12:.1f} TFLOP/s", flush=True)
M = 15968
x = torch.randn(8192, 8192, device=dev)
for _ in range(20): y = x @ x   # warm clocks
for (N, K) in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    A = torch.randn(M, K).bfloat16().to(dev); W = torch.randn(N, K).bfloat16().to(dev); Wt = W.t().contiguous()
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); dY = torch.randn(M, N).bfloat16().to(dev)
    dX = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
    slabs = torch.zeros(8 * N * K, device=dev)
    for tr in (4, 7):
        bench(f"fwd   M{M} N{N} K{K} tr{tr}", lambda: run(A, W, Y, M, N, K, 0, 0, L.BF16, tr_mode=tr), 2*M*N*K)
        bench(f"dgrad M{M} N{N} K{K} tr{tr}", lambda: run(dY, W, dX, M, K, N, 0, 1, L.BF16, tr_mode=tr), 2*M*N*K)
        bench(f"wgrad M{M} N{N} K{K} split8 slabs tr{tr}", lambda: run(dY, A, slabs, N, K, M, 1, 1, L.BF16, out_f32=1, split_k=8, split_stride=N*K, tr_mode=tr), 2*M*N*K)

