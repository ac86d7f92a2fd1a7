import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import _lib as L
from tools.gpu_check_gemm import run, view, dev

def bench(name, fn, flops, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"TIME {name}: {dt*1e6:.1f} us  {flops/dt/1e12:.1f} TFLOP/s", flush=True)

M = 15968
for (N, K) in ((768, 768), (768, 3072), (3072, 768), (768, 1536)):
    A = torch.randn(M, K).bfloat16().to(dev); W = torch.randn(N, K).bfloat16().to(dev)
    Wt = W.t().contiguous(); At = A.t().contiguous()
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    bench(f"NT M{M} N{N} K{K}", lambda: run(A, W, Y, M, N, K, 0, 0, L.BF16), 2*M*N*K)
    ref = (A[:256].float() @ W.float().t())
    print("   maxerr", (Y[:256].float() - ref).abs().max().item(), "refmax", ref.abs().max().item())
    bench(f"NN(b_rc) M{M} N{N} K{K}", lambda: run(A, Wt, Y, M, N, K, 0, 1, L.BF16), 2*M*N*K)
    bench(f"TN(a_rc,b kc) M{M} N{N} K{K}", lambda: run(At, W, Y, M, N, K, 1, 0, L.BF16), 2*M*N*K)
