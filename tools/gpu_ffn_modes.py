"""The two FFN GEMMs with heavy epilogues at the encoder's shape (15 968 x 3072 x 768): forward with GELU + second output, data
gradient with the activation-derivative side input - every kernel variant that has the class (non-saved forms: all kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import ACT_GELU, view
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 15968, 3072, 768
A = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev) * 0.1
Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
aux = torch.empty_like(Y)
dY = torch.randn(M, K, device=dev).bfloat16()        # dgrad of FFN2: dH[M, 3072] = dY[M, 768] @ W2[768, 3072]
W2 = (torch.randn(K, N, device=dev) * 0.05).bfloat16()


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = {
    "fwd act + pre-activation copy": lambda m: ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, act=ACT_GELU, aux_out=aux, tr_mode=m),
    "fwd act, saved derivative": lambda m: ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=aux, drop=(0.1, 3), tr_mode=m),
    "fwd linear": lambda m: ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, tr_mode=m),
    "dgrad x act'(aux)": lambda m: ops.gemm(dY, W2, Y, M, N, K, ops.BF16, b_rc=True, bv=view(N), aux_in=aux, act=ACT_GELU, tr_mode=m),
    "dgrad, saved derivative": lambda m: ops.gemm(dY, W2, Y, M, N, K, ops.BF16, b_rc=True, bv=view(N), aux_in=aux, act=ACT_GELU | ops.ACT_SAVE_GRAD, tr_mode=m),
    "dgrad linear": lambda m: ops.gemm(dY, W2, Y, M, N, K, ops.BF16, b_rc=True, bv=view(N), tr_mode=m),
}
for name, fn in cases.items():
    row = f"{name:32s}"
    for m in (1, 11, 8, 12, 13):
        try:
            t = timeit(lambda: fn(m))
            row += f"  m{m} {t:6.1f} us ({2e-6 * M * N * K / t:4.0f} TF/s)"
        except RuntimeError:
            row += f"  m{m}      -            "
    print(row, flush=True)
