"""Mid-size GEMMs of the LM text encoder (M = 7 968 rows: fewer output tiles than the chip has workgroup slots): the
128x128 kernel plain / with a K split + split-K epilogue, the ping-pong kernel with K splits; forward and data-gradient
layouts.  Event timing, 20 launches each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view
from tools.gpu_check_pp import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(7968, 768, 3072), (7968, 768, 2304), (7968, 768, 768), (7968, 3072, 768), (7968, 2304, 768), (7968, 1536, 768),
          (15968, 768, 768), (15968, 768, 3072)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Wt = W.t().contiguous()
    R = torch.randn(M, N, device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    slabs = torch.empty(8 * M * N, dtype=torch.float32, device=dev)
    for name, kw in (("fwd ", dict()), ("dgrd", dict(b_rc=True, bv=view(N)))):
        Bm = W if not kw else Wt
        res = []
        t = bench(lambda: ops.gemm(A, Bm, Y, M, N, K, ops.BF16, bias=bias, resid=R, tr_mode=1, **kw), n=20)
        res.append(("128", t))
        for sp in (2, 3, 4):
            kst = (K + 63) // 64
            per = (kst + sp - 1) // sp
            if (sp - 1) * per >= kst:
                continue
            t = bench(lambda: ops.gemm_splitk(A, Bm, Y, M, N, K, ops.BF16, sp, slabs, bias=bias, resid=R, tr_mode=1, **kw), n=20)
            res.append((f"128/s{sp}", t))
        try:
            t = bench(lambda: ops.gemm(A, Bm, Y, M, N, K, ops.BF16, bias=bias, resid=R, tr_mode=8, **kw), n=20)
            res.append(("pp", t))
            for sp in (2, 3):
                t = bench(lambda: ops.gemm_splitk(A, Bm, Y, M, N, K, ops.BF16, sp, slabs, bias=bias, resid=R, tr_mode=8, **kw), n=20)
                res.append((f"pp/s{sp}", t))
        except RuntimeError:
            pass
        print(f"{name} M={M} N={N} K={K}: " + " | ".join(f"{n} {t:.0f}us {fl / t / 1e6:.0f}TF" for n, t in res), flush=True)
