"""Fixed and per-K-tile cost of a few-workgroup GEMM launch (run under rocprofv3 --kernel-trace, then tools/trace_cfgs.py):
the 128x128 (tr_mode 1) and 64x128 (tr_mode 9) kernels at K = 64 (ONE K tile), 768 and 3072 on the decoder's M = 1 024 rows,
on the text encoder's 7 968 rows and on a single 128 x 128 tile.  Round-3 numbers (profiles/r03_small_gemm_floor.txt):
5.5 / 7.0 us for one K tile, +0.49 / +0.8 us per further tile, the same per-tile cost for ONE workgroup on an idle chip -
32 KB of fills per step at the ~0.8 us a first touch of a line costs (every launch starts with cold caches) is the
40 GB/s per CU that the CU's outstanding-miss capacity allows; more stages in flight do not change it, more CUs do."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
mark = torch.zeros(64, device=dev)
cfg = 0
for (M, N, K) in ((1024, 768, 64), (1024, 768, 768), (1024, 768, 3072), (7968, 768, 768), (128, 128, 768)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev)
    for mode in (1, 9):
        torch.cuda.synchronize()
        mark.fill_(float(cfg))
        for _ in range(10):
            ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, tr_mode=mode)
        torch.cuda.synchronize()
        print(f"CFG {cfg} M={M} N={N} K={K} mode={mode}", flush=True)
        cfg += 1
