"""tr_mode 1 vs 11 (and 8) on the large-backbone GEMM shapes, many repetitions.   python tools/gpu_w8_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view, ACT_GELU
from tools.gpu_check_pp import bench
dev = torch.device("cuda:0")
M = 15968
for (N, K, kind) in ((4096, 1024, "fwd_act"), (1024, 4096, "fwd_lin"), (3072, 1024, "fwd_lin"), (1024, 1024, "fwd_lin"),
                     (4096, 1024, "dgrad_actgrad"), (1024, 4096, "dgrad_lin"), (1024, 3072, "dgrad_lin"), (1024, 1024, "dgrad_lin")):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    Wt = W.t().contiguous()
    bias = torch.randn(N, device=dev) * 0.1
    R = torch.randn(M, N, device=dev).bfloat16(); S = torch.randn(M, N, device=dev).bfloat16()
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = torch.empty_like(Y)
    if kind == "fwd_act":
        kw = dict(bias=bias, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=aux, drop=(0.1, 3)); B = W
    elif kind == "fwd_lin":
        kw = dict(bias=bias, resid=R, drop=(0.1, 3)); B = W
    elif kind == "dgrad_actgrad":
        kw = dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU | ops.ACT_SAVE_GRAD); B = Wt
    else:
        kw = dict(b_rc=True, bv=view(N), resid=R); B = Wt
    ts = {}
    for mode in (1, 11, 8):
        try:
            ts[mode] = bench(lambda: ops.gemm(A, B, Y, M, N, K, ops.BF16, tr_mode=mode, **kw), n=30)
        except RuntimeError:
            ts[mode] = float("nan")
    fl = 2.0 * M * N * K
    print(f"{kind:14s} N={N:5d} K={K:5d}: 128x128 {ts[1]:7.1f} us ({fl/ts[1]/1e6:5.0f} TF) | 8-wave 256x128 {ts[11]:7.1f} us ({fl/ts[11]/1e6:5.0f} TF) | "
          f"ping-pong {ts[8]:7.1f} us ({fl/ts[8]/1e6:5.0f} TF)", flush=True)
