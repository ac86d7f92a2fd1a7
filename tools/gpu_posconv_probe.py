"""Positional-conv GEMM shapes, plain vs time-blocked.  The grouped conv (16 groups x 48 channels, k = 128) is a batched
GEMM with N = 48 outputs per group: 62 % of a 128-wide tile computes nothing.  Blocking J consecutive frames into one GEMM row
(row = frames J t' .. J t' + J - 1 of one clip, A = the (k + J - 1) x 48 inputs they see, B = the taps shifted J times) gives
N = 48 J at (k + J - 1) / k of the flops.  Times the forward-layout launch for several J and kernels.
    python tools/gpu_posconv_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view

dev = torch.device("cuda:0")
B, T, G, Cg, K = 32, 499, 16, 48, 128


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for J in (1, 4, 8, 16):
    Tb = (T + J - 1) // J
    Kp = (K + J - 1) * Cg
    Tp = J * Tb + K - 1
    xg = (torch.randn(G * B * Tp * Cg, device=dev) * 0.5).bfloat16()
    w = (torch.randn(G * J * Cg * Kp, device=dev) * 0.02).bfloat16()
    out = torch.empty(G * B * Tb * J * Cg, dtype=torch.bfloat16, device=dev)
    flops = 2.0 * B * T * G * Cg * K * Cg
    for mode in (1, 11, 8, 12, 13):
        def f():
            ops.gemm(xg, w, out, B * Tb, J * Cg, Kp, ops.BF16, av=view(J * Cg, Tb, Tp * Cg), bv=view(Kp), cv=view(J * Cg), nbatch=G,
                     batch_a=B * Tp * Cg, batch_b=J * Cg * Kp, batch_c=B * Tb * J * Cg, tr_mode=mode)
        try:
            us = timeit(f)
            print(f"J={J:2d} M={B * Tb:6d} N={J * Cg:4d} K={Kp:5d} mode {mode:2d}: {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s (useful flops)", flush=True)
        except RuntimeError as e:
            print(f"J={J:2d} mode {mode}: {str(e)[:60]}")
    # weight-gradient layout: out[J Cg, Kp] += dy^T x over the B Tb rows (rows-contiguous operands through batched views)
    dy = (torch.randn(G * B * (J * Tb) * Cg, device=dev) * 0.5).bfloat16()
    n = J * Cg * Kp
    for mode, split in ((1, 4), (8, 1), (8, 2), (8, 4)):
        slabs = torch.empty(G * split * n, dtype=torch.float32, device=dev)

        def g():
            ops.gemm(dy, xg, slabs, J * Cg, Kp, B * Tb, ops.BF16, a_rc=True, b_rc=True, av=view(J * Cg, Tb, J * Tb * Cg),
                     bv=view(J * Cg, Tb, Tp * Cg), cv=view(Kp), out_f32=True, atomic=0, split_k=split, split_stride=G * n if split > 1 else 0,
                     nbatch=G, batch_a=B * J * Tb * Cg, batch_b=B * Tp * Cg, batch_c=n, tr_mode=mode)
        try:
            us = timeit(g)
            print(f"J={J:2d} wgrad [{J * Cg} x {Kp}] over {B * Tb} rows, mode {mode} split {split}: {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s", flush=True)
        except RuntimeError as e:
            print(f"J={J:2d} wgrad mode {mode} split {split}: {str(e)[:60]}")
