"""Where a K tile's time goes in the free-running kernel (tr_mode 12), per operand layout: a 4096 x 4096 output with K = 8192
(256 items, one per CU, 128 K tiles each: the epilogue is < 3 % of the launch).  Run once per build:
   python tools/gpu_fr_ablation.py                                      the product kernel
   SMX_LIB=tools/lab/libsmx_frlab1.so python tools/gpu_fr_ablation.py   K loop without fragment reads
   SMX_LIB=tools/lab/libsmx_frlab2.so python tools/gpu_fr_ablation.py   K loop without MFMAs
(tools/lab/build_variant.sh frlab1 gemm_fr.hip "-DSMX_FR_LAB=1", ... =2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = N = 4096
K = 8192
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
At = A.t().contiguous(); Wt = W.t().contiguous()          # [K, M], [K, N]: rows-contiguous forms
Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
G = torch.zeros(M, N, dtype=torch.float32, device=dev)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


lib = os.path.basename(os.environ.get("SMX_LIB", "product"))
# the model's weight-gradient geometry: 2304 x 3072 output (108 tiles), 15 968 reduction rows in 2 K slices = 216 items of 125 K tiles
No, Ko, Mred = 2304, 3072, 15968
dY = torch.randn(Mred, No, device=dev).bfloat16(); X = torch.randn(Mred, Ko, device=dev).bfloat16()
S = torch.zeros(2, No, Ko, dtype=torch.float32, device=dev)
# ... and its forward / data-gradient geometry: 15 968 x 768 x 3072 on 256 x 256 tiles (189 items of 48 K tiles)
Am = torch.randn(15968, 3072, device=dev).bfloat16(); Wm = (torch.randn(768, 3072, device=dev) * 0.05).bfloat16(); Wmt = Wm.t().contiguous()
Ym = torch.zeros(15968, 768, dtype=torch.bfloat16, device=dev)
for mode in (12, 8):
    t_w = timeit(lambda: ops.gemm(dY, X, S, No, Ko, Mred, ops.BF16, a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True, split_k=2,
                                  split_stride=No * Ko, tr_mode=mode))
    t_f = timeit(lambda: ops.gemm(Am, Wm, Ym, 15968, 768, 3072, ops.BF16, tr_mode=mode))
    t_d = timeit(lambda: ops.gemm(Am, Wmt, Ym, 15968, 768, 3072, ops.BF16, b_rc=True, bv=view(768), tr_mode=mode))
    print(f"{lib} mode {mode}: model shapes  wgrad RC.RC {t_w:.1f} us = {t_w / 125:.3f} us per K tile | fwd KC.KC {t_f:.1f} us = {t_f / 48:.3f} | dgrad KC.RC {t_d:.1f} us = {t_d / 48:.3f} (incl. epilogue)", flush=True)
for mode in (12, 8):
    t_kk = timeit(lambda: ops.gemm(A, W, Y, M, N, K, ops.BF16, tr_mode=mode))
    t_kr = timeit(lambda: ops.gemm(A, Wt, Y, M, N, K, ops.BF16, b_rc=True, bv=view(N), tr_mode=mode))
    t_rr = timeit(lambda: ops.gemm(At, Wt, G, M, N, K, ops.BF16, a_rc=True, b_rc=True, av=view(M), bv=view(N), out_f32=True, tr_mode=mode))
    print(f"{lib} mode {mode}: us per K tile  KC.KC {t_kk / 128:.3f}  KC.RC {t_kr / 128:.3f}  RC.RC {t_rr / 128:.3f}   (launch {t_kk:.0f} / {t_kr:.0f} / {t_rr:.0f} us)", flush=True)
