"""Is the rows-contiguous weight-gradient loop bound by operand fetch?  The same launch (768 x 3072 output, 15 968 reduction rows,
7 K slices, tr_mode 8 / 12) with the operands' row stride set to 0: every k-row then reads row 0 (L1 / L2-resident), the LDS
traffic and the MFMAs are unchanged."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view
dev = torch.device("cuda:0")
torch.manual_seed(0)
No, Ko, Mred, sp = 768, 3072, 15968, 7
dY = torch.randn(Mred, No, device=dev).bfloat16()
X = torch.randn(Mred, Ko, device=dev).bfloat16()
S = torch.zeros(sp, No, Ko, dtype=torch.float32, device=dev)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for mode in (8, 12, 1):
    for name, av, bv in (("real strides", view(No), view(Ko)), ("row stride 0 (operands cache-resident)", view(0), view(0))):
        t = timeit(lambda: ops.gemm(dY, X, S, No, Ko, Mred, ops.BF16, a_rc=True, b_rc=True, av=av, bv=bv, out_f32=True, split_k=sp,
                                    split_stride=No * Ko, tr_mode=mode))
        print(f"mode {mode:2d} {name:42s}: {t:7.1f} us  ({2e-6 * No * Ko * Mred / t:5.0f} TF/s)", flush=True)
