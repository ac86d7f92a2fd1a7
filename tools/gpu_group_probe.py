"""The encoder layer's grouped weight gradient (four problems, 108 tiles, 2 K slices = 216 items of 125 K tiles): ping-pong (mode 8)
against free-running (mode 12) group kernels, and the same work as ONE problem of the same tile count (2304 x 3072) on the
single-problem kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view
dev = torch.device("cuda:0")
torch.manual_seed(0)
Mred = 15968


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
probs = []
for (No, Ko) in shapes:
    dY = torch.randn(Mred, No, device=dev).bfloat16(); X = torch.randn(Mred, Ko, device=dev).bfloat16()
    S = torch.zeros(2, No, Ko, dtype=torch.float32, device=dev)
    probs.append((dY, X, S, No, Ko, Mred, dict(a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True, atomic=0, split_k=2, split_stride=No * Ko)))
for order in (probs, probs[::-1], [probs[2], probs[0], probs[3], probs[1]]):
    for mode in (8, 12):
        t = timeit(lambda: ops.gemm_group(order, ops.BF16, mode=mode))
        print(f"group {[p[3:5] for p in order]} mode {mode}: {t:.1f} us", flush=True)
No, Ko = 2304, 3072
dY = torch.randn(Mred, No, device=dev).bfloat16(); X = torch.randn(Mred, Ko, device=dev).bfloat16()
S = torch.zeros(2, No, Ko, dtype=torch.float32, device=dev)
for mode in (8, 12):
    t = timeit(lambda: ops.gemm(dY, X, S, No, Ko, Mred, ops.BF16, a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True, split_k=2, split_stride=No * Ko, tr_mode=mode))
    print(f"single 2304 x 3072 (108 tiles) mode {mode}: {t:.1f} us", flush=True)
    one = [(dY, X, S, No, Ko, Mred, dict(a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True, atomic=0, split_k=2, split_stride=No * Ko))]
    t = timeit(lambda: ops.gemm_group(one, ops.BF16, mode=mode))
    print(f"the same problem through the group kernel, mode {mode}: {t:.1f} us", flush=True)
