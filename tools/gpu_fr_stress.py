"""Race screen of a GEMM variant (SMX_DEBUG_TR, default 12) on rows-contiguous operands: repeated launches, count of mismatching 64x64 blocks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view
dev = torch.device("cuda:0")
TR = int(os.environ.get("SMX_DEBUG_TR", "12"))
torch.manual_seed(0)
tot = 0
for (No, Ko, Mred, split) in [(512, 512, 64, 1), (200, 136, 300, 5), (768, 768, 4096, 1), (768, 3072, 15968, 7), (512, 512, 640, 1)]:
    Yb = torch.randn(Mred, No, device=dev).bfloat16()
    A = torch.randn(Mred, Ko, device=dev).bfloat16()
    ref = torch.zeros(split, No, Ko, dtype=torch.float32, device=dev)
    kw = dict(a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True, split_k=split, split_stride=No * Ko if split > 1 else 0)
    ops.gemm(Yb, A, ref, No, Ko, Mred, ops.BF16, tr_mode=1, **kw)
    bad = {}
    for rep in range(20):
        S = torch.zeros_like(ref)
        ops.gemm(Yb, A, S, No, Ko, Mred, ops.BF16, tr_mode=TR, **kw)
        d = (S - ref).abs()
        if d.max().item() > 1e-2:
            for s in range(split):
                for i in range((No + 63) // 64):
                    for j in range((Ko + 63) // 64):
                        if d[s, i*64:(i+1)*64, j*64:(j+1)*64].max().item() > 1e-2:
                            bad[(i % 4, j % 4)] = bad.get((i % 4, j % 4), 0) + 1
    n = sum(bad.values()); tot += n
    print(f"{No}x{Ko}x{Mred} split{split}: {n} bad blocks in 20 launches; by (row block % 4, col block % 4): {sorted(bad.items())}", flush=True)
print("TOTAL", tot)
