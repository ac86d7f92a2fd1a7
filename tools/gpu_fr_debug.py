"""Localise mismatches of a GEMM variant (SMX_DEBUG_TR, default 12) against tr_mode 1 on rows-contiguous operands: per 128x128 block error map."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view
dev = torch.device("cuda:0")
TR = int(os.environ.get("SMX_DEBUG_TR", "12"))
torch.manual_seed(0)
for (No, Ko, Mred, split) in [(200, 136, 300, 1), (512, 256, 300, 1), (256, 512, 300, 1), (512, 512, 64, 1), (512, 512, 128, 1), (512, 512, 640, 1),
                              (200, 136, 300, 5), (200, 136, 128, 2), (768, 768, 4096, 1)]:
    Yb = torch.randn(Mred, No, device=dev).bfloat16()
    A = torch.randn(Mred, Ko, device=dev).bfloat16()
    outs = []
    for t in (1, TR):
        S = torch.zeros(split, No, Ko, dtype=torch.float32, device=dev)
        ops.gemm(Yb, A, S, No, Ko, Mred, ops.BF16, a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True,
                 split_k=split, split_stride=No * Ko if split > 1 else 0, tr_mode=t)
        outs.append(S)
    torch.cuda.synchronize()
    ref = (Yb.float().t() @ A.float())
    for s in range(split):
        d = (outs[1][s] - outs[0][s]).abs()
        print(f"wgrad {No}x{Ko}x{Mred} split{split} slice {s}: max diff {d.max().item():.3e} (ref scale {ref.abs().max().item():.1f})")
        if d.max().item() > 1e-2:
            nb, kb = (No + 63) // 64, (Ko + 63) // 64
            for i in range(nb):
                print("   ", " ".join(f"{d[i*64:(i+1)*64, j*64:(j+1)*64].max().item():8.1e}" for j in range(kb)))
