import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view
dev = torch.device("cuda:0")
TR = int(os.environ.get("SMX_DEBUG_TR", "12"))
torch.manual_seed(0)
No, Ko, Mred, split = 200, 136, 300, 5
Yb = torch.randn(Mred, No, device=dev).bfloat16()
A = torch.randn(Mred, Ko, device=dev).bfloat16()
kw = dict(a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True, split_k=split, split_stride=No * Ko)
ref = torch.zeros(split, No, Ko, dtype=torch.float32, device=dev)
ops.gemm(Yb, A, ref, No, Ko, Mred, ops.BF16, tr_mode=1, **kw)
Yf, Af = Yb.float(), A.float()
for rep in range(6):
    S = torch.zeros_like(ref)
    ops.gemm(Yb, A, S, No, Ko, Mred, ops.BF16, tr_mode=TR, **kw)
    for s in range(split):
        d = (S[s] - ref[s]).abs()
        if d.max().item() > 1e-2:
            k0, k1 = s * 64, min(Mred, s * 64 + 64)
            rows = (d.max(dim=1).values > 1e-2).nonzero().flatten().tolist()
            cols = (d.max(dim=0).values > 1e-2).nonzero().flatten().tolist()
            print(f"rep {rep} slice {s}: bad rows {rows[0]}..{rows[-1]} ({len(rows)}), bad cols {cols[0]}..{cols[-1]} ({len(cols)})")
            r0 = rows[0]
            # hypotheses: only first / second 32-deep half step; A rows shifted; zero
            h0 = Yf[k0:k0+32].t() @ Af[k0:k0+32]
            h1 = Yf[k0+32:k1].t() @ Af[k0+32:k1]
            for name, cand in (("half0 only", h0), ("half1 only", h1), ("zero", torch.zeros_like(h0))):
                e = (S[s][rows][:, cols] - cand[rows][:, cols]).abs().max().item()
                print(f"     vs {name}: {e:.3e}")
            # which A rows (columns of Yb) would explain it: solve per-row correlation
            sub = S[s][rows][:, cols]                      # [r, c]
            # least squares: sub = X^T A_k  -> X = pinv(A_k[:, cols]^T) ...; compare with Yf[k0:k1, :]
            Ak = Af[k0:k1][:, cols]                         # [k, c]
            X = torch.linalg.lstsq(Ak.t().cpu(), sub.t().cpu()).solution   # [k, r]
            Ytrue = Yf[k0:k1].cpu()
            # for each bad row, find which true column of Y it matches best
            best = []
            for i, r in enumerate(rows[:8]):
                diffs = (Ytrue - X[:, i:i+1]).abs().max(dim=0).values
                j = int(diffs.argmin()); best.append((r, j, round(diffs[j].item(), 3)))
            print("     recovered A-operand rows (out row, matching source row, err):", best)
            break
