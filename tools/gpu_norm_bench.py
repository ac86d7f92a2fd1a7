"""Time the LayerNorm forward / fused backward on the model's shapes and check the backward against torch autograd.
   python tools/gpu_norm_bench.py            (SMX_LIB=... for A/B builds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from tools.gpu_check_pp import bench
dev = torch.device("cuda:0")
for M, D, res, act in ((15968, 768, True, 0), (7968, 768, True, 0), (1024, 768, True, 0), (15968, 1024, True, 0), (511968, 512, False, 0), (511968, 512, False, 1), (255968, 512, False, 1)):
    g = torch.Generator(device="cpu").manual_seed(0)
    x32 = torch.randn(M, D, generator=g)
    x = x32.to(dev).bfloat16(); dy = torch.randn(M, D, generator=g).to(dev).bfloat16()
    dres = torch.randn(M, D, generator=g).to(dev).bfloat16() if res else None
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev); beta = (0.1 * torch.randn(D, generator=g)).to(dev)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    tf = bench(lambda: ops.norm_fwd(x, y, gamma, beta, mean, rstd, M, D, ops.BF16, act=act), n=30)
    folds = ops.FoldQueue()
    def bw():
        ops.norm_bwd(dy, x, dx, gamma, beta, mean, rstd, dg, db, M, D, ops.BF16, dres=dres, folds=folds, act=act)
        folds.items.clear()
    tb = bench(bw, n=30)
    t3 = float("nan")
    if not act and res:          # the post-LN fused site: dx x dropout mask of the producing Linear + its bias-gradient partial row
        dxd = torch.empty_like(x); gb2 = torch.zeros(D, device=dev)
        def bw3():
            ops.norm_bwd(dy, x, dx, gamma, beta, mean, rstd, dg, db, M, D, ops.BF16, dres=dres, folds=folds, drop2=(0.1, 7), dx_drop=dxd, gb2=gb2)
            folds.items.clear()
        t3 = bench(bw3, n=30)
    # reference
    dg.zero_(); db.zero_()
    ops.norm_bwd(dy, x, dx, gamma, beta, mean, rstd, dg, db, M, D, ops.BF16, dres=dres, folds=folds, act=act); folds.flush()
    torch.cuda.synchronize()
    n = min(M, 4096)
    xr = x[:n].float().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gamma, beta, 1e-5)
    if act:
        yr = torch.nn.functional.gelu(yr)
    yr.backward(dy[:n].float())
    ref = xr.grad + (dres[:n].float() if res else 0)
    err = (dx[:n].float() - ref).abs().max().item() / ref.abs().max().item()
    xa = x.float(); xh = (xa - xa.mean(1, keepdim=True)) * torch.rsqrt(xa.var(1, unbiased=False, keepdim=True) + 1e-5)
    eg = eb = float("nan")
    if not act:
        eg = ((dg - (dy.float() * xh).sum(0)).abs().max() / dg.abs().max()).item()
        eb = ((db - dy.float().sum(0)).abs().max() / db.abs().max()).item()
    by_f = M * D * 2 * 2; by_b = M * D * 2 * (4 if res else 3)
    print(f"M={M} D={D} dres={res} act={act}: fwd {tf:.1f} us ({by_f / tf / 1e6:.2f} TB/s) | bwd {tb:.1f} us ({by_b / tb / 1e6:.2f} TB/s) | "
          f"with masked-dx output {t3:.1f} us ({M * D * 2 * 5 / t3 / 1e6 if t3 == t3 else 0:.2f} TB/s) | dx rel err {err:.2e} dgamma {eg:.2e} dbeta {eb:.2e}", flush=True)
