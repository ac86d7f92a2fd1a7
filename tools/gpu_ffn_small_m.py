"""fp32-path FFN forward / backward at a tiny row count (M = 24, d = 1024, F = 4096: config 4's LM encoder on 2 x 2 s clips) vs torch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import view, ACT_GELU
dev = torch.device("cuda:0")
torch.manual_seed(0)
for M in (24, 148, 1000):
    d, F = 1024, 4096
    X = torch.randn(M, d, device=dev); W1 = torch.randn(F, d, device=dev) * 0.03; b1 = torch.randn(F, device=dev) * 0.1
    dH = torch.randn(M, F, device=dev) * 0.01
    for ld in (F, F + 64):
        pre = torch.zeros(M, ld, device=dev); h = torch.zeros(M, ld, device=dev)
        ops.gemm(X, W1, h, M, F, d, ops.F32, bias=b1, act=ACT_GELU, aux_out=pre, cv=view(ld))
        pr = X @ W1.t() + b1
        e_pre = (pre[:, :F] - pr).abs().max().item(); e_h = (h[:, :F] - torch.nn.functional.gelu(pr)).abs().max().item()
        # backward through the activation: dpre = dH * gelu'(pre), fused into a GEMM epilogue in the model (aux_in); here via act_bwd
        dpre = torch.zeros(M, F, device=dev)
        ops.act_bwd(dH, pre[:, :F].contiguous(), dpre, M, F, view(F), ACT_GELU, ops.F32)
        prr = pr.clone().requires_grad_(True)
        torch.nn.functional.gelu(prr).backward(dH)
        e_d = (dpre - prr.grad).abs().max().item()
        # dgrad GEMM with aux_in epilogue: dX2 = (dY W2) * gelu'(pre)
        W2 = torch.randn(d, F, device=dev) * 0.03; dY = torch.randn(M, d, device=dev) * 0.01
        out = torch.zeros(M, ld, device=dev)
        ops.gemm(dY, W2, out, M, F, d, ops.F32, b_rc=True, bv=view(F), aux_in=pre, act=ACT_GELU, cv=view(ld), ev=view(ld))
        ref = (dY @ W2) * (prr.grad / dH)
        e_g = (out[:, :F] - ref).abs().max().item()
        bad = ((out[:, :F] - ref).abs() > 1e-5).nonzero()
        print(f"M={M} ld={ld}: pre {e_pre:.2e} gelu {e_h:.2e} act_bwd {e_d:.2e} dgrad+act' {e_g:.2e} (scale {ref.abs().max().item():.2e}) bad {bad[:4].tolist()}", flush=True)
