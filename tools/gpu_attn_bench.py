"""Attention microbench at the encoder shape (B=32, H=12, T=499, D=64) + correctness vs torch (fp32 math)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
dev = torch.device("cuda:0")


def bench(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(B, H, T, D, causal, drop):
    d = H * D
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = (torch.randn(B * T, 3 * d, generator=g) * 0.7).to(dev, torch.bfloat16)
    do = torch.randn(B * T, d, generator=g).to(dev, torch.bfloat16)
    desc = ops.AttnDesc(B, H, T, T, D, causal, D ** -0.5, drop=drop)
    desc.set("Q", qkv, 0, T * 3 * d, 3 * d); desc.set("K", qkv, d, T * 3 * d, 3 * d); desc.set("V", qkv, 2 * d, T * 3 * d, 3 * d)
    o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H * T, device=dev)
    desc.set("O", o, 0, T * d, d)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * H * T, device=dev)
    desc.set("dO", do, 0, T * d, d)
    desc.set("dQ", dqkv, 0, T * 3 * d, 3 * d); desc.set("dK", dqkv, d, T * 3 * d, 3 * d); desc.set("dV", dqkv, 2 * d, T * 3 * d, 3 * d)
    tf = bench(lambda: ops.attention_fwd(desc, lse, ops.BF16))
    tb = bench(lambda: ops.attention_bwd(desc, lse, delta, ops.BF16))
    fl = 4.0 * B * H * T * T * D * (0.5 if causal else 1.0)
    # reference (no dropout only)
    err = ""
    if drop is None:
        x = qkv.float().view(B, T, 3, H, D).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)
        s = (x[0] @ x[1].transpose(-1, -2)) * D ** -0.5
        if causal:
            s = s.masked_fill(torch.ones(T, T, device=dev).triu(1).bool(), float("-inf"))
        ref = (torch.softmax(s, -1) @ x[2]).permute(0, 2, 1, 3).reshape(B * T, d)
        ref.backward(do.float())
        dref = x.grad.permute(1, 3, 0, 2, 4).reshape(B * T, 3 * d)
        err = f" err_o {(o.float() - ref.detach()).abs().max().item():.3e} err_dqkv {(dqkv.float() - dref).abs().max().item():.3e}"
    print(f"B={B} H={H} T={T} D={D} causal={causal} drop={drop}: fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF)  bwd {tb:.1f} us ({2.5 * fl / tb / 1e6:.0f} TF){err}", flush=True)


run(32, 12, 499, 64, False, None)
run(32, 12, 499, 64, False, (0.1, 1234))
run(32, 12, 249, 64, False, None)
run(32, 12, 32, 64, True, None)
run(4, 12, 499, 64, False, None)
