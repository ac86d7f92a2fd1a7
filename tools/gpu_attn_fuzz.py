"""Randomised screen of the attention kernels (MFMA bf16 head_dim 64 and the fp32 path) with per-clip key lengths: random B, H,
Tq, Tk, causal, klen against fp32 torch (forward O, backward dQ / dK / dV; padded keys must get exactly zero gradient).
   python tools/gpu_attn_fuzz.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"bf16": 0.0, "fp32": 0.0}
bad = 0
for case in range(cases):
    dtype = rng.choice(["bf16", "bf16", "fp32"])
    D = 64
    dt, tdt, tol = (ops.BF16, torch.bfloat16, 3e-2) if dtype == "bf16" else (ops.F32, torch.float32, 3e-4)
    B, H = rng.randrange(1, 5), rng.randrange(1, 4)
    causal = rng.random() < 0.3
    Tk = rng.choice([1, 7, 63, 64, 65, 127, 128, 129, 200, 249, 499, 512, 600])
    Tq = Tk if causal and rng.random() < 0.7 else (rng.randrange(1, Tk + 1) if causal else rng.choice([1, 5, 32, 64, 100, 249, 499]))
    use_klen = rng.random() < 0.8
    lens = [rng.randrange(1, Tk + 1) if rng.random() < 0.7 else Tk for _ in range(B)] if use_klen else [Tk] * B
    HD = H * D
    g = torch.Generator().manual_seed(case)
    q = torch.randn(B, Tq, HD, generator=g).to(tdt); k = torch.randn(B, Tk, HD, generator=g).to(tdt); v = torch.randn(B, Tk, HD, generator=g).to(tdt)
    do = torch.randn(B, Tq, HD, generator=g).to(tdt)
    qd, kd, vd, dod = q.to(dev), k.to(dev), v.to(dev), do.to(dev)
    o = torch.zeros(B, Tq, HD, dtype=tdt, device=dev); lse = torch.zeros(B * H * Tq, device=dev); delta = torch.zeros_like(lse)
    dq, dk, dv = torch.zeros_like(qd), torch.full_like(kd, 7.0), torch.full_like(vd, 7.0)
    klen = torch.tensor(lens, dtype=torch.int32, device=dev) if use_klen else None
    desc = ops.AttnDesc(B, H, Tq, Tk, D, causal, D ** -0.5, klen=klen)
    for name, t, n in (("Q", qd, Tq), ("K", kd, Tk), ("V", vd, Tk), ("O", o, Tq), ("dO", dod, Tq), ("dQ", dq, Tq), ("dK", dk, Tk), ("dV", dv, Tk)):
        desc.set(name, t, 0, n * HD, HD)
    ops.attention_fwd(desc, lse, dt)
    ops.attention_bwd(desc, lse, delta, dt)
    torch.cuda.synchronize()
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(B, -1, H, D).transpose(1, 2) for t in (qf, kf, vf))
    s = qh @ kh.transpose(-1, -2) * D ** -0.5
    pad = torch.arange(Tk)[None, :] >= torch.tensor(lens)[:, None]
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    if causal:
        s = s.masked_fill(~torch.ones(Tq, Tk, dtype=torch.bool).tril(diagonal=Tk - Tq), float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Tq, HD)
    ref.backward(do.float())
    msg = f"case {case} {dtype} B{B} H{H} Tq{Tq} Tk{Tk} causal{int(causal)} lens{lens if use_klen else None}"
    ok = True
    for b, n in enumerate(lens):
        if n < Tk and (dk[b, n:].abs().max().item() != 0.0 or dv[b, n:].abs().max().item() != 0.0):
            print("NONZERO gradient on padded keys:", msg); ok = False
    for name, got, want in (("O", o, ref.detach()), ("dQ", dq, qf.grad), ("dK", dk, kf.grad), ("dV", dv, vf.grad)):
        e = (got.float().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-1)     # (floor: with one visible key dQ = dK = 0 exactly)
        worst[dtype] = max(worst[dtype], e)
        if not (e < tol):
            print(f"MISMATCH {name} rel err {e:.3e}:", msg); ok = False
    bad += 0 if ok else 1
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} bad, worst rel err bf16 {worst['bf16']:.2e} fp32 {worst['fp32']:.2e}", flush=True)
print("TOTAL bad", bad, worst)
