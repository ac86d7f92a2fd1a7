"""GPU bring-up check for norm + attention kernels."""
import ctypes as C, sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import _lib as L
from tools.gpu_check_gemm import check

dev = torch.device("cuda:0")
lib = L.lib()
stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: t.data_ptr() if t is not None else None


def gelu(x): return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def norm_case(dtype, tdt, tol, M, D, rms, act, with_pos):
    ok = True
    x = torch.randn(M, D).to(tdt); gamma = torch.randn(D); beta = None if rms else torch.randn(D)
    S = 5
    pos = torch.randn(S + 2, D).to(tdt) if with_pos else None
    xd = x.to(dev); gd = gamma.to(dev); bd = beta.to(dev) if beta is not None else None
    posd = pos.to(dev) if pos is not None else None
    y = torch.empty_like(xd); xs = torch.empty_like(xd) if with_pos else None
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    p = L.NormParams(ptr(xd), ptr(posd), ptr(xs), ptr(y), ptr(gd), ptr(bd), ptr(mean), ptr(rstd), M, D, S, 2, rms, act, 1e-5)
    assert lib.smx_norm_fwd(C.byref(p), dtype, stream()) == 0
    xr = x.float().clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True) if beta is not None else None
    posr = pos.float().clone().requires_grad_(True) if pos is not None else None
    xin = xr
    if pos is not None:
        idx = (torch.arange(M) % S) + 2
        xin = (xr + posr[idx])
        if tdt == torch.bfloat16: xin = xin + (xin.detach().bfloat16().float() - xin.detach())
    if rms:
        yr = gr * (xin * torch.rsqrt(xin.pow(2).mean(-1, keepdim=True) + 1e-5))
    else:
        yr = torch.nn.functional.layer_norm(xin, (D,), gr, br, 1e-5)
    if act == 1: yr = gelu(yr)
    nm = f"norm dt{dtype} M{M} D{D} rms{rms} act{act} pos{int(with_pos)}"
    ok &= check(nm + " fwd", y, yr.detach(), tol)
    dy = torch.randn(M, D).to(tdt); dres = torch.randn(M, D).to(tdt)
    yr.backward(dy.float())
    dx = torch.empty_like(xd); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    dpos = torch.zeros(S + 2, D, device=dev) if with_pos else None
    dyd = dy.to(dev); dresd = dres.to(dev)
    pb = L.NormBwdParams(ptr(dyd), ptr(xs if with_pos else xd), ptr(dresd), ptr(dx), ptr(gd), ptr(bd), ptr(mean), ptr(rstd),
                         ptr(dg), ptr(db) if not rms else None, ptr(dpos), ptr(torch.empty(4 << 20, device=dev)), M, D, S, 2, rms, act)
    assert lib.smx_norm_bwd(C.byref(pb), dtype, stream()) == 0
    ok &= check(nm + " dx", dx, xr.grad + dres.float(), tol * 4)
    ok &= check(nm + " dgamma", dg, gr.grad, tol * 4)
    if not rms: ok &= check(nm + " dbeta", db, br.grad, tol * 4)
    if with_pos: ok &= check(nm + " dpos", dpos, posr.grad, tol * 4)
    return ok


def attn_case(dtype, tdt, tol, B, H, Tq, Tk, D, causal, with_bias, scale, verify=True):
    ok = True
    HD = H * D
    # Q lives in a fused [B,Tq,3*HD] buffer to exercise strides
    qkv = torch.randn(B, Tq, 3 * HD).to(tdt); kv = torch.randn(B, Tk, 2 * HD).to(tdt)
    bias = torch.randn(H, Tq, Tk) if with_bias else None
    qd = qkv.to(dev); kvd = kv.to(dev); bd = bias.to(dev) if with_bias else None
    O = torch.zeros(B, Tq, HD, dtype=tdt, device=dev); lse = torch.zeros(B, H, Tq, device=dev)
    p = L.AttnParams()
    es = qd.element_size()
    p.Q, p.K, p.V, p.O, p.lse, p.bias = qd.data_ptr() + HD * es, kvd.data_ptr(), kvd.data_ptr() + HD * es, O.data_ptr(), lse.data_ptr(), ptr(bd)
    p.q_bs, p.q_ld, p.k_bs, p.k_ld, p.v_bs, p.v_ld, p.o_bs, p.o_ld = Tq * 3 * HD, 3 * HD, Tk * 2 * HD, 2 * HD, Tk * 2 * HD, 2 * HD, Tq * HD, HD
    p.B, p.H, p.Tq, p.Tk, p.D, p.causal, p.scale = B, H, Tq, Tk, D, causal, scale
    assert lib.smx_attention_fwd(C.byref(p), dtype, stream()) == 0
    dO = torch.randn(B, Tq, HD).to(tdt)
    dOd = dO.to(dev); dQ = torch.zeros(B, Tq, HD, dtype=tdt, device=dev)
    dKV = torch.zeros(B, Tk, 2 * HD, dtype=tdt, device=dev); delta = torch.zeros(B, H, Tq, device=dev)
    p.dO, p.dQ, p.dK, p.dV, p.delta = dOd.data_ptr(), dQ.data_ptr(), dKV.data_ptr(), dKV.data_ptr() + HD * es, delta.data_ptr()
    p.do_bs, p.do_ld, p.dq_bs, p.dq_ld, p.dk_bs, p.dk_ld, p.dv_bs, p.dv_ld = Tq * HD, HD, Tq * HD, HD, Tk * 2 * HD, 2 * HD, Tk * 2 * HD, 2 * HD
    assert lib.smx_attention_bwd(C.byref(p), dtype, stream()) == 0
    keep = (qd, kvd, O, lse, dOd, dQ, dKV, delta, bd)
    if not verify:
        return True, p, keep
    q = qkv[:, :, HD:2 * HD].float().clone().requires_grad_(True)
    k = kv[:, :, :HD].float().clone().requires_grad_(True); v = kv[:, :, HD:].float().clone().requires_grad_(True)
    qh = q.view(B, Tq, H, D).transpose(1, 2); kh = k.view(B, Tk, H, D).transpose(1, 2); vh = v.view(B, Tk, H, D).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if with_bias: s = s + bias[None]
    if causal:
        msk = torch.ones(Tq, Tk, dtype=torch.bool).tril(diagonal=Tk - Tq)
        s = s.masked_fill(~msk, float("-inf"))
    pr = torch.softmax(s, -1)
    o = (pr @ vh).transpose(1, 2).reshape(B, Tq, HD)
    nm = f"attn dt{dtype} B{B} H{H} Tq{Tq} Tk{Tk} D{D} c{causal} b{int(with_bias)}"
    ok &= check(nm + " O", O, o.detach(), tol)
    ok &= check(nm + " lse", lse, torch.logsumexp(s, -1).detach(), tol)
    o.backward(dO.float())
    ok &= check(nm + " dQ", dQ, q.grad, tol * 2)
    ok &= check(nm + " dK", dKV[:, :, :HD], k.grad, tol * 2)
    ok &= check(nm + " dV", dKV[:, :, HD:], v.grad, tol * 2)
    return ok, p, keep


def main():
    torch.manual_seed(0)
    allok = True
    for dtype, tdt, tol in ((L.F32, torch.float32, 3e-5), (L.BF16, torch.bfloat16, 2e-2)):
        for (M, D) in ((37, 64), (130, 768), (9, 1024), (64, 512), (5, 32)):
            for rms, act, pos in ((0, 0, False), (0, 1, False), (1, 0, False), (0, 0, True)):
                allok &= norm_case(dtype, tdt, tol, M, D, rms, act, pos)
    for (B, H, Tq, Tk, causal, bias) in ((2, 3, 70, 70, 0, False), (2, 2, 33, 33, 1, False), (1, 4, 6, 131, 0, False),
                                          (2, 2, 40, 40, 1, True), (1, 2, 150, 150, 0, True), (1, 1, 64, 64, 0, False),
                                          (1, 12, 499, 499, 0, False)):
        ok, _, _ = attn_case(L.BF16, torch.bfloat16, 2e-2, B, H, Tq, Tk, 64, causal, bias, 0.125)
        allok &= ok
    for (B, H, Tq, Tk, D, causal, bias) in ((2, 4, 24, 24, 16, 0, False), (2, 4, 6, 6, 16, 1, False), (2, 4, 6, 12, 16, 0, True),
                                             (1, 2, 70, 70, 64, 1, True)):
        ok, _, _ = attn_case(L.F32, torch.float32, 3e-5, B, H, Tq, Tk, D, causal, bias, 1.0 / math.sqrt(D))
        allok &= ok
    # timing: encoder self-attention at config-2 size (B=32, H=12, T=499)
    ok, p, keep = attn_case(L.BF16, torch.bfloat16, 2e-2, 32, 12, 499, 499, 64, 0, False, 0.125, verify=False)
    def tm(name, fn, flops):
        for _ in range(30): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"TIME {name}: {dt*1e6:.1f} us {flops/dt/1e12:.1f} TFLOP/s")
    fl = 4 * 32 * 12 * 499 * 499 * 64
    tm("attn fwd B32 H12 T499", lambda: lib.smx_attention_fwd(C.byref(p), L.BF16, stream()), fl)
    tm("attn bwd B32 H12 T499", lambda: lib.smx_attention_bwd(C.byref(p), L.BF16, stream()), fl * 2.5)
    M, D = 15968, 768
    x = torch.randn(M, D, device=dev).bfloat16(); y = torch.empty_like(x); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev); dx = torch.empty_like(x); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    ws = torch.empty(4 << 20, device=dev)
    pn = L.NormParams(ptr(x), None, None, ptr(y), ptr(g), ptr(b), ptr(mean), ptr(rstd), M, D, 0, 0, 0, 0, 1e-5)
    pb = L.NormBwdParams(ptr(y), ptr(x), None, ptr(dx), ptr(g), ptr(b), ptr(mean), ptr(rstd), ptr(dg), ptr(db), None, ptr(ws), M, D, 0, 0, 0, 0)
    for name, fn, byts in (("ln fwd 15968x768", lambda: lib.smx_norm_fwd(C.byref(pn), L.BF16, stream()), 2 * M * D * 2),
                           ("ln bwd 15968x768", lambda: lib.smx_norm_bwd(C.byref(pb), L.BF16, stream()), 3 * M * D * 2)):
        for _ in range(200): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"TIME {name}: {dt*1e6:.1f} us {byts/dt/1e9:.0f} GB/s")
    print("ALL OK" if allok else "SOME FAILED")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
