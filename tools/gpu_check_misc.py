"""GPU bring-up check for conv0 / CE / embedding / optimizer kernels against torch CPU references."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from speechmix_amd import ops
from tools.gpu_check_gemm import check

dev = torch.device("cuda:0")


def gelu(x): return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def conv0_case(dtype, tdt, tol, group, B=2, N=4000, C=32, k=10, s=5, bias=False):
    ok = True
    T0 = (N - k) // s + 1
    wave = torch.randn(B, N) * 0.3
    w = torch.randn(C, 1, k) * 0.3
    cb = torch.randn(C) * 0.1 if bias else None
    gm, bt = torch.randn(C), torch.randn(C) * 0.1
    wr = w.clone().requires_grad_(True); gr = gm.clone().requires_grad_(True); br = bt.clone().requires_grad_(True)
    cbr = cb.clone().requires_grad_(True) if bias else None
    u = F.conv1d(wave[:, None], wr, cbr, stride=s)
    if group:
        y = gelu(F.group_norm(u, C, gr, br, 1e-5))
    else:
        y = u
    yr = y.transpose(1, 2)  # [B,T0,C]
    wd, gd, bd = w.to(dev).contiguous(), gm.to(dev), bt.to(dev)
    cbd = cb.to(dev) if bias else None
    waved = wave.to(dev)
    yd = torch.zeros(B * T0, C, dtype=tdt, device=dev)
    stats = torch.zeros(B * C * 2, dtype=torch.float64, device=dev)
    ws = torch.empty(ops.conv0_workspace_floats(B, C, k), dtype=torch.float32, device=dev)
    p = ops.conv0_params(waved, wd, cbd, gd if group else None, bd if group else None, stats if group else None, yd, B, N, C, k, s, T0, group,
                         partials=ws)
    ops.conv0_fwd(p, dtype)
    nm = f"conv0 dt{dtype} group{int(group)} bias{int(bias)}"
    ok &= check(nm + " fwd", yd.view(B, T0, C), yr.detach(), tol)
    dy = torch.randn(B, T0, C).to(tdt)
    yr.backward(dy.float())
    dw = torch.zeros(C, k, device=dev); dcb = torch.zeros(C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    bst = torch.zeros(B * C * 2, dtype=torch.float64, device=dev)
    ops.conv0_bwd(p, dy.to(dev).contiguous(), bst if group else None, dw, dcb if (bias and not group) else None,
                  dg if group else None, db if group else None, dtype)
    ok &= check(nm + " dw", dw, wr.grad.view(C, k), tol * 20)
    if group:
        ok &= check(nm + " dgamma", dg, gr.grad, tol * 20)
        ok &= check(nm + " dbeta", db, br.grad, tol * 20)
    elif bias:
        ok &= check(nm + " dcbias", dcb, cbr.grad, tol * 20)
    return ok


def main():
    torch.manual_seed(0)
    allok = True
    for dtype, tdt, tol in ((ops.F32, torch.float32, 3e-5), (ops.BF16, torch.bfloat16, 2e-2)):
        allok &= conv0_case(dtype, tdt, tol, True)
        allok &= conv0_case(dtype, tdt, tol, False, bias=True)
        allok &= conv0_case(dtype, tdt, tol, True, B=3, N=8000, C=512)
    print("ALL OK" if allok else "SOME FAILED")


if __name__ == "__main__":
    main()


def conv_dgrad_case(dtype, tdt, tol, B, Tin, Cin, Co, k, s):
    """Replicates Engine.cnn_bwd's phase-decomposed data gradient for one conv layer."""
    from speechmix_amd.ops import view
    To = (Tin - k) // s + 1
    PAD = 2
    Tp = To + 2 * PAD
    w = (torch.randn(Co, Cin, k) * 0.2).to(tdt)
    dy = torch.randn(B, To, Co).to(tdt)
    xr = torch.zeros(B, Cin, Tin, requires_grad=True)
    y = F.conv1d(xr, w.float(), stride=s)
    y.backward(dy.float().transpose(1, 2))
    ref = xr.grad.transpose(1, 2)
    wp = w.float().permute(0, 2, 1).contiguous().view(Co, k * Cin).to(tdt).to(dev)
    dpre = torch.zeros(B, Tp, Co, dtype=tdt, device=dev)
    dpre[:, PAD:PAD + To] = dy.to(dev)
    dprev = torch.full((B * Tin, Cin), 7.0, dtype=tdt, device=dev)
    for r in range(s):
        taps = list(range(r, k, s)); nj = len(taps)
        U = (Tin - 1 - r) // s + 1
        av = view(Co, U, Tp * Co, (PAD - (nj - 1)) * Co)
        bv = view(k * Cin, Co, -s * Cin, (r + (nj - 1) * s) * Cin)
        cv = view(s * Cin, U, Tin * Cin, r * Cin)
        ops.gemm(dpre, wp, dprev, B * U, Cin, nj * Co, dtype, b_rc=True, av=av, bv=bv, cv=cv)
    return check(f"conv dgrad dt{dtype} B{B} Tin{Tin} Cin{Cin} Co{Co} k{k} s{s}", dprev.view(B, Tin, Cin), ref, tol)


def main2():
    torch.manual_seed(1)
    ok = True
    for dtype, tdt, tol in ((ops.F32, torch.float32, 3e-5), (ops.BF16, torch.bfloat16, 2e-2)):
        for (B, Tin, Cin, Co, k, s) in ((2, 399, 32, 32, 3, 2), (2, 1599, 32, 32, 3, 2), (2, 1598, 32, 32, 3, 2), (1, 49, 32, 32, 2, 2),
                                        (2, 1599, 32, 40, 3, 2), (3, 200, 64, 32, 3, 2)):
            ok &= conv_dgrad_case(dtype, tdt, tol, B, Tin, Cin, Co, k, s)
    print("DGRAD ALL OK" if ok else "DGRAD SOME FAILED")


if __name__ == "__main__":
    main2()
