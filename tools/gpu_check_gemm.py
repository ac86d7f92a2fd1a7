"""GPU bring-up check for smx_gemm (run on the MI355X box).  Prints max errors and timings."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import _lib as L

dev = torch.device("cuda:0")
lib = L.lib()
stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def view(ld, rpb=0, bs=0, off=0):
    return L.RowView(bs, ld, off, rpb, 0)


def run(A, B, Cout, M, N, K, a_rc, b_rc, dtype, av=None, bv=None, cv=None, bias=None, resid=None, aux_out=None,
        aux_in=None, act=0, out_f32=0, atomic=0, split_k=1, tr_mode=1, alpha=1.0, nbatch=1, ba=0, bb=0, bc=0, split_stride=0):
    p = L.GemmParams()
    p.A, p.B, p.C = A.data_ptr(), B.data_ptr(), Cout.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.resid = resid.data_ptr() if resid is not None else None
    p.aux_out = aux_out.data_ptr() if aux_out is not None else None
    p.aux_in = aux_in.data_ptr() if aux_in is not None else None
    p.a = av or view(A.stride(0)); p.b = bv or view(B.stride(0)); p.c = cv or view(Cout.stride(0))
    p.e = p.c
    p.batch_a, p.batch_b, p.batch_c, p.batch_bias, p.batch_e = ba, bb, bc, 0, bc
    p.M, p.N, p.K, p.a_rc, p.b_rc = M, N, K, a_rc, b_rc
    p.act, p.out_f32, p.atomic, p.nbatch, p.split_k, p.tr_mode, p.alpha = act, out_f32, atomic, nbatch, split_k, tr_mode, alpha
    p.split_stride = split_stride
    rc = lib.smx_gemm(C.byref(p), dtype, stream())
    assert rc == 0, rc


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def check(name, got, ref, tol):
    err = (got.float().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    ok = err <= tol * max(scale, 1.0)
    print(f"{'OK  ' if ok else 'FAIL'} {name}: max_err={err:.3e} ref_max={scale:.3e}")
    return ok


def main():
    torch.manual_seed(0)
    allok = True
    for dtype, tdt, tol in ((L.F32, torch.float32, 2e-5), (L.BF16, torch.bfloat16, 1.5e-2)):
        for (M, N, K) in ((200, 136, 72), (128, 128, 64), (257, 384, 200), (96, 50, 328)):
            if dtype == L.BF16:
                K = (K + 7) // 8 * 8
            for tr in ((1, 0) if dtype == L.BF16 else (1,)):
                Ah = torch.randn(M, K); Bh = torch.randn(N, K)
                A = Ah.to(tdt).to(dev); B = Bh.to(tdt).to(dev)
                Af, Bf = A.float().cpu(), B.float().cpu()
                bias = torch.randn(N, device=dev)
                # NT + bias + gelu + aux_out + resid
                Np = (N + 7) // 8 * 8
                Cc = torch.zeros(M, Np, dtype=tdt, device=dev); aux = torch.zeros_like(Cc)
                resid = torch.randn(M, Np).to(tdt).to(dev)
                run(A, B, Cc, M, N, K, 0, 0, dtype, bias=bias, resid=resid, aux_out=aux, act=1, tr_mode=tr)
                pre = Af @ Bf.t() + bias.cpu()
                allok &= check(f"dt{dtype} NT {M}x{N}x{K} tr{tr} aux", aux[:, :N], pre, tol)
                allok &= check(f"dt{dtype} NT {M}x{N}x{K} tr{tr} out", Cc[:, :N], gelu(pre) + resid[:, :N].float().cpu(), tol)
                # NN (dgrad): C[M,K2] = A[M,N2] * W[N2,K2]  -> here: A [M,K] , W [K, N] rows-contiguous operand
                Wt = B.t().contiguous()  # [K, N]
                Cc2 = torch.zeros(M, Np, dtype=tdt, device=dev)
                auxin = torch.randn(M, Np).to(tdt).to(dev)
                run(A, Wt, Cc2, M, N, K, 0, 1, dtype, bv=view(Wt.stride(0)), aux_in=auxin, act=1, tr_mode=tr)
                x = auxin[:, :N].float().cpu()
                gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * 3.141592653589793) ** 0.5
                allok &= check(f"dt{dtype} NN {M}x{N}x{K} tr{tr}", Cc2[:, :N], (Af @ Bf.t()) * gp, tol)
                # TN (wgrad): C[M,N] = At[K,M]^T * Bt[K,N], fp32 atomic accumulate with split_k
                At = A.t().contiguous(); Bt = B.t().contiguous()
                Cc3 = torch.ones(M, N, dtype=torch.float32, device=dev)
                run(At, Bt, Cc3, M, N, K, 1, 1, dtype, av=view(At.stride(0)), bv=view(Bt.stride(0)), out_f32=1,
                    atomic=1, split_k=3, tr_mode=tr, alpha=0.5)
                allok &= check(f"dt{dtype} TN {M}x{N}x{K} tr{tr} splitk3", Cc3, 1.0 + 0.5 * (Af @ Bf.t()), tol)
    # conv-as-GEMM with an overlapping row view: x [Bz, T, Cin] channels-last, k=3, s=2
    for dtype, tdt, tol in ((L.F32, torch.float32, 2e-5), (L.BF16, torch.bfloat16, 1.5e-2)):
        Bz, T, Cin, Cout, k, s = 3, 41, 32, 40, 3, 2
        To = (T - k) // s + 1
        x = torch.randn(Bz, T, Cin).to(tdt).to(dev)
        w = torch.randn(Cout, Cin, k).to(tdt)
        wp = w.permute(0, 2, 1).contiguous().view(Cout, k * Cin).to(dev)   # [co, tap*Cin + ci]
        y = torch.zeros(Bz, To, Cout, dtype=tdt, device=dev)
        run(x, wp, y, Bz * To, Cout, k * Cin, 0, 0, dtype, av=view(s * Cin, To, T * Cin), cv=view(Cout))
        ref = torch.nn.functional.conv1d(x.float().cpu().transpose(1, 2), w.float(), stride=s).transpose(1, 2)
        allok &= check(f"dt{dtype} conv-view k3 s2", y, ref, tol)
        # wgrad through the same overlapping view: dW[co, tap*Cin+ci] = sum_rows dy[row,co] * xcol[row, :]
        dy = torch.randn(Bz, To, Cout).to(tdt).to(dev)
        dW = torch.zeros(Cout, k * Cin, dtype=torch.float32, device=dev)
        run(dy, x, dW, Cout, k * Cin, Bz * To, 1, 1, dtype, av=view(Cout), bv=view(s * Cin, To, T * Cin), out_f32=1,
            atomic=1, split_k=2)
        xc = x.float().cpu()
        cols = torch.stack([xc[:, j * s:j * s + k, :].reshape(Bz, -1) for j in range(To)], 1)  # [B,To,k*Cin]
        refW = dy.float().cpu().reshape(-1, Cout).t() @ cols.reshape(-1, k * Cin)
        allok &= check(f"dt{dtype} conv-view wgrad", dW, refW, tol)
    # batched (grid.z) GEMM
    G, M, N, K = 4, 70, 48, 96
    A = torch.randn(G, M, K).bfloat16().to(dev); B = torch.randn(G, N, K).bfloat16().to(dev)
    Cc = torch.zeros(M, G * N, dtype=torch.bfloat16, device=dev)
    run(A, B, Cc, M, N, K, 0, 0, L.BF16, av=view(K), bv=view(K), cv=view(G * N), nbatch=G, ba=M * K, bb=N * K, bc=N)
    ref = torch.cat([A[g].float().cpu() @ B[g].float().cpu().t() for g in range(G)], 1)
    allok &= check("bf16 batched grid.z", Cc, ref, 1.5e-2)

    # ---- timings (bf16) ----
    def bench(name, fn, flops, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print(f"TIME {name}: {dt*1e6:.1f} us  {flops/dt/1e12:.1f} TFLOP/s")
    M, N, K = 15968, 3072, 768
    A = torch.randn(M, K).bfloat16().to(dev); W = torch.randn(N, K).bfloat16().to(dev)
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev)
    bench("NT ffn1 15968x3072x768 +bias+gelu", lambda: run(A, W, Y, M, N, K, 0, 0, L.BF16, bias=bias, act=1), 2 * M * N * K)
    dX = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
    bench("NN dgrad 15968x768x3072 tr1", lambda: run(Y, W, dX, M, K, N, 0, 1, L.BF16), 2 * M * N * K)
    bench("NN dgrad 15968x768x3072 tr0", lambda: run(Y, W, dX, M, K, N, 0, 1, L.BF16, tr_mode=0), 2 * M * N * K)
    dW = torch.zeros(N, K, dtype=torch.float32, device=dev)
    for sk in (1, 4, 8):
        bench(f"TN wgrad 3072x768x15968 splitk{sk} tr1", lambda: run(Y, A, dW, N, K, M, 1, 1, L.BF16, out_f32=1, atomic=1, split_k=sk), 2 * M * N * K)
    bench("TN wgrad 3072x768x15968 splitk4 tr0", lambda: run(Y, A, dW, N, K, M, 1, 1, L.BF16, out_f32=1, atomic=1, split_k=4, tr_mode=0), 2 * M * N * K)
    M2, N2, K2 = 15968, 768, 768
    A2 = torch.randn(M2, K2).bfloat16().to(dev); W2 = torch.randn(N2, K2).bfloat16().to(dev)
    Y2 = torch.zeros(M2, N2, dtype=torch.bfloat16, device=dev)
    bench("NT proj 15968x768x768", lambda: run(A2, W2, Y2, M2, N2, K2, 0, 0, L.BF16), 2 * M2 * N2 * K2)
    Af = torch.randn(2048, 768, device=dev); Wf = torch.randn(3072, 768, device=dev); Yf = torch.zeros(2048, 3072, device=dev)
    bench("f32 simple NT 2048x3072x768", lambda: run(Af, Wf, Yf, 2048, 3072, 768, 0, 0, L.F32), 2 * 2048 * 3072 * 768, iters=5)
    print("ALL OK" if allok else "SOME FAILED")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
