"""Few-workgroup GEMMs (decoder side, text encoder) on the 128x128 kernel: the engine's split-K policy against other K
splits, and a K sweep that separates a launch's fixed cost from its per-K-tile cost.  `trace` mode runs under
rocprofv3 --kernel-trace (tools/trace_cfgs.py turns the trace into per-configuration kernel times): event timing of
launches this short measures the host's launch rate.  MODES: kernel variants to compare (tr_mode values)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.ops import ACT_GELU, view
from tools.gpu_check_pp import bench, cmp
MODES = tuple(int(x) for x in os.environ.get("SMX_SMALL_MODES", "1").split(","))
dev = torch.device("cuda:0")
TRACE = "trace" in sys.argv      # under rocprofv3 --kernel-trace: a marker kernel, then 10 launches per configuration
_mark = None
_cfg = [0]


def timed(name, fn, n=20):
    """-> us per call from events (bounded below by the host's launch rate), or in trace mode 0 after emitting the marker."""
    global _mark
    if not TRACE:
        return bench(fn, n=n)
    if _mark is None:
        _mark = torch.zeros(64, device=dev)
    torch.cuda.synchronize()
    _mark.fill_(float(_cfg[0]))
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"CFG {_cfg[0]} {name}", flush=True)
    _cfg[0] += 1
    return 0.0


def fsplit(Mo, No, Kred):            # Engine._fsplit
    tiles = ((Mo + 127) // 128) * ((No + 127) // 128)
    ksteps = (Kred + 63) // 64
    if tiles > 256 or ksteps < 6:
        return 1
    want = int(min(1024 // tiles, ksteps // 3 if ksteps <= 16 else ksteps // 4, 24))
    if want < 2:
        return 1
    per = (ksteps + want - 1) // want
    return (ksteps + per - 1) // per


def valid_split(K, want):
    kst = (K + 63) // 64
    want = max(1, min(want, kst))
    per = (kst + want - 1) // want
    return (kst + per - 1) // per


def run(A, W, Y, M, N, K, mode, split, slabs, **kw):
    if split > 1:
        ops.gemm_splitk(A, W, Y, M, N, K, ops.BF16, split, slabs, tr_mode=mode, **kw)
    else:
        ops.gemm(A, W, Y, M, N, K, ops.BF16, tr_mode=mode, **kw)


def ksweep():
    M, N = 1024, 768
    for K in (64, 128, 256, 512, 768, 1536):
        A = torch.randn(M, K, device=dev).bfloat16()
        Wm = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        F = torch.zeros(M, N, dtype=torch.float32, device=dev)
        if "lab" in sys.argv:
            timed(f"ksweep_lab_return K{K} m1", lambda: ops.gemm(A, Wm, Y, M, N, K, ops.BF16, tr_mode=1, drop=(0.0, 0xdead0001)))
            timed(f"ksweep_lab_noepi K{K} m1", lambda: ops.gemm(A, Wm, Y, M, N, K, ops.BF16, tr_mode=1, drop=(0.0, 0xdead0002)))
        for mode in MODES:
            timed(f"ksweep_bf16out K{K} m{mode}", lambda: ops.gemm(A, Wm, Y, M, N, K, ops.BF16, tr_mode=mode))
            timed(f"ksweep_f32out K{K} m{mode}", lambda: ops.gemm(A, Wm, F, M, N, K, ops.BF16, out_f32=True, tr_mode=mode))


def main():
    torch.manual_seed(0)
    ok = True
    if "ksweep" in sys.argv:
        ksweep()
        return 0
    shapes = [] if "wg16k" in sys.argv else [(1024, 768, 768), (1024, 3072, 768), (1024, 768, 3072), (1024, 2304, 768), (7968, 768, 768), (7968, 768, 3072),
              (7968, 3072, 768), (7968, 2304, 768), (7968, 1536, 768), (7968, 768, 1536), (15968, 768, 768)]
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device=dev).bfloat16()
        Wm = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = Wm.t().contiguous()
        bias = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev).bfloat16()
        slabs = torch.empty(24 * M * N, dtype=torch.float32, device=dev)
        Y0 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        Y1 = torch.zeros_like(Y0)
        s0 = fsplit(M, N, K)
        for name, Bop, kw in (("fwd", Wm, dict(bias=bias, resid=R, drop=(0.1, 5))),
                              ("dgrad", Wt, dict(b_rc=True, bv=view(N)))):
            run(A, Bop, Y0, M, N, K, 1, s0, slabs, **kw)
            t0 = timed(f"{name} {M}x{N}x{K} 128/s{s0}", lambda: run(A, Bop, Y0, M, N, K, 1, s0, slabs, **kw))
            res = [f"128/s{s0}: {t0:.1f}"]
            for mode in MODES:
                for want in (1, 2, 3, 4, 6):
                    sp = valid_split(K, want)
                    if sp != want:
                        continue
                    Y1.zero_()
                    run(A, Bop, Y1, M, N, K, mode, sp, slabs, **kw)
                    ok &= cmp(f"{name} {M}x{N}x{K} mode{mode} split{sp}", Y1, Y0)
                    t = timed(f"{name} {M}x{N}x{K} {mode}/s{sp}", lambda: run(A, Bop, Y1, M, N, K, mode, sp, slabs, **kw))
                    res.append(f"{mode}/s{sp}: {t:.1f}")
            print(f"TIME {name} M={M} N={N} K={K}: " + "  ".join(res), flush=True)
    # weight gradients of the decoder (reduction over 1024 rows) and the encoder's square projections
    wg_shapes = [(768, 768, 1024), (768, 3072, 1024), (3072, 768, 1024), (2304, 768, 1024), (768, 768, 7968)]
    if "wg16k" in sys.argv:
        wg_shapes = [(768, 768, 15968), (768, 768, 7968), (1536, 768, 7968)]
    for (No, Ko, Mred) in wg_shapes:
        Yb = torch.randn(Mred, No, device=dev).bfloat16(); Xb = torch.randn(Mred, Ko, device=dev).bfloat16()
        n = No * Ko
        out0 = torch.zeros(No, Ko, device=dev); out1 = torch.zeros_like(out0)
        slabs = torch.empty(32 * n, dtype=torch.float32, device=dev)

        def wg(mode, sp, out):
            if sp <= 1:
                ops.gemm(Yb, Xb, out, No, Ko, Mred, ops.BF16, a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True, tr_mode=mode)
            else:
                ops.gemm(Yb, Xb, slabs, No, Ko, Mred, ops.BF16, a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True,
                         split_k=sp, split_stride=n, tr_mode=mode)
                ops.reduce_slabs(slabs, sp, n, n, out, accumulate=False)
        tiles = ((No + 127) // 128) * ((Ko + 127) // 128)
        ksteps = (Mred + 63) // 64
        s0 = valid_split(Mred, int(max(1, min(1024 // max(tiles, 1), ksteps // 8, 32))))
        wg(1, s0, out0)
        res = [f"128/s{s0}: {timed(f'wgrad {No}x{Ko}x{Mred} 128/s{s0}', lambda: wg(1, s0, out0)):.1f}"]
        for mode in MODES:
            for want in ((7, 14, 21, 28, 32) if "wg16k" in sys.argv else (1, 2, 4, 8)):
                sp = valid_split(Mred, want)
                if sp != want:
                    continue
                out1.zero_()
                wg(mode, sp, out1)
                ok &= cmp(f"wgrad {No}x{Ko}x{Mred} mode{mode} split{sp}", out1, out0, 1e-3)
                res.append(f"{mode}/s{sp}: {timed(f'wgrad {No}x{Ko}x{Mred} {mode}/s{sp}', lambda: wg(mode, sp, out1)):.1f}")
        print(f"TIME wgrad {No}x{Ko}x{Mred}: " + "  ".join(res), flush=True)
    print("ALL OK" if ok else "SOME FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
