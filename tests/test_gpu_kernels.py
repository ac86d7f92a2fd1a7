"""Kernel-level parity through the C ABI on the MI355X: each HIP kernel against a CPU fp32 reference of the same
op on the same seeded inputs (bf16 inputs are rounded first; tolerances are in the tools/ check scripts:
fp32 2e-5..3e-5 relative to the tensor max, bf16 1.5e-2..2e-2)."""
import pytest

pytestmark = pytest.mark.gpu


def test_gemm_variants_views_epilogues_splitk():
    from tools import gpu_check_gemm
    assert gpu_check_gemm.main() == 0


def test_pingpong_gemm_matches_128_kernel_all_layouts_and_epilogues(capsys):
    """256x256 persistent kernel (tr_mode 8) vs the 128x128 kernel on the model's shapes: fwd / dgrad / split-K wgrad,
    conv row views, batched launches, every fast epilogue class and the generic one, repeated (race screen)."""
    import sys
    from tools import gpu_check_pp
    argv, sys.argv = sys.argv, ["gpu_check_pp.py", "quick", "notime"]
    try:
        assert gpu_check_pp.main() == 0
    finally:
        sys.argv = argv
    assert "FAIL" not in capsys.readouterr().out


def test_colsum_two_stage_and_atomic():
    import torch
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (M, N, ld) in ((15968, 768, 768), (7968, 3072, 3072), (1024, 768, 768), (300, 50, 56), (4096, 512, 1536), (140000, 512, 512)):
        x = torch.randn(M, ld, device=dev)
        for dt, tdt, tol in ((ops.F32, torch.float32, 2e-5), (ops.BF16, torch.bfloat16, 2e-5)):
            xx = x.to(tdt)
            out = torch.full((N,), 0.5, dtype=torch.float32, device=dev)
            ops.colsum(xx, out, M, N, ld, dt, alpha=0.25)
            ref = 0.5 + 0.25 * xx[:, :N].double().sum(0)
            err = (out.double() - ref).abs().max().item()
            assert err <= tol * max(1.0, ref.abs().max().item()) * 4, (M, N, dt, err)


def test_dropout_colsum_equals_dropout_then_colsum():
    import torch
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    for (M, N) in ((15968, 768), (1024, 768), (300, 64), (7968, 3072)):
        for dt, tdt in ((ops.BF16, torch.bfloat16), (ops.F32, torch.float32)):
            x = torch.randn(M, N, device=dev).to(tdt)
            o1, o2 = torch.empty_like(x), torch.empty_like(x)
            ops.dropout(x, o1, M * N, 0.1, 77, dt)
            s2 = torch.full((N,), 0.25, dtype=torch.float32, device=dev)
            ops.dropout_colsum(x, o2, M, N, 0.1, 77, s2, dt)
            assert torch.equal(o1, o2)
            ref = 0.25 + o1.double().sum(0)
            assert (s2.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()) * 4


def test_adafactor_flat_step_matches_oracle():
    """smx_adafactor_step over a flat buffer of model-shaped tensors vs the oracle restatement of HF Adafactor (pinned to
    the HF class in tests/test_oracle_golden.py): three steps, one tensor without a gradient in step 2, global-norm
    clipping on, bf16 compute copy refreshed."""
    import torch
    from oracle import speechmix_oracle as O
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    shapes = [(768, 3072), (512, 512, 3), (3072,), (1, 5027), (768, 48, 128), (1, 1, 128), (512, 1, 10), (100, 4100), (768,)]
    offs, total = [], 0
    for s in shapes:
        offs.append(total)
        n = 1
        for d in s:
            n *= d
        total += (n + 63) // 64 * 64
    p = torch.randn(total) * 0.3
    ref = [p[o:o + torch.Size(s).numel()].view(s).clone() for o, s in zip(offs, shapes)]
    states = [dict() for _ in shapes]
    pd = p.to(dev)
    shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev)
    plan = ops.AdafactorPlan(list(zip(offs, shapes)), dev)
    gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
    for step in range(3):
        g = torch.zeros(total)
        for i, (o, s) in enumerate(zip(offs, shapes)):
            g[o:o + torch.Size(s).numel()] = torch.randn(torch.Size(s).numel()) * (5.0 if (i == 2 and step == 1) else 0.05)
        active = [not (step == 1 and i == 4) for i in range(len(shapes))]
        gd = g.to(dev)
        ops.sumsq(gd, total, gnorm)
        clip = min(1.0, 1.0 / (g.norm().item() * 0.5 + 1e-6))
        plan.step(pd, gd, shadow, 5e-4, active=active, grad_scale=0.5, max_grad_norm=1.0)
        for i, (o, s) in enumerate(zip(offs, shapes)):
            if active[i]:
                O.adafactor_step(ref[i], g[o:o + torch.Size(s).numel()].view(s) * (0.5 * clip), states[i], lr=5e-4)
        got = pd.cpu()
        for i, (o, s) in enumerate(zip(offs, shapes)):
            a, b = got[o:o + torch.Size(s).numel()].view(s), ref[i]
            err = (a - b).abs().max().item()
            assert err < 2e-6 + 2e-5 * 5e-4, (step, s, err)
    sh = shadow.float().cpu()
    for i, (o, s) in enumerate(zip(offs, shapes)):
        n = torch.Size(s).numel()
        assert torch.equal(sh[o:o + n], got[o:o + n].bfloat16().float())


def test_norm_and_attention_fwd_bwd():
    from tools import gpu_check_ops
    assert gpu_check_ops.main() == 0


def test_conv0_groupnorm_gelu_and_conv_dgrad_views(capsys):
    from tools import gpu_check_misc
    gpu_check_misc.main()
    gpu_check_misc.main2()
    out = capsys.readouterr().out
    assert "ALL OK" in out and "DGRAD ALL OK" in out and "FAIL" not in out


def test_native_step_runner_trains_and_matches_autograd_path():
    import torch
    from speechmix_amd.model import SpeechMixEED
    from speechmix_amd.trainer import StepRunner
    from tests.golden_util import load_case
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32").eval()
    model.load_state_dict(sd, strict=False)
    # (1) gradients through the autograd.Function == gradients left in the flat buffer by the native runner
    out = model(inp["input_values"], labels=inp["labels"])
    out["loss"].backward()
    g_auto = model.store.grad.clone()
    runner = StepRunner(model, lr=0.0, optimizer="sgd", max_grad_norm=0.0)
    l0 = runner.step(inp["input_values"], inp["labels"])
    assert abs(l0.item() - gold["loss"].item()) < 1e-4
    assert torch.allclose(model.store.grad, g_auto, atol=1e-6, rtol=1e-4)
    # (2) loss goes down with a real learning rate, parameters move, bf16 copies (if any) stay in sync
    runner2 = StepRunner(model, lr=1e-3, optimizer="adamw")
    first = runner2.step(inp["input_values"], inp["labels"]).item()
    for _ in range(8):
        last = runner2.step(inp["input_values"], inp["labels"]).item()
    assert last < first - 0.1


def test_deferred_folds_equal_the_immediate_second_stages():
    """smx_fold_many against torch, and the queued forms of colsum / dropout_colsum / norm_bwd against their immediate forms."""
    import torch
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    # (1) the fold kernel itself: short and long entries, a row offset inside the scratch, alpha, a shared destination
    q = ops.FoldQueue()
    ws1 = torch.randn(40, 776, device=dev); d1 = torch.randn(768, device=dev); r1 = d1 + 0.5 * ws1[:, :768].sum(0)
    ws2 = torch.randn(998, 2 * 512, device=dev); d2 = torch.zeros(512, device=dev); d3 = torch.zeros(512, device=dev)
    q.add(ws1, 0, d1, 40, 768, 776, 0.5)
    q.add(ws2, 0, d2, 998, 512, 1024)
    q.add(ws2, 512, d3, 998, 512, 1024)
    q.add(ws1, 0, d3, 40, 512, 776)                          # second entry on the same destination
    q.flush()
    assert torch.allclose(d1, r1, atol=1e-4, rtol=1e-5)
    assert torch.allclose(d2, ws2[:, :512].sum(0), atol=2e-3, rtol=1e-4)
    assert torch.allclose(d3, ws2[:, 512:].sum(0) + ws1[:, :512].sum(0), atol=2e-3, rtol=1e-4)
    assert not q.items
    # more entries than one table holds
    outs = [torch.zeros(64, device=dev) for _ in range(60)]
    srcs = [torch.randn(8, 64, device=dev) for _ in range(60)]
    for s, o in zip(srcs, outs):
        q.add(s, 0, o, 8, 64, 64)
    q.flush()
    assert all(torch.allclose(o, s.sum(0), atol=1e-5) for s, o in zip(srcs, outs))
    # (2) queued == immediate at the three sites
    M, N = 8192, 768
    x = torch.randn(M, N, device=dev).bfloat16()
    a, b = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    ops.colsum(x, a, M, N, N, ops.BF16)
    ops.colsum(x, b, M, N, N, ops.BF16, folds=q)
    assert len(q.items) == 1 and float(b.abs().max()) == 0.0
    q.flush()
    assert torch.allclose(a, b, atol=1e-3, rtol=1e-5)
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    a.zero_(); b.zero_()
    ops.dropout_colsum(x, o1, M, N, 0.1, 77, a, ops.BF16)
    ops.dropout_colsum(x, o2, M, N, 0.1, 77, b, ops.BF16, folds=q)
    q.flush()
    assert torch.equal(o1, o2) and torch.allclose(a, b, atol=1e-3, rtol=1e-5)
    D = 768
    xn = torch.randn(M, D, device=dev).bfloat16(); dy = torch.randn(M, D, device=dev).bfloat16()
    gamma, beta = torch.randn(D, device=dev), torch.randn(D, device=dev)
    mean, rstd = xn.float().mean(1), 1.0 / xn.float().std(1)
    res = []
    for folds in (None, q):
        dx = torch.empty_like(xn); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
        ops.norm_bwd(dy, xn, dx, gamma, beta, mean, rstd, dg, db, M, D, ops.BF16, folds=folds)
        if folds is not None:
            folds.flush()
        res.append((dx, dg, db))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.allclose(res[0][1], res[1][1], atol=5e-2, rtol=1e-4) and torch.allclose(res[0][2], res[1][2], atol=5e-2, rtol=1e-4)


def test_saved_activation_derivative_matches_the_recomputed_one():
    """ACT_SAVE_GRAD: the forward GEMM writes act'(pre) * dropout multiplier into the side tensor and the backward GEMM
    multiplies it in; same outputs as the pre-activation form up to the bf16 rounding of the derivative.  Both the direct
    kernels and the split-K pair (decoder-sized M) are covered."""
    import torch
    from speechmix_amd import ops
    from speechmix_amd.ops import ACT_GELU, ACT_RELU, view
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (M, N, K, act) in ((4000, 3072, 768, ACT_GELU), (1000, 512, 256, ACT_RELU)):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev) * 0.1
        dY = torch.randn(M, K, device=dev).bfloat16()          # gradient wrt the NEXT layer's output (fc2: N -> K)
        W2 = (torch.randn(K, N, device=dev) * 0.05).bfloat16()
        for drop in (None, (0.1, 99)):
            outs = []
            for flag in (0, ops.ACT_SAVE_GRAD):
                Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); S = torch.zeros_like(Y)
                ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, act=act | flag, aux_out=S, drop=drop, tr_mode=1)
                D = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                ops.gemm(dY, W2, D, M, N, K, ops.BF16, b_rc=True, bv=view(N), aux_in=S, act=act | flag, drop=drop, tr_mode=1)
                outs.append((Y, S, D))
            (Y0, S0, D0), (Y1, S1, D1) = outs
            assert torch.equal(Y0, Y1)                                   # the forward output does not change
            pre = S0.float()
            if act == ACT_GELU:
                ref = 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5
            else:
                ref = (pre > 0).float()
            if drop is not None:
                mask = torch.empty(M * N, device=dev)
                ops.dropout(torch.ones(M * N, device=dev), mask, M * N, drop[0], drop[1], ops.F32)
                ref = ref * mask.view(M, N)
            # derivative computed from the fp32 pre-activation vs from its bf16 copy: |gelu''| <= 0.8, |d pre| <= 2^-9 |pre|
            assert (S1.float() - ref).abs().max().item() <= 2e-2 * (1.0 / (1.0 - (drop[0] if drop else 0.0)))
            scale = D0.float().abs().max().item()
            assert (D1.float() - D0.float()).abs().max().item() <= 2e-2 * scale
    # split-K pair: epilogue kernel honours the flag, the 128x128 kernels refuse it outside their ACT / ACTGRAD classes
    M, N, K = 1024, 3072, 768
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    slabs = torch.empty(4 * M * N, dtype=torch.float32, device=dev)
    Y0 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); S0 = torch.zeros_like(Y0); Y1 = torch.zeros_like(Y0); S1 = torch.zeros_like(Y0)
    ops.gemm(A, W, Y0, M, N, K, ops.BF16, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=S0, drop=(0.1, 5), tr_mode=1)
    ops.gemm_splitk(A, W, Y1, M, N, K, ops.BF16, 4, slabs, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=S1, drop=(0.1, 5), tr_mode=1)
    assert (Y0.float() - Y1.float()).abs().max().item() <= 2e-2 * Y0.float().abs().max().item()
    assert (S0.float() - S1.float()).abs().max().item() <= 2e-2
    with pytest.raises(RuntimeError):
        ops.gemm(A, W, Y0, M, N, K, ops.BF16, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=S0, resid=Y1, tr_mode=1)


def test_grouped_weight_gradient_launch_equals_the_separate_launches():
    """smx_gemm_group: four weight-gradient problems of different output shapes (one with a ragged tile edge) and split
    counts in one launch; every slab must be bit-identical to the slab the single-problem launch writes (same kernel body,
    same K slices), and the plain (split 1) fp32 form must work inside a group too."""
    import torch
    from speechmix_amd import ops
    from speechmix_amd.ops import view
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    Kred = 4000
    shapes = [(768, 3072, 2), (3072, 768, 2), (2304, 768, 3), (200, 776, 1)]
    probs, refs, outs = [], [], []
    for (No, Ko, sp) in shapes:
        dy = torch.randn(Kred, No, device=dev).bfloat16()
        x = torch.randn(Kred, Ko, device=dev).bfloat16()
        n = No * Ko
        ref = torch.zeros(sp * n, dtype=torch.float32, device=dev)
        out = torch.zeros(sp * n, dtype=torch.float32, device=dev)
        kw = dict(a_rc=True, b_rc=True, av=view(No), bv=view(Ko), out_f32=True, atomic=0, alpha=0.5)
        if sp > 1:
            kw.update(split_k=sp, split_stride=n)
        ops.gemm(dy, x, ref, No, Ko, Kred, ops.BF16, tr_mode=8, **kw)
        probs.append((dy, x, out, No, Ko, Kred, kw))
        refs.append(ref); outs.append(out)
    ops.gemm_group(probs, ops.BF16)
    torch.cuda.synchronize()
    for (No, Ko, sp), ref, out in zip(shapes, refs, outs):
        assert ref.abs().max().item() > 0
        assert torch.equal(ref, out), (No, Ko, sp)
    fp = (probs[0][0].float().t() @ probs[0][1].float()) * 0.5
    got = outs[0].view(2, 768, 3072).sum(0)
    assert (got - fp).abs().max().item() <= 2e-3 * fp.abs().max().item()
    # second stage in one launch: every destination += the sum of its slabs
    dsts = [torch.ones(No * Ko, dtype=torch.float32, device=dev) for (No, Ko, sp) in shapes[:3]]
    ops.reduce_slabs_many([(o, sp, No * Ko, d) for (No, Ko, sp), o, d in zip(shapes[:3], outs[:3], dsts)], accumulate=True)
    for (No, Ko, sp), o, d in zip(shapes[:3], outs[:3], dsts):
        assert torch.equal(d, 1.0 + o.view(sp, -1).sum(0)) or (d - (1.0 + o.view(sp, -1).sum(0))).abs().max().item() <= 1e-5 * d.abs().max().item()
    with pytest.raises(RuntimeError):                                  # bf16 outputs are not a group class
        a, b, c, M, N, K, kw = probs[0]
        ops.gemm_group([(a, b, torch.zeros(M, N, dtype=torch.bfloat16, device=dev), M, N, K, dict(a_rc=True, b_rc=True, av=view(M), bv=view(N)))], ops.BF16)


@pytest.mark.parametrize("dtype_name", ["bf16", "fp32"])
def test_norm_backward_emits_the_masked_gradient_and_bias_gradient_of_the_dropped_linear(dtype_name):
    """Post-LN layers: the norm's input is x + dropout(Linear(...)), so its backward can also write dx * mask (the gradient
    entering that Linear) and that Linear's bias gradient.  Must equal the two-step form (smx_norm_bwd, then
    smx_dropout_colsum over the stored dx): the masked tensor bit for bit, the column sums up to fp32 summation order."""
    import torch
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    dt, tdt = (ops.BF16, torch.bfloat16) if dtype_name == "bf16" else (ops.F32, torch.float32)
    for (M, D) in ((1000, 768), (4099, 640), (130, 1024)):      # (D <= 512 takes another dx kernel: not bit-comparable)
        g = torch.Generator().manual_seed(M)
        x = torch.randn(M, D, generator=g).to(dev).to(tdt); dy = torch.randn(M, D, generator=g).to(dev).to(tdt)
        dres = torch.randn(M, D, generator=g).to(dev).to(tdt)
        gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev); beta = (0.1 * torch.randn(D, generator=g)).to(dev)
        y = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
        ops.norm_fwd(x, y, gamma, beta, mean, rstd, M, D, dt)
        drop2 = (0.1, 4242)
        # two-step reference
        f0 = ops.FoldQueue()
        dx0 = torch.empty_like(x); dg0 = torch.zeros(D, device=dev); db0 = torch.zeros(D, device=dev)
        ops.norm_bwd(dy, x, dx0, gamma, beta, mean, rstd, dg0, db0, M, D, dt, dres=dres, folds=f0)
        m0 = torch.empty_like(x); gb0 = torch.zeros(D, device=dev)
        ops.dropout_colsum(dx0, m0, M, D, drop2[0], drop2[1], gb0, dt, folds=f0)
        f0.flush()
        # fused
        f1 = ops.FoldQueue()
        dx1 = torch.empty_like(x); dg1 = torch.zeros(D, device=dev); db1 = torch.zeros(D, device=dev)
        m1 = torch.empty_like(x); gb1 = torch.zeros(D, device=dev)
        ops.norm_bwd(dy, x, dx1, gamma, beta, mean, rstd, dg1, db1, M, D, dt, dres=dres, folds=f1, drop2=drop2, dx_drop=m1, gb2=gb1)
        f1.flush()
        torch.cuda.synchronize()
        assert torch.equal(dx0, dx1) and torch.equal(m0, m1), (M, D)
        for r0, r1 in ((dg0, dg1), (db0, db1)):          # same partial rows, folded in a launch of another shape
            assert (r0 - r1).abs().max().item() <= 1e-5 * r0.abs().max().item() + 1e-6
        assert gb0.abs().max().item() > 0
        assert (gb0 - gb1).abs().max().item() <= 1e-5 * gb0.abs().max().item() + 1e-6, (M, D)
    with pytest.raises(RuntimeError):                       # the third partial row exists only in the deferred-fold form
        ops.norm_bwd(dy, x, dx1, gamma, beta, mean, rstd, dg1, db1, M, D, dt, drop2=drop2, dx_drop=m1, gb2=gb1, folds=None)


def test_eight_wave_256x128_kernel_is_bit_identical_to_the_128_kernel():
    """tr_mode 11 (256 x 128 tiles, eight waves, two workgroups per CU: 25 % fewer bytes into LDS per flop) runs the same K
    order and epilogue arithmetic as tr_mode 1: every instantiated class must agree bit for bit - forward LINEAR / ACT / ACT
    with the saved derivative, data gradient LINEAR / ACTGRAD / saved-derivative form - incl. ragged edges and a batched launch."""
    import torch
    from speechmix_amd import ops
    from speechmix_amd.ops import ACT_GELU, view
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    for (M, N, K) in ((15968, 768, 3072), (8000, 3072, 768), (1000, 200, 192), (300, 1536, 256)):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = W.t().contiguous()
        bias = torch.randn(N, device=dev) * 0.1
        R = torch.randn(M, N, device=dev).bfloat16()
        S = torch.randn(M, N, device=dev).bfloat16()
        cases = {
            "linear": dict(bias=bias, resid=R, drop=(0.1, 7)),
            "act": dict(bias=bias, act=ACT_GELU, aux_out="aux", drop=(0.1, 8)),
            "act_saved": dict(bias=bias, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out="aux", drop=(0.1, 9)),
            "dgrad": dict(b_rc=True, bv=view(N), resid=R),
            "dgrad_actgrad": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU, drop=(0.1, 10)),
            "dgrad_saved": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU | ops.ACT_SAVE_GRAD),
        }
        for name, kw in cases.items():
            outs = []
            for mode in (1, 11):
                Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                aux = torch.zeros_like(Y)
                k2 = {k: (aux if isinstance(v, str) else v) for k, v in kw.items()}
                ops.gemm(A, Wt if kw.get("b_rc") else W, Y, M, N, K, ops.BF16, tr_mode=mode, **k2)
                outs.append((Y, aux))
            assert torch.equal(outs[0][0], outs[1][0]), (name, M, N, K)
            assert torch.equal(outs[0][1], outs[1][1]), (name, M, N, K)
            assert outs[0][0].float().abs().max().item() > 0
    # batched (grid.z) launch
    G, M, N, K = 3, 700, 256, 320
    A = torch.randn(G, M, K, device=dev).bfloat16(); W = (torch.randn(G, N, K, device=dev) * 0.05).bfloat16()
    outs = []
    for mode in (1, 11):
        Y = torch.zeros(G, M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(A, W, Y, M, N, K, ops.BF16, nbatch=G, batch_a=M * K, batch_b=N * K, batch_c=M * N, tr_mode=mode)
        outs.append(Y)
    assert torch.equal(outs[0], outs[1]) and outs[0].float().abs().max().item() > 0
    with pytest.raises(RuntimeError):
        ops.gemm(A[0], W[0], torch.zeros(M, N, dtype=torch.float32, device=dev), M, N, K, ops.BF16, out_f32=True, tr_mode=11)


def test_half_height_tiles_of_the_128_kernel_are_bit_identical_to_it():
    """tr_mode 9 (64 x 128 tiles: twice the workgroups for launches that leave most resident slots empty) runs the same
    K order and the same epilogue arithmetic as tr_mode 1, so every instantiated class must agree bit for bit: forward
    LINEAR (+bias, dropout, residual), ACT (+pre-activation copy), ACT with the saved derivative, data gradient LINEAR,
    ACTGRAD and its saved-derivative form; ragged M; classes it is not built for are refused."""
    import torch
    from speechmix_amd import ops
    from speechmix_amd.ops import ACT_GELU, view
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    for (M, N, K) in ((7968, 768, 3072), (1024, 768, 768), (1000, 200, 192), (130, 3072, 768)):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = W.t().contiguous()                                  # [K, N]: rows-contiguous B operand of a data gradient
        bias = torch.randn(N, device=dev) * 0.1
        R = torch.randn(M, N, device=dev).bfloat16()
        S = torch.randn(M, N, device=dev).bfloat16()
        cases = {
            "linear": dict(bias=bias, resid=R, drop=(0.1, 7)),
            "act": dict(bias=bias, act=ACT_GELU, aux_out="aux", drop=(0.1, 8)),
            "act_saved": dict(bias=bias, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out="aux", drop=(0.1, 9)),
            "dgrad": dict(b_rc=True, bv=view(N)),
            "dgrad_actgrad": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU, drop=(0.1, 10)),
            "dgrad_saved": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU | ops.ACT_SAVE_GRAD),
        }
        for name, kw in cases.items():
            outs = []
            for mode in (1, 9):
                Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                aux = torch.zeros_like(Y)
                k2 = {k: (aux if isinstance(v, str) else v) for k, v in kw.items()}
                ops.gemm(A, Wt if kw.get("b_rc") else W, Y, M, N, K, ops.BF16, tr_mode=mode, **k2)
                outs.append((Y, aux))
            assert torch.equal(outs[0][0], outs[1][0]), (name, M, N, K)
            assert torch.equal(outs[0][1], outs[1][1]), (name, M, N, K)
            assert outs[0][0].float().abs().max().item() > 0
    Yf = torch.zeros(1024, 768, dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError):                            # fp32 outputs (split-K slabs) stay on the 128-row kernel
        ops.gemm(torch.zeros(1024, 768, device=dev).bfloat16(), torch.zeros(768, 768, device=dev).bfloat16(), Yf, 1024, 768, 768,
                 ops.BF16, out_f32=True, tr_mode=9)


def test_saved_derivative_epilogues_of_the_256_wide_kernels_match_the_128_kernel():
    """SMX_ACT_SAVE_GRAD on the ping-pong / free-running kernels (tr_mode 8, 12, 13; round 3): forward ACT writes
    act(pre) x mask and the local derivative act'(pre) x mask, the data gradient multiplies by the saved derivative.  Same
    arithmetic as the 128-family epilogues (epilogue_staged_fast<4 / 5>) on the same fp32 accumulators (same K order): bit for
    bit, with and without dropout, GELU and ReLU, ragged M; unsupported combinations are refused."""
    import torch
    from speechmix_amd import ops
    from speechmix_amd.ops import ACT_GELU, ACT_RELU, view
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    for (M, N, K) in ((15968, 3072, 768), (4000, 768, 256), (1000, 1024, 192)):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = W.t().contiguous()
        bias = torch.randn(N, device=dev) * 0.1
        S = torch.randn(M, N, device=dev).bfloat16()
        for act in (ACT_GELU, ACT_RELU):
            for drop in (None, (0.1, 11)):
                ref = None
                for mode in (1, 8, 12, 13):
                    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                    aux = torch.zeros_like(Y)
                    ops.gemm(A, W, Y, M, N, K, ops.BF16, bias=bias, act=act | ops.ACT_SAVE_GRAD, aux_out=aux, drop=drop, tr_mode=mode)
                    D = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                    ops.gemm(A, Wt, D, M, N, K, ops.BF16, b_rc=True, bv=view(N), aux_in=S, act=act | ops.ACT_SAVE_GRAD, tr_mode=mode)
                    if ref is None:
                        ref = (Y, aux, D)
                        assert Y.float().abs().max().item() > 0 and aux.float().abs().max().item() > 0
                        continue
                    assert torch.equal(ref[0], Y), ("out", M, N, K, act, drop, mode)
                    assert torch.equal(ref[1], aux), ("derivative", M, N, K, act, drop, mode)
                    assert torch.equal(ref[2], D), ("dgrad", M, N, K, act, drop, mode)
    M, N, K = 1024, 512, 256
    A = torch.zeros(M, K, device=dev).bfloat16(); W = torch.zeros(N, K, device=dev).bfloat16()
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    for mode in (8, 12):
        with pytest.raises(RuntimeError):          # a data gradient with a bias has no saved-derivative class
            ops.gemm(A, W.t().contiguous(), Y, M, N, K, ops.BF16, b_rc=True, bv=view(N), aux_in=Y, bias=torch.zeros(N, device=dev),
                     act=ACT_GELU | ops.ACT_SAVE_GRAD, tr_mode=mode)
        with pytest.raises(RuntimeError):          # residual + saved derivative: generic epilogue, not available
            ops.gemm(A, W, Y, M, N, K, ops.BF16, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=torch.zeros_like(Y), resid=torch.zeros_like(Y), tr_mode=mode)


@pytest.mark.parametrize("V", [50265, 130, 64])
def test_cross_entropy_two_pass_kernel_matches_torch(V):
    """loss (mean over valid tokens, ignore_index -100), first arg max and d loss / d logits of the vectorised two-pass kernel."""
    import torch
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(V)
    M, Vp = 96, (V + 7) // 8 * 8
    logits = torch.zeros(M, Vp, device=dev)
    logits[:, :V] = torch.randn(M, V, device=dev) * 3
    logits[5, 7] = logits[5, 3] = 50.0                      # a tie: the FIRST index wins (torch.argmax semantics on CPU)
    labels = torch.randint(0, V, (M,), device=dev)
    labels[::7] = -100
    for dt, tdt, tol in ((ops.F32, torch.float32, 1e-6), (ops.BF16, torch.bfloat16, 1e-2)):
        loss = torch.zeros(1, device=dev); am = torch.empty(M, dtype=torch.int64, device=dev)
        dl = torch.full((M, Vp), 7.0, dtype=tdt, device=dev)
        ops.cross_entropy(logits, labels, loss, am, dl, M, V, Vp, Vp, dt, gscale=2.0)
        ref_in = logits[:, :V].clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(ref_in, labels, ignore_index=-100)
        (2.0 * ref).backward()
        assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
        assert am[5].item() == 3 and torch.equal(am, logits[:, :V].cpu().argmax(-1).to(dev))
        assert (dl[:, :V].float() - ref_in.grad).abs().max().item() <= tol * ref_in.grad.abs().max().item() + 1e-9
        assert float(dl[:, V:].float().abs().max()) == 0.0 if Vp > V else True


def test_gradient_accumulation_equals_one_step_on_the_concatenated_batch():
    """StepRunner(grad_accum=2) over two micro-batches == one step over their concatenation (fp32, eval mode, SGD): the
    loss of every micro-batch is divided by the count, gradients add up, the update runs on the last one."""
    import torch
    from speechmix_amd.model import SpeechMixEED
    from speechmix_amd.trainer import StepRunner
    from tests.golden_util import load_case
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    wave, labels = inp["input_values"], inp["labels"].clone()
    labels[labels == -100] = 5                      # same number of valid tokens in both halves: mean of means == global mean
    def fresh():
        model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32").eval()
        model.load_state_dict(sd, strict=False)
        return model
    a = fresh()
    ra = StepRunner(a, lr=1e-2, optimizer="sgd", max_grad_norm=0.0)
    ra.step(wave, labels)
    b = fresh()
    rb = StepRunner(b, lr=1e-2, optimizer="sgd", max_grad_norm=0.0, grad_accum=2)
    before = b.store.master.clone()
    rb.step(wave[:1], labels[:1])
    assert torch.equal(b.store.master, before)      # nothing is updated on the first micro-batch
    rb.step(wave[1:], labels[1:])
    assert not torch.equal(b.store.master, before)
    assert torch.allclose(a.store.master, b.store.master, atol=2e-6, rtol=1e-4)


@pytest.mark.parametrize("case,dtype", [("eed_w2v2_bart", "fp32"), ("eed_w2v2_bart", "bf16")])
def test_lm_weight_gradients_on_the_second_stream_equal_the_single_stream_result(case, dtype, monkeypatch):
    """The LM stage runs its weight-gradient GEMMs on a second stream (engine.Engine.wgrad); same gradients as with
    everything on one stream, repeated to give a race a chance to show."""
    import torch
    from speechmix_amd.model import SpeechMixEED
    from speechmix_amd.trainer import StepRunner
    from tests.golden_util import load_case
    sd, inp, gold, m = load_case(case)
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype=dtype).eval()
    model.load_state_dict(sd, strict=False)
    runner = StepRunner(model, lr=0.0, optimizer="sgd", max_grad_norm=0.0)
    monkeypatch.setenv("SMX_LM_WGRAD_STREAM", "0")
    runner.step(inp["input_values"], inp["labels"])
    ref = model.store.grad.clone()
    monkeypatch.setenv("SMX_LM_WGRAD_STREAM", "1")
    for _ in range(5):
        runner.step(inp["input_values"], inp["labels"])
        assert model.engine._side is not None
        assert torch.allclose(model.store.grad, ref, atol=1e-6 if dtype == "fp32" else 1e-3, rtol=1e-5 if dtype == "fp32" else 1e-2)
