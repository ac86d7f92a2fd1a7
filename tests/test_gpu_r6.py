"""Round 6: the wave-specialised GEMM kernel (tr_mode 14, csrc/gemm_ws.hip), the batched weight transposes and the data gradients that
read the K-contiguous weight copies (Engine._wt).  Linear layers of ref:speechmix/model.py:148 -> TF:models/wav2vec2/modeling_wav2vec2.py:466-572."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_wave_specialised_gemm_matches_the_128_kernel_bit_for_bit():
    """Twelve compute + four loader waves on 192 x 256 tiles against the 128 x 128 kernel: same K order, same epilogue arithmetic ->
    bit-identical outputs for every epilogue class the step uses, on ragged M / N / K (K from 32: one-tile items; K % 64 != 0: the tail
    tile; split-K slabs for the weight gradient), each launch twice (warm LDS stages / item records / bias slots)."""
    from speechmix_amd import ops
    from speechmix_amd.ops import ACT_GELU, view
    dev = torch.device("cuda:0")
    rng = random.Random(23)
    torch.manual_seed(23)
    kinds = ["fwd", "fwd_act", "fwd_saved", "dgrad", "dgrad_actgrad", "dgrad_saved", "wgrad", "fwd_plain"]
    compared = 0
    for case in range(40):
        M = rng.choice([264, 1000, 4000, 7968, 15968]) + 8 * rng.randrange(0, 4)
        N = 8 * rng.randrange(8, 400)
        K = 8 * rng.randrange(4, 200)
        kind = kinds[case % len(kinds)]
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = W.t().contiguous()
        bias = torch.randn(N, device=dev) * 0.1
        S = torch.randn(M, N, device=dev).bfloat16()
        ref, split = None, rng.choice([1, 3])
        for mode in (1, 14, 14):
            try:
                if kind == "wgrad":
                    kst = (M + 63) // 64
                    per = (kst + split - 1) // split
                    sp = (kst + per - 1) // per
                    G = torch.zeros(sp, N, K, dtype=torch.float32, device=dev)
                    ops.gemm(S, A, G, N, K, M, ops.BF16, a_rc=True, b_rc=True, av=view(N), bv=view(K), out_f32=True, split_k=sp,
                             split_stride=N * K if sp > 1 else 0, tr_mode=mode)
                    res = (G,)
                else:
                    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                    aux = torch.zeros_like(Y)
                    kw = {"fwd": dict(bias=bias, resid=S, drop=(0.1, 4)), "fwd_act": dict(bias=bias, act=ACT_GELU, aux_out=aux, drop=(0.1, 5)),
                          "fwd_saved": dict(bias=bias, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=aux, drop=(0.1, 6)),
                          "dgrad": dict(b_rc=True, bv=view(N), resid=S), "dgrad_actgrad": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU),
                          "dgrad_saved": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU | ops.ACT_SAVE_GRAD), "fwd_plain": dict()}[kind]
                    ops.gemm(A, Wt if kw.get("b_rc") else W, Y, M, N, K, ops.BF16, tr_mode=mode, **kw)
                    res = (Y, aux)
            except RuntimeError:          # (a shape the kernel family refuses: the tuner never offers it)
                continue
            if mode == 1:
                ref = res
            else:
                compared += 1
                for a_, b_ in zip(ref, res):
                    assert torch.equal(a_, b_), (case, kind, M, N, K, mode, (a_.float() - b_.float()).abs().max().item())
    assert compared >= 70, compared


def test_transpose_many_is_exact():
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    shapes = [(768, 768), (2304, 768), (768, 3072), (3072, 768), (8, 8), (72, 200), (1000, 24), (50264, 768)]
    shapes = shapes * 9          # 72 matrices: two launches (64 per table)
    jobs = [(torch.randn(r, c, device=dev).bfloat16(), torch.zeros(c, r, dtype=torch.bfloat16, device=dev)) for r, c in shapes]
    ops.transpose_many(jobs)
    for src, dst in jobs:
        assert torch.equal(dst, src.t().contiguous()), src.shape


def test_data_gradient_through_the_transposed_weight_copy_is_bit_identical():
    """The same kernel (tr_mode forced) reading W rows-contiguous and W^T K-contiguous: identical operand values in the same K order."""
    from speechmix_amd import ops
    from speechmix_amd.ops import view
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    for (M, N, K) in [(15968, 2304, 768), (7968, 768, 768), (1024, 3072, 768), (4000, 768, 3072)]:
        dy = torch.randn(M, N, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        r = torch.randn(M, K, device=dev).bfloat16()
        wt = torch.empty(K, N, dtype=torch.bfloat16, device=dev)
        ops.transpose_many([(w, wt)])
        for mode in (1, 13, 14):
            a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
            b = torch.zeros_like(a)
            ops.gemm(dy, w, a, M, K, N, ops.BF16, b_rc=True, bv=view(K), resid=r, tr_mode=mode)
            ops.gemm(dy, wt, b, M, K, N, ops.BF16, bv=view(N), resid=r, tr_mode=mode)
            assert torch.equal(a, b), (M, N, K, mode)


def _attn_case(B, H, Tq, Tk, causal, drop, klen, bias, seed):
    import os
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    D, d = 64, H * 64
    g = torch.Generator(device="cpu").manual_seed(seed)
    q = (torch.randn(B * Tq, d, generator=g) * 0.7).to(dev, torch.bfloat16)
    kv = (torch.randn(B * Tk, 2 * d, generator=g) * 0.7).to(dev, torch.bfloat16)
    do = torch.randn(B * Tq, d, generator=g).to(dev, torch.bfloat16)
    kl = torch.tensor(klen, dtype=torch.int32, device=dev) if klen is not None else None
    bs = (torch.randn(H, Tq, Tk, generator=g) * 0.5).to(dev) if bias else None
    out = {}
    for v3 in ("0", "1"):          # 0: tile-staged kernels, 1: resident-operand forward + backward
        os.environ["SMX_ATTN_V3"] = v3
        desc = ops.AttnDesc(B, H, Tq, Tk, D, causal, D ** -0.5, bias=bs, drop=drop, klen=kl)
        desc.set("Q", q, 0, Tq * d, d); desc.set("K", kv, 0, Tk * 2 * d, 2 * d); desc.set("V", kv, d, Tk * 2 * d, 2 * d)
        o = torch.zeros(B * Tq, d, dtype=torch.bfloat16, device=dev)
        lse = torch.zeros(B * H * Tq, device=dev)
        delta = torch.zeros(B * H * Tq, device=dev)
        dq = torch.zeros_like(q); dkv = torch.zeros_like(kv)
        desc.set("O", o, 0, Tq * d, d); desc.set("dO", do, 0, Tq * d, d)
        desc.set("dQ", dq, 0, Tq * d, d); desc.set("dK", dkv, 0, Tk * 2 * d, 2 * d); desc.set("dV", dkv, d, Tk * 2 * d, 2 * d)
        ops.attention_fwd(desc, lse, ops.BF16)
        ops.attention_bwd(desc, lse, delta, ops.BF16)
        torch.cuda.synchronize()
        out[v3] = (o, lse, delta, dq, dkv)
    os.environ.pop("SMX_ATTN_V3", None)
    for name, a, b in zip(("o", "lse", "delta", "dq", "dkv"), out["0"], out["1"]):
        assert torch.equal(a, b), (name, B, H, Tq, Tk, causal, drop, klen, bias, (a.float() - b.float()).abs().max().item())
    assert out["1"][0].float().abs().sum().item() > 0


def test_resident_operand_attention_is_bit_identical_to_the_tile_staged_kernels():
    """attention_v3.h (a head's K / V - or Q / dO - resident in LDS, no per-tile barrier) against attention_v2.h: same tile bodies in the same
    order -> identical O, log-sum-exp, delta, dQ, dK, dV; self-attention at the encoders' lengths (499, 249), ragged lengths, cross shapes,
    causal, dropout (bit masks), per-clip key lengths and the T5 bias."""
    cases = [
        (2, 3, 499, 499, False, None, None, False), (2, 3, 499, 499, False, (0.1, 77), None, False), (3, 2, 249, 249, False, (0.1, 5), None, False),
        (2, 2, 131, 131, False, None, None, False), (2, 2, 512, 512, True, None, None, False), (2, 2, 300, 300, True, (0.2, 9), None, False),
        (2, 2, 499, 499, False, None, [499, 313], False), (2, 2, 249, 249, False, (0.1, 3), [100, 249], False),
        (2, 2, 200, 384, False, None, None, False), (1, 2, 257, 129, False, (0.1, 11), None, False), (2, 2, 256, 256, False, None, None, True),
        (1, 2, 130, 130, True, (0.1, 4), None, True),
    ]
    for i, c in enumerate(cases):
        _attn_case(*c, seed=100 + i)


def test_fused_adafactor_under_gradient_accumulation_with_layerdrop():
    """with k accumulated micro-batches (the reference's train.py default is 3) a LayerDrop-dropped layer is without a gradient only
    if EVERY micro-batch dropped it; FusedAdafactor used the last forward's draws and threw away the gradient of a layer kept in micro-batches
    1 - 2 and dropped in 3.  Four updates of three micro-batches each at layerdrop 0.5, against transformers' Adafactor on a twin whose
    all-dropped layers get `.grad = None` (what the HF model's autograd leaves), loss / 3 per micro-batch (a non-power-of-two device scalar:
    the fp32 scaling of the backward seed, model.py)."""
    transformers = pytest.importorskip("transformers")
    import contextlib, io
    from transformers.optimization import Adafactor
    from speechmix_amd.model import SpeechMixEED
    from speechmix_amd.optim import FusedAdafactor
    from tests.test_gpu_r5b import ENC, LM

    def build():
        with contextlib.redirect_stdout(io.StringIO()):
            return SpeechMixEED(dict(ENC, layerdrop=0.5), LM, down_scale=2, compute_dtype="fp32", init_seed=2).train()
    a, b = build(), build()
    hf = Adafactor([p for p in a.parameters() if p.requires_grad], lr=1e-2, scale_parameter=False, relative_step=False, warmup_init=False)
    a.store.external_updates = True
    fu = FusedAdafactor(b, lr=1e-2, max_grad_norm=0.0)
    g = torch.Generator().manual_seed(0)
    L = ENC["num_hidden_layers"]
    pre = a.engine.ep + "encoder.layers."
    late_drop = 0          # updates in which a layer was kept in an earlier micro-batch and dropped in the last one
    for step in range(4):
        sets = []
        for micro in range(3):
            wave = (torch.randn(3, 9000, generator=g) * 0.1).cuda()
            labels = torch.randint(4, 120, (3, 6), generator=g).cuda()
            losses = []
            for m in (a, b):
                torch.manual_seed(1000 + 10 * step + micro)          # the same LayerDrop draws for both models
                import numpy as np
                np.random.seed(1000 + 10 * step + micro)
                loss = m(wave, labels=labels)["loss"]
                (loss / 3).backward()
                losses.append(loss.item())
            assert a.engine.last_dropped == b.engine.last_dropped
            assert abs(losses[0] - losses[1]) <= 2e-4 * max(1.0, abs(losses[0])), (step, micro, losses)
            sets.append(set(a.engine.last_dropped))
        all_dropped = sets[0] & sets[1] & sets[2]
        late_drop += int(bool(sets[2] - all_dropped))
        assert b.engine.dropped_since_zero == all_dropped, (b.engine.dropped_since_zero, sets)
        for n, p in a.named_parameters():
            if n.startswith(pre) and int(n[len(pre):].split(".", 1)[0]) in all_dropped:
                p.grad = None
        hf.step()
        a.zero_grad(set_to_none=True)
        fu.step()
        b.zero_grad(set_to_none=True)
    assert late_drop >= 1, "the seeds never produced the case under test"
    torch.cuda.synchronize()
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for n in pa:
        if n.endswith("k_proj.bias"):
            continue          # (zero gradient in exact arithmetic: tests/test_gpu_r5b.py)
        x, y = pa[n].detach(), pb[n].detach()
        scale = max(x.abs().max().item(), 1e-3)
        assert (x - y).abs().max().item() <= 5e-4 * scale + 1e-7, (n, (x - y).abs().max().item(), scale)


def test_scale_dev_multiplies_in_fp32():
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    x = torch.randn(1024 * 50265 // 64 + 5, device=dev).bfloat16()
    s = torch.tensor(1.0 / 3.0, device=dev)
    want = (x.float() * s).bfloat16()
    y = x.clone()
    ops.scale_dev(y, s)
    assert torch.equal(y, want)
    z = torch.randn(1003, device=dev)
    w = z.clone()
    ops.scale_dev(w, s)
    assert torch.equal(w, z * s)


def test_graph_replay_with_the_speech_encoder_in_eval_mode_and_the_lm_in_train_mode():
    """`model.train(); model.encoder_model.eval()` - a frozen-encoder set-up - replays the captured step with the LM's dropout sites
    live; the replay must refresh the step key (it did so only when the ENCODER was in train mode, so every replayed step drew the masks of
    whatever key was set last).  Eager vs replayed: bit-identical gradients of the weight matrices step by step, and successive replayed
    steps differ."""
    import contextlib, io
    from speechmix_amd import graphs
    from speechmix_amd.model import SpeechMixEED
    from speechmix_amd.trainer import StepRunner
    from tests.test_gpu_r5 import ENC, LM, _compare

    def run(use_graphs, steps=9):
        graphs.MODE, graphs.ENABLED = "1", True
        g = torch.Generator().manual_seed(0)
        wave = (torch.randn(4, 12000, generator=g) * 0.1).cuda()
        labels = torch.randint(4, 200, (4, 6), generator=g).cuda()
        with contextlib.redirect_stdout(io.StringIO()):
            m = SpeechMixEED(ENC, LM, down_scale=2, compute_dtype="bf16", init_seed=0)
        m.train()
        m.encoder_model.eval()
        r = StepRunner(m, lr=0.0, optimizer="sgd", max_grad_norm=0.0, seed=5)
        r.use_graphs = use_graphs
        out = []
        for s in range(steps):
            loss = r.step(wave, labels)
            torch.cuda.synchronize()
            out.append(dict(grad=m.store.grad.clone(), dropped=list(m.engine.last_dropped), loss=float(loss.item()),
                            master=m.store.master.clone(), graphed=r._graphs is not None))
        return out, {n: (o, k, s) for n, (o, k, s) in m.store.offsets.items()}
    eager, offs = run(False)
    graph, _ = run(True)
    assert not any(s["graphed"] for s in eager) and graph[-1]["graphed"] and graph[-2]["graphed"]
    _compare(eager, graph, offs)
    assert not torch.equal(graph[-1]["grad"], graph[-2]["grad"])          # fresh LM dropout masks every replayed step
    assert not torch.equal(eager[-1]["grad"], eager[-2]["grad"])


def test_capture_abort_behind_a_fork_leaves_no_stream_capturing():
    """A capture pass that stops in the middle of backward - after the weight-gradient streams were forked - must end the capture cleanly:
    round 6 saw `~CUDAGraph: operation not permitted when stream is capturing` take a live-tuning bench process down (the forked streams
    stayed in capture mode after the failed end of capture).  Here the 5th data gradient of the first capture pass raises CaptureAbort; the
    runner falls back to eager, captures again a few steps later, replays, and everything can be destroyed."""
    import contextlib, gc, io, warnings
    from speechmix_amd import graphs, ops
    from speechmix_amd.model import SpeechMixEED
    from speechmix_amd.trainer import StepRunner
    from tests.test_gpu_r5 import ENC, LM
    graphs.MODE, graphs.ENABLED = "1", True
    g = torch.Generator().manual_seed(0)
    wave = (torch.randn(4, 12000, generator=g) * 0.1).cuda()
    labels = torch.randint(4, 200, (4, 6), generator=g).cuda()
    with contextlib.redirect_stdout(io.StringIO()):
        m = SpeechMixEED(ENC, LM, down_scale=2, compute_dtype="bf16", init_seed=0).train()
    r = StepRunner(m, lr=0.0, optimizer="sgd", max_grad_norm=0.0, seed=5)
    r.use_graphs = True
    eng = m.engine
    state = {"calls": 0, "raised": 0}
    orig = eng.dgrad

    def dgrad(*a, **k):
        if ops.CAPTURING and not state["raised"]:
            state["calls"] += 1
            if state["calls"] == 5:
                state["raised"] = 1
                raise ops.CaptureAbort("test: abort behind the fork")
        return orig(*a, **k)
    eng.dgrad = dgrad
    graphed = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(12):
            loss = r.step(wave, labels)
            torch.cuda.synchronize()
            assert torch.isfinite(loss).item(), step
            graphed.append(r._graphs is not None)
    assert state["raised"] == 1 and r._graph_failures == 1
    assert graphed[-1] and not all(graphed), graphed          # the second attempt captured; the steps in between ran eagerly
    for name in ("_side", "_wg_side", "_cs_stream", "_mask_stream"):
        st = getattr(eng, name, None)
        if st is not None:
            with torch.cuda.stream(st):
                assert not torch.cuda.is_current_stream_capturing(), name
    del r, m, eng
    gc.collect()
    torch.cuda.synchronize()


def test_optimizer_tail_beside_the_next_front_end_changes_no_bit():
    """SMX_OPT_OVERLAP (default on): Adafactor's statistics pass and the update of the front-end tensors on the compute stream, the update of
    every other tensor on a second stream that the next forward joins before its first encoder layer.  (a) The phased step against the
    one-launch-sequence step from the SAME parameters, gradients and optimizer state: identical parameters, compute copies and second-moment
    factors, three steps in a row (same kernels per tensor in the same order; only the streams differ).  (b) Through StepRunner, eager and
    replayed: the training runs (which differ run to run in the last bits anyway - fp32 atomics ahead of Adafactor's g / RMS(g)) stay
    together, and a state_dict() right after a step, without a device synchronisation, sees the final parameters."""
    import contextlib, io
    from speechmix_amd import graphs, ops
    from speechmix_amd.model import SpeechMixEED
    from speechmix_amd.trainer import StepRunner
    from tests.test_gpu_r5 import ENC, LM
    dev = torch.device("cuda:0")
    # (a) plan level
    torch.manual_seed(1)
    shapes = [(64, 10), (64,), (128, 64, 3), (128,), (300, 520), (520,), (1000, 128), (77,), (2048, 256), (256, 2048)]
    offs, off = [], 0
    for sh in shapes:
        n = 1
        for d in sh:
            n *= d
        offs.append((off, sh))
        off = (off + n + 63) // 64 * 64
    plans = [ops.AdafactorPlan(offs, dev) for _ in range(3)]
    p = [torch.randn(off, device=dev) * 0.1 for _ in range(3)]
    p[1].copy_(p[0])
    p[2].copy_(p[0])
    sh16 = [x.bfloat16() for x in p]
    side = torch.cuda.Stream()
    active = [True] * len(shapes)
    active[3] = False
    for step in range(3):
        g = torch.randn(off, device=dev) * (0.5 + step)
        plans[0].step(p[0], g, sh16[0], 1e-2, active=active, max_grad_norm=1.0)
        done = plans[1].step(p[1], g, sh16[1], 1e-2, active=active, max_grad_norm=1.0, split=(2, 6), tail_stream=side)
        assert done is not None
        torch.cuda.current_stream().wait_event(done)
        # third form: the statistics of everything outside the range taken EARLY on the second stream (the trainer starts them when the last
        # encoder layer's gradients are final), the rest as above
        o, nact = plans[2].prepare(p[2], g, sh16[2], 1e-2, active=active, max_grad_norm=1.0)
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            plans[2].early_stats(o, (2, 6))
        done2 = plans[2].finish(o, nact, (2, 6), side, early=True)
        torch.cuda.current_stream().wait_event(done2)
        torch.cuda.synchronize()
        for k in (1, 2):
            assert torch.equal(p[0], p[k]) and torch.equal(sh16[0], sh16[k]), (step, k)
            assert torch.equal(plans[0].row, plans[k].row) and torch.equal(plans[0].col, plans[k].col) and torch.equal(plans[0].rmean, plans[k].rmean)
    assert plans[1].tile0_of(0) == 0 and plans[1].tile0_of(len(shapes)) == plans[1].ntiles

    # (b) through the runner
    def run(overlap, use_graphs):
        graphs.MODE, graphs.ENABLED = "1", True
        g = torch.Generator().manual_seed(0)
        wave = (torch.randn(4, 12000, generator=g) * 0.1).cuda()
        labels = torch.randint(4, 200, (4, 6), generator=g).cuda()
        with contextlib.redirect_stdout(io.StringIO()):
            m = SpeechMixEED(ENC, LM, down_scale=2, compute_dtype="bf16", init_seed=0).train()
        r = StepRunner(m, lr=1e-3, optimizer="adafactor", max_grad_norm=1.0, seed=5)
        if not overlap:
            r._af_split = None
        else:
            assert r._af_split is not None and r._af_split[1] - r._af_split[0] < len(r.af_names)
            assert all(n.startswith("encoder_model.") and ".encoder.layers." not in n for n in r.af_names[r._af_split[0]:r._af_split[1]])
        r.use_graphs = use_graphs
        losses = []
        for s in range(8):
            losses.append(float(r.step(wave, labels).item()))
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}          # (no device synchronisation before it, on purpose)
        torch.cuda.synchronize()
        return losses, sd, m
    for use_graphs in (False, True):
        l0, _, _ = run(False, use_graphs)
        l1, sd, m = run(True, use_graphs)
        assert all(abs(a - b) <= 0.05 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (use_graphs, l0, l1)
        assert l1[-1] < l1[0]
        own = dict(m.named_parameters())
        for k, v in sd.items():
            if k in own:
                assert torch.equal(v, own[k].detach()), k
