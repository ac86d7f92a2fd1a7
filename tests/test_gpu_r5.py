"""Round-5 GPU tests (run on the MI355X: `pytest -m gpu`).

* the library's STEP KEY (csrc/smx_common.h, `smx_set_step_key`): every kernel that hashes a dropout mask uses (seed + key) - a
  launch with (seed s, key k) must equal, bit for bit, the same launch with (seed s + k, key 0), for every kernel family that
  hashes (stand-alone dropout, the dropout + column-sum pass, all five bf16 GEMM kernels' epilogues + the fp32 GEMM, the split-K
  epilogue, LayerNorm forward / backward incl. the fused masked-dx output, the attention mask generator and the fp32 attention);
* a training step replayed from captured HIP graphs (speechmix_amd/graphs.py) against the eager step: same seeds, same host
  streams -> the same LayerDrop / SpecAugment decisions and bit-identical weight-matrix gradients, step after step, with
  dropout on; AdamW-updated parameters stay bit-identical too; eval mode as well;
* the time-blocked positional conv (J = 4) against the plain form on the fp32 path.
"""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _key(k):
    from speechmix_amd import ops
    ops.set_step_key(k)


@pytest.fixture(autouse=True)
def _reset_key():
    yield
    if torch.cuda.is_available():
        _key(0)


def _both(run, seed, key):
    """run(seed) -> list of output tensors; (seed, key) vs (seed + key, 0)."""
    _key(key)
    a = [t.clone() for t in run(seed)]
    _key(0)
    b = [t.clone() for t in run((seed + key) & 0xffffffff)]
    b2 = [t.clone() for t in run((seed + key) & 0xffffffff)]
    c = [t.clone() for t in run(seed)]
    torch.cuda.synchronize()
    for i, (x, y, y2) in enumerate(zip(a, b, b2)):
        if torch.equal(y, y2):
            assert torch.equal(x, y), i
        else:                                        # (an output that is not bit-reproducible run to run: fp32 atomics)
            assert torch.allclose(x.float(), y.float(), rtol=1e-5, atol=1e-5 * float(y.float().abs().max())), i
    assert any(not torch.equal(x, z) for x, z in zip(a, c))          # and the key really changes the mask


def test_step_key_dropout_kernels():
    from speechmix_amd import ops
    g = torch.Generator().manual_seed(0)
    for dt, tdt in ((ops.BF16, torch.bfloat16), (ops.F32, torch.float32)):
        x = torch.randn(1000, 768, generator=g).to(DEV, tdt)

        def run(seed):
            out = torch.empty_like(x)
            ops.dropout(x, out, x.numel(), 0.1, seed, dt)
            out2 = torch.empty_like(x)
            cs = torch.zeros(768, dtype=torch.float32, device=DEV)
            ops.dropout_colsum(x, out2, 1000, 768, 0.1, seed, cs, dt)
            return [out, out2, cs]
        _both(run, 12345, 0x9e3779b9)


@pytest.mark.parametrize("mode", [1, 8, 9, 11, 12, 13, "f32", "splitk"])
def test_step_key_gemm_epilogues(mode):
    from speechmix_amd import ops
    g = torch.Generator().manual_seed(1)
    M, N, K = 2048, 768, 512
    f32 = mode == "f32"
    dt, tdt = (ops.F32, torch.float32) if f32 else (ops.BF16, torch.bfloat16)
    a = torch.randn(M, K, generator=g).to(DEV, tdt)
    b = (torch.randn(N, K, generator=g) * 0.05).to(DEV, tdt)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV, tdt)

    def run(seed):
        outs = []
        # linear + dropout + residual; activation + saved pre-activation + dropout
        for kw in (dict(resid=res), dict(act=ops.ACT_GELU, aux_out=torch.empty(M, N, dtype=tdt, device=DEV))):
            c = torch.empty(M, N, dtype=tdt, device=DEV)
            if mode == "splitk":
                slabs = torch.empty(2 * M * N, dtype=torch.float32, device=DEV)
                ops.gemm_splitk(a, b, c, M, N, K, dt, 2, slabs, bias=bias, drop=(0.1, seed), **kw)
            else:
                ops.gemm(a, b, c, M, N, K, dt, bias=bias, drop=(0.1, seed), **({} if f32 else dict(tr_mode=mode)), **kw)
            outs.append(c)
            if "aux_out" in kw:
                outs.append(kw["aux_out"])
        if not f32 and mode != "splitk":
            # saved-derivative form (forward of an FFN's first Linear in bf16): the side tensor carries the mask
            aux = torch.empty(M, N, dtype=tdt, device=DEV)
            c = torch.empty(M, N, dtype=tdt, device=DEV)
            try:
                ops.gemm(a, b, c, M, N, K, dt, bias=bias, drop=(0.1, seed), act=ops.ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=aux, tr_mode=mode)
                outs += [c, aux]
            except RuntimeError:
                pass                                  # (a class this variant is not instantiated for)
        return outs
    _both(run, 777, 0x01234567)


def test_step_key_norm_and_attention():
    from speechmix_amd import ops
    g = torch.Generator().manual_seed(2)
    M, D = 3000, 768
    for dt, tdt in ((ops.BF16, torch.bfloat16), (ops.F32, torch.float32)):
        x = torch.randn(M, D, generator=g).to(DEV, tdt)
        dy = torch.randn(M, D, generator=g).to(DEV, tdt)
        gamma, beta = torch.randn(D, generator=g).to(DEV), torch.randn(D, generator=g).to(DEV)

        def run(seed):
            y = torch.empty_like(x)
            mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
            ops.norm_fwd(x, y, gamma, beta, mean, rstd, M, D, dt, drop=(0.1, seed))
            dx = torch.empty_like(x)
            dgm, dbt = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
            folds = ops.FoldQueue()
            dxd = torch.empty_like(x)
            gb2 = torch.zeros(D, device=DEV)
            ops.norm_bwd(dy, x, dx, gamma, beta, mean, rstd, dgm, dbt, M, D, dt, drop=(0.1, seed), folds=folds,
                         drop2=(0.1, (seed + 0x55) & 0xffffffff), dx_drop=dxd, gb2=gb2)
            folds.flush()
            return [y, dx, dxd, dgm, gb2]
        _both(run, 4242, 0x0badf00d)
    # attention: the bf16 path's bit masks, the fp32 path's in-kernel hash
    B, H, T, hd = 2, 4, 200, 64
    d = H * hd
    for dt, tdt in ((ops.BF16, torch.bfloat16), (ops.F32, torch.float32)):
        qkv = torch.randn(B * T, 3 * d, generator=g).to(DEV, tdt)

        def run(seed):
            desc = ops.AttnDesc(B, H, T, T, hd, False, hd ** -0.5, drop=(0.2, seed))
            desc.set("Q", qkv, 0, T * 3 * d, 3 * d)
            desc.set("K", qkv, d, T * 3 * d, 3 * d)
            desc.set("V", qkv, 2 * d, T * 3 * d, 3 * d)
            o = torch.empty(B * T, d, dtype=tdt, device=DEV)
            desc.set("O", o, 0, T * d, d)
            lse = torch.empty(B * H * T, device=DEV)
            ops.attention_fwd(desc, lse, dt)
            return [o]
        _both(run, 99, 0x7777)


ENC = dict(model_type="wav2vec2", hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256,
           conv_dim=[64] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2], num_conv_pos_embeddings=16,
           num_conv_pos_embedding_groups=4, layerdrop=0.3)
LM = dict(model_type="bart", vocab_size=200, d_model=128, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
          decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256, max_position_embeddings=128)


def _run_steps(use_graphs, steps, train=True, optimizer="sgd", lr=0.0, enc=ENC, lm=LM, batches=None, mode="1"):
    from speechmix_amd import graphs
    from speechmix_amd.model import SpeechMixEED
    from speechmix_amd.trainer import StepRunner
    graphs.MODE, graphs.ENABLED = mode, True          # "1": replay from the capture on (the default, "auto", times both and keeps the faster)
    g = torch.Generator().manual_seed(0)
    wave = (torch.randn(4, 12000, generator=g) * 0.1).cuda()
    labels = torch.randint(4, 200, (4, 6), generator=g).cuda()
    with contextlib.redirect_stdout(io.StringIO()):
        m = SpeechMixEED(enc, lm, down_scale=2, compute_dtype="bf16", init_seed=0)
    m.train(train)
    r = StepRunner(m, lr=lr, optimizer=optimizer, max_grad_norm=0.0 if optimizer == "sgd" else 1.0, seed=5)
    r.use_graphs = use_graphs
    out = []
    for s in range(steps):
        w, lab = (wave, labels) if batches is None else batches[s % len(batches)]
        loss = r.step(w, lab)
        torch.cuda.synchronize()
        out.append(dict(grad=m.store.grad.clone(), dropped=list(m.engine.last_dropped), loss=float(loss.item()),
                        master=m.store.master.clone(), graphed=r._graphs is not None))
    return out, {n: (o, k, s) for n, (o, k, s) in m.store.offsets.items()}, r


def _compare(a, b, offs, exact_params=False):
    dropped_any = False
    for step, (x, y) in enumerate(zip(a, b)):
        assert x["dropped"] == y["dropped"], step
        dropped_any |= bool(x["dropped"])
        assert abs(x["loss"] - y["loss"]) <= 2e-6 * max(1.0, abs(x["loss"])), (step, x["loss"], y["loss"])
        for name, (o, k, shape) in offs.items():
            ga, gb = x["grad"][o:o + k], y["grad"][o:o + k]
            if len(shape) == 2 and "shared" not in name and "embed_positions" not in name:
                assert torch.equal(ga, gb), (step, name)            # weight matrices: no atomics anywhere on their path
            else:
                assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-6 * max(1.0, float(ga.abs().max()))), (step, name)
            if any(name.startswith(f"encoder_model.encoder.layers.{i}.") for i in y["dropped"]):
                assert not gb.any(), (step, name)
    return dropped_any


def test_graph_replayed_training_steps_equal_eager_steps_bit_for_bit():
    """Train mode (dropout 0.1 everywhere, LayerDrop 0.3, SpecAugment): 10 steps eager vs 10 steps of which the last 7 are
    replayed from the captured chain.  lr = 0 keeps the weights, so every step's gradient can be compared."""
    eager, offs, _ = _run_steps(False, 10)
    graph, _, r = _run_steps(True, 10)
    assert not any(s["graphed"] for s in eager)
    assert [s["graphed"] for s in graph] == [False] * 3 + [True] * 7, [s["graphed"] for s in graph]
    assert _compare(eager, graph, offs)             # (LayerDrop did drop layers in the replayed steps)
    # successive replayed steps draw different dropout masks (the step key moves; the seeds are baked)
    assert not torch.equal(graph[-1]["grad"], graph[-2]["grad"])
    names = [n for n, _ in r._graphs.graphs]
    assert names[0] == "front" and "stage:lm" in names and names.count("tail") == 1 and len(names) == 2 * 4 + 6


def test_graph_replayed_steps_with_an_optimizer_and_in_eval_mode():
    """SGD with a real learning rate: parameters after 8 steps agree (the weights move, so every later step depends on every
    earlier one; SGD is linear in the gradient, so the fp32 atomics' last-bit noise in the embedding / bias gradients stays
    last-bit noise - Adam would turn it into lr-sized differences wherever a gradient is ~0); eval mode (no dropout, no
    LayerDrop): replay == eager as well."""
    eager, offs, _ = _run_steps(False, 8, optimizer="sgd", lr=2e-2)
    graph, _, _ = _run_steps(True, 8, optimizer="sgd", lr=2e-2)
    assert graph[-1]["graphed"]
    assert (eager[0]["master"] - eager[-1]["master"]).abs().max().item() > 1e-3          # the weights did move
    d = (eager[-1]["master"] - graph[-1]["master"]).abs().max().item()
    assert d <= 2e-5, d
    assert abs(eager[-1]["loss"] - graph[-1]["loss"]) <= 1e-3
    e2, offs, _ = _run_steps(False, 6, train=False)
    g2, _, _ = _run_steps(True, 6, train=False)
    assert g2[-1]["graphed"] and not _compare(e2, g2, offs)


def test_capture_tolerates_event_queries_from_another_thread():
    """torch.distributed's NCCL watchdog thread polls its work events (hipEventQuery) every few hundred milliseconds.  Under the
    default capture mode such a query from ANY thread while a capture is open fails with hipErrorStreamCaptureUnsupported - it killed
    tools/gpu_dist_single.py once in four runs - so the step is captured thread-locally (graphs.StepGraphs._begin).  Here a thread
    queries an event in a tight loop while steps are captured and replayed: no error in the thread, the capture succeeds."""
    import threading
    ev, s2 = torch.cuda.Event(), torch.cuda.Stream()
    with torch.cuda.stream(s2):          # (like a collective's end event: recorded behind real work on a stream of its own)
        torch.zeros(1 << 20, device=DEV).add_(1)
        ev.record()
    torch.cuda.synchronize()
    stop, errors, polls = threading.Event(), [], [0]

    def poll():
        torch.cuda.set_device(0)
        while not stop.is_set():
            try:
                ev.query()
                polls[0] += 1
            except Exception as e:          # noqa: BLE001
                errors.append(repr(e))
                return

    t = threading.Thread(target=poll, daemon=True)
    t.start()
    try:
        graph, _, r = _run_steps(True, 6)
    finally:
        stop.set()
        t.join(10)
    assert not errors, errors[:1]
    assert polls[0] > 100
    assert graph[-1]["graphed"] and r._graph_failures == 0


def test_graph_replay_follows_new_inputs_and_falls_back_on_a_new_shape():
    """The graphs read static input tensors that every replay refills; another batch shape drops the captured chain and runs
    eagerly (then captures the new configuration after its warm-up)."""
    g = torch.Generator().manual_seed(3)
    batches = [((torch.randn(4, 12000, generator=g) * 0.1).cuda(), torch.randint(4, 200, (4, 6), generator=g).cuda()) for _ in range(3)]
    eager, offs, _ = _run_steps(False, 9, batches=batches)
    graph, _, _ = _run_steps(True, 9, batches=batches)
    assert graph[-1]["graphed"]
    _compare(eager, graph, offs)
    assert not torch.equal(graph[-1]["grad"], graph[-2]["grad"])
    other = [((torch.randn(2, 9000, generator=g) * 0.1).cuda(), torch.randint(4, 200, (2, 5), generator=g).cuda())]
    mixed = batches * 2 + other * 5
    eager, offs, _ = _run_steps(False, len(mixed), batches=mixed)
    graph, _, _ = _run_steps(True, len(mixed), batches=mixed)
    assert [s["graphed"] for s in graph] == [False] * 3 + [True] * 3 + [False] * 3 + [True] * 2
    _compare(eager, graph, offs)


def test_graph_replay_of_speechmixself_with_a_t5_teacher_pass_and_of_a_weighted_sum_model():
    """The other model families through the captured chain: SpeechMixSelf (wav2vec2 -> frozen T5 in eval mode, text teacher pass,
    CE + KLD + MSE; relative-position buckets cached on the device; only the speech side's dropout is on) in train mode - 9 steps
    eager vs replayed, weight-matrix gradients bit-identical - and a weighted-sum SpeechMixEED in EVAL mode (train mode of such
    models stays eager: a LayerDrop-dropped layer's share of the weighted gradient is added outside its graph)."""
    from speechmix_amd import graphs
    from speechmix_amd.model import SpeechMixEED, SpeechMixSelf
    from speechmix_amd.trainer import StepRunner
    from tests.golden_util import load_case
    graphs.MODE, graphs.ENABLED = "1", True
    sd, inp, gold, m = load_case("self_w2v2_t5")
    res = {}
    for use in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):
            model = SpeechMixSelf(m["enc_cfg"], m["lm_cfg"], share_layer_ratio=0.5, down_scale=4, compute_dtype="bf16")
        model.load_state_dict(sd, strict=False)
        model.train()
        r = StepRunner(model, lr=0.0, optimizer="sgd", max_grad_norm=0.0, seed=9)
        r.use_graphs = use
        steps = []
        for _ in range(9):
            loss = r.step(inp["input_values"], inp["labels"], text_input_ids=inp["text_input_ids"])
            torch.cuda.synchronize()
            steps.append(dict(grad=model.store.grad.clone(), dropped=list(model.engine.last_dropped), loss=float(loss.item()),
                              graphed=r._graphs is not None))
        res[use] = (steps, {n: (o, k, sh) for n, (o, k, sh) in model.store.offsets.items()})
    assert [s["graphed"] for s in res[True][0]] == [False] * 3 + [True] * 6
    _compare(res[False][0], res[True][0], res[True][1])
    assert res[True][0][-1]["grad"].abs().max().item() > 0
    # weighted sum, eval mode
    res = {}
    g = torch.Generator().manual_seed(0)
    wave = (torch.randn(4, 12000, generator=g) * 0.1).cuda()
    labels = torch.randint(4, 200, (4, 6), generator=g).cuda()
    for use in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):
            model = SpeechMixEED(ENC, LM, down_scale=2, compute_dtype="bf16", init_seed=0, weighted_sum=True).eval()
        r = StepRunner(model, lr=0.0, optimizer="sgd", max_grad_norm=0.0, seed=9)
        r.use_graphs = use
        steps = []
        for _ in range(6):
            loss = r.step(wave, labels)
            torch.cuda.synchronize()
            steps.append(dict(grad=model.store.grad.clone(), dropped=[], loss=float(loss.item()), graphed=r._graphs is not None))
        res[use] = (steps, {n: (o, k, sh) for n, (o, k, sh) in model.store.offsets.items()})
    assert res[True][0][-1]["graphed"]
    _compare(res[False][0], res[True][0], res[True][1])
    o, k, _ = res[True][1]["weights_sum"]
    assert res[True][0][-1]["grad"][o:o + k].abs().max().item() > 0          # the layer weights did receive a gradient


def test_auto_mode_times_both_and_keeps_one():
    """SMX_STEP_GRAPHS=auto (the default): 3 eager steps, the capture, T replayed + T eager timed steps (T = graphs.TRIAL_STEPS, whole steps
    between HIP events), then ONE mode for good - the replayed one only if it wins by 2 %; whatever it picks, every step equals the eager run."""
    from speechmix_amd import graphs
    T = graphs.TRIAL_STEPS
    n = 3 + 2 * T + 5
    eager, offs, _ = _run_steps(False, n)
    auto, _, r = _run_steps(True, n, mode="auto")
    assert [s["graphed"] for s in auto][:3 + 2 * T] == [False] * 3 + [True] * (2 * T)          # (captured chain alive through the trial)
    assert r.graph_trial_ms is not None and set(r.graph_trial_ms) == {"replay", "eager"}
    choice = list(r._graph_choice.values())
    assert choice in (["eager"], ["replay"])
    assert all(s["graphed"] == (choice == ["replay"]) for s in auto[3 + 2 * T + 1:])
    _compare(eager, auto, offs)


@pytest.mark.parametrize("Cg,T,train", [(48, 499, True), (64, 250, True), (48, 203, False)])
def test_time_blocked_positional_conv_equals_the_plain_form_in_fp32(Cg, T, train):
    """the J = 4 time-blocked positional conv (shifted-tap pack, group-major unpack, folded weight gradient) is the
    bf16 default but the fp32 parity tests run J = 1.  SMX_POSCONV_J_F32=1 runs the blocked form in fp32: forward, dh, dg, dv
    against the plain form at fp32 tolerance; T % J != 0; parameters trainable and frozen."""
    from speechmix_amd.model import SpeechMixEED
    G = 4
    d = G * Cg
    enc = dict(ENC, hidden_size=d, num_attention_heads=2, intermediate_size=64, num_hidden_layers=1, num_conv_pos_embeddings=128,
               num_conv_pos_embedding_groups=G, layerdrop=0.0)
    res = {}
    for J in ("0", "1"):
        os.environ["SMX_POSCONV_J_F32"] = J
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                m = SpeechMixEED(enc, dict(LM, d_model=64, encoder_ffn_dim=64, decoder_ffn_dim=64), down_scale=2, compute_dtype="fp32",
                                 init_seed=0).eval()
            m._need_engine()
            eng = m.engine
            pre = "encoder_model.encoder.pos_conv_embed.conv."
            for n, p in m.named_parameters():
                if n.startswith(pre):
                    p.requires_grad_(train)
            m.store.refresh_shadow(force=True)
            g = torch.Generator().manual_seed(7)
            B = 3
            h = torch.randn(B * T, d, generator=g).cuda()
            ds = torch.randn(B * T, d, generator=g).cuda()
            eng.begin_grads(zero=True, lazy=False)
            s, sv = eng.posconv_fwd(h, B, T)
            assert sv["J"] == (4 if J == "1" else 1)
            dh = eng.posconv_bwd(ds, sv)
            if eng.folds is not None:
                eng.folds.flush()
            torch.cuda.synchronize()
            res[J] = dict(s=s.clone(), dh=dh.clone(), g={n: eng.G(n).clone() for n in m.store.offsets if n.startswith(pre)})
        finally:
            os.environ.pop("SMX_POSCONV_J_F32", None)
    a, b = res["0"], res["1"]
    for k in ("s", "dh"):
        scale = a[k].abs().max().item()
        assert (a[k] - b[k]).abs().max().item() <= 2e-5 * scale, k
    for n in a["g"]:
        scale = max(a["g"][n].abs().max().item(), 1e-6)
        if train:
            assert a["g"][n].abs().max().item() > 0, n
        assert (a["g"][n] - b["g"][n]).abs().max().item() <= 5e-5 * scale, n
