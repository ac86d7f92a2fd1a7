"""Round-2 GPU parity tests (run on the MI355X: `pytest -m gpu`).

* fixtures made from the reference itself by tests/golden/make_golden_r2.py: trainable T5 (relative-position bias table
  gradients), a ragged -100-padded batch through the collator, the text prompt of ref:speechmix/model.py, the greedy
  label loop of ref:train.py:18-34 (BART and T5, KV-cached here);
* the oracle where no runnable reference exists (SpeechMixAdapter), and for the hooks of the drop-in boundary
  (`cal_loss` overrides, differentiable `decoder_model(...)`, autograd_param_inputs).
Tolerances: fp32 compute path <= 1e-3 on logits (north_star), gradients <= 3e-3 of the tensor's max; bf16 bounds are
3x what the path measures on these cases (printed by every test), see tests/test_gpu_fullsize_parity.py for the real
dimensions.
"""
import pytest
import torch

from tests.golden_util import load_case
from tests.test_gpu_e2e import _build, _err

pytestmark = pytest.mark.gpu


def _grads_vs_gold(model, gold, tol_grad, tag):
    named = dict(model.named_parameters())
    worst = 0.0
    for k, g in gold.items():
        if not k.startswith("grad::"):
            continue
        got = named[k[6:]].grad
        assert got is not None, k
        e = _err(got, g)
        scale = max(g.abs().max().item(), 1e-3)
        print(f"   [{tag}] grad {k[6:]}: err {e:.3e} (max {scale:.3e})")
        worst = max(worst, e / scale)
        # bf16: the T5 bias tables' gradients are sums of small score gradients of mixed sign (max 4e-3 here): measured
        # 8.9e-2 of the tensor's max on the MI355X, bound 3x that; every other tensor stays under tol_grad
        tol = max(tol_grad, 0.27) if ("relative_attention_bias" in k and tag == "bf16") else tol_grad
        assert e <= tol * scale, (k, e, scale)
    return worst


@pytest.mark.parametrize("dtype,tol,tol_grad", [("fp32", 1e-3, 3e-3), ("bf16", 6e-2, 8e-2)])
def test_trainable_t5_matches_reference_incl_relative_bias_gradients(dtype, tol, tol_grad):
    model, inp, gold, m = _build("eed_w2v2_t5_trainable", dtype)
    out = model(inp["input_values"], labels=inp["labels"], return_model_detail=True)
    e_log = _err(out["raw_logits"], gold["raw_logits"])
    e_loss = abs(out["loss"].item() - gold["loss"].item())
    print(f"[t5 trainable {dtype}] logits {e_log:.3e} loss {e_loss:.3e}")
    assert e_log < tol and e_loss < tol
    out["loss"].backward()
    _grads_vs_gold(model, gold, tol_grad, dtype)
    for side in ("encoder", "decoder"):
        p = dict(model.named_parameters())[f"decoder_model.{side}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        assert p.grad is not None and p.grad.abs().max().item() > 1e-5


@pytest.mark.parametrize("dtype,tol,tol_grad", [("fp32", 1e-3, 3e-3), ("bf16", 6e-2, 8e-2)])
def test_ragged_minus100_padded_batch_through_collator_matches_reference(dtype, tol, tol_grad):
    """clips of different length, the collator's -100 waveform padding convolved like audio (no mask
    reaches the encoder, ref:speechmix/model.py:148) and label rows with -100 tails (ref:train.py:100-133)."""
    from speechmix_amd.data import DataCollatorWithPadding, DevicePrefetcher
    model, inp, gold, m = _build("eed_ragged_batch", dtype)

    class Tok:
        pad_token_id, bos_token_id = m["lm_cfg"]["pad_token_id"], m["lm_cfg"]["bos_token_id"]
    feats = [{"input_values": inp[f"clip{i}"].numpy(), "labels": row} for i, row in enumerate(m["label_rows"])]
    batch = next(DevicePrefetcher([DataCollatorWithPadding(Tok())(feats)], model.device))
    assert batch["input_values"].is_cuda and batch["input_values"].shape == inp["input_values"].shape
    out = model(batch["input_values"], labels=batch["labels"], return_model_detail=True)
    e_enc = _err(out["encoder_last_hidden_state"], gold["encoder_last_hidden_state"])
    e_log = _err(out["raw_logits"], gold["raw_logits"])
    e_loss = abs(out["loss"].item() - gold["loss"].item())
    print(f"[ragged {dtype}] enc {e_enc:.3e} logits {e_log:.3e} loss {e_loss:.3e}")
    assert e_enc < tol and e_log < tol and e_loss < tol
    if dtype == "fp32":
        assert torch.equal(out["logits"].cpu(), gold["logits"])
    out["loss"].backward()
    _grads_vs_gold(model, gold, tol_grad, dtype)


def test_text_prompt_matches_reference_model_py():
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case("eed_route2_prompt")
    sd.pop("weights_sum", None)            # ref:speechmix/model.py keeps L (unused here) entries, the HF twin and this build L + 1
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32").eval()
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and res.missing_keys == ["weights_sum"], res
    out = model(inp["input_values"], labels=inp["labels"], input_text_prompt=inp["prompt_ids"], return_model_detail=True)
    assert _err(out["raw_logits"], gold["raw_logits"]) < 1e-3
    assert abs(out["loss"].item() - gold["loss"].item()) < 1e-4
    assert torch.equal(out["logits"].cpu(), gold["logits"])


@pytest.mark.parametrize("kind", ["bart", "t5"])
def test_greedy_label_creation_matches_reference_loop(kind):
    """ref:train.py:18-34 re-runs the LM per token; `generate_from_text` must emit the same ids through the KV cache."""
    from speechmix_amd.model import SpeechMixEED
    sd, inp, _, m = load_case(f"greedy_labels_{kind}")
    enc_cfg = load_case("eed_w2v2_bart")[3]["enc_cfg"]
    model = SpeechMixEED(enc_cfg, m["lm_cfg"], down_scale=2, compute_dtype="fp32").eval()
    res = model.decoder_model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    model.store.invalidate()
    model.decoder_model.config.max_length = int(inp["max_length"])
    got = model.generate_from_text(inp["gen_input"][None])
    print(f"[greedy {kind}] {got[0]}")
    assert got[0] == inp["predicted"].tolist()


def test_t5_cached_decode_steps_equal_full_recompute():
    from oracle import speechmix_oracle as O
    model, inp, gold, m = _build("eed_w2v2_t5_trainable", "fp32")
    lc = model.decoder_model.config
    wave = inp["input_values"]
    B, steps = wave.shape[0], 6
    forced = torch.randint(2, lc.vocab_size, (B, steps), generator=torch.Generator().manual_seed(3))
    eng = model.engine
    model.store.refresh_shadow()
    wv = model._prep_wave(wave)
    x, ssv = eng.speech_fwd(wv, B, wv.shape[1], False)
    e, S, _ = eng.bridge_fwd(x, B, ssv["T"])
    enc = eng.lm_encode(e, None, B, S)
    kept = []
    eng.greedy_decode(enc, B, S, steps, lc.decoder_start_token_id, -1, lc.pad_token_id, forced=forced.to(model.device),
                      keep_logits=kept)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    full = torch.cat([torch.full((B, 1), lc.decoder_start_token_id, dtype=torch.int64), forced[:, :-1]], 1)
    ref = O.speechmix_eed_forward(sd, m["enc_cfg"], m["lm_cfg"], wave, decoder_input_ids=full, down_scale=m["down_scale"])["raw_logits"]
    worst = max((kept[t].float().cpu() - ref[:, t]).abs().max().item() for t in range(steps))
    print(f"[t5] cached-step logits vs full recompute: max err {worst:.3e}")
    assert worst < 1e-3


@pytest.mark.parametrize("lm", ["eed_w2v2_bart", "eed_w2v2_t5_trainable"])
def test_speechmix_adapter_matches_oracle(lm):
    from oracle import speechmix_oracle as O
    from speechmix_amd.model import SpeechMixAdapter
    sd, inp, gold, m = load_case(lm)
    model = SpeechMixAdapter(m["enc_cfg"], m["lm_cfg"], down_scale=m["down_scale"], compute_dtype="fp32", adapter_seed=5).eval()
    model.load_state_dict(sd, strict=False)
    full = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in full.items()
              if not k.endswith(("embed_tokens.weight", "lm_head.weight", "nlp_emb.weight"))}
    ref = O.speechmix_eed_forward(leaves, m["enc_cfg"], m["lm_cfg"], inp["input_values"], labels=inp["labels"],
                                  down_scale=m["down_scale"])
    ref["loss"].backward()
    out = model(inp["input_values"], labels=inp["labels"], return_model_detail=True)
    e = _err(out["raw_logits"], ref["raw_logits"])
    print(f"[adapter {lm}] logits {e:.3e} loss {abs(out['loss'].item() - ref['loss'].item()):.3e}")
    assert e < 1e-3 and abs(out["loss"].item() - ref["loss"].item()) < 1e-4
    out["loss"].backward()
    named = dict(model.named_parameters())
    n_lm = m["lm_cfg"].get("encoder_layers") or m["lm_cfg"]["num_layers"]
    for name in ("adapters.0.1.weight", "adapters.1.3.bias", f"adapters.{n_lm}.0.weight", f"adapters.{2 * n_lm - 1}.3.weight",
                 "enc_to_dec_proj.weight", "encoder_model.encoder.layers.1.attention.q_proj.weight"):
        g = leaves[name].grad
        ee = _err(named[name].grad, g)
        assert ee <= 3e-3 * max(g.abs().max().item(), 1e-3), (name, ee)
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert frozen and all(named[n].grad is None for n in frozen)
    # greedy decoding goes through the adapters too
    got = model.generate(inp["input_values"], max_length=3)
    assert len(got) == inp["input_values"].shape[0]


def test_cal_loss_override_is_dispatched_and_differentiable():
    """ref:speechmix/model.py:172-173: forward ends in `self.cal_loss(...)`; a subclass override must be honoured."""
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case("eed_w2v2_bart")

    class Same(SpeechMixEED):                      # override that restates the reference's hook
        def cal_loss(self, inputs_embeds=None, attention_mask=None, decoder_input_ids=None, labels=None):
            self.called = True
            return self.decoder_model(inputs_embeds=inputs_embeds, decoder_input_ids=decoder_input_ids, labels=labels)

    class Extra(SpeechMixEED):                     # custom objective: CE + a term on the logits + one on inputs_embeds
        def cal_loss(self, inputs_embeds=None, attention_mask=None, decoder_input_ids=None, labels=None):
            out = self.decoder_model(inputs_embeds=inputs_embeds, decoder_input_ids=decoder_input_ids, labels=labels)
            out["loss"] = out["loss"] + 1e-3 * out["logits"].float().pow(2).mean() + 1e-2 * inputs_embeds.float().pow(2).mean()
            return out

    def run(cls):
        model = cls(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32").eval()
        model.load_state_dict(sd, strict=False)
        out = model(inp["input_values"], labels=inp["labels"], return_model_detail=True)
        out["loss"].backward()
        return model, out

    base, ob = run(SpeechMixEED)
    same, os_ = run(Same)
    assert same.called
    assert abs(os_["loss"].item() - gold["loss"].item()) < 1e-4
    assert torch.equal(os_["logits"].cpu(), gold["logits"])
    assert _err(os_["raw_logits"], gold["raw_logits"]) < 1e-3
    _grads_vs_gold(same, gold, 3e-3, "override")
    nb, ns = dict(base.named_parameters()), dict(same.named_parameters())
    for k in nb:
        if nb[k].grad is not None:
            assert torch.allclose(nb[k].grad, ns[k].grad, rtol=1e-4, atol=1e-7), k
    # custom objective vs the oracle + autograd
    from oracle import speechmix_oracle as O
    extra, oe = run(Extra)
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()
              if not k.endswith(("embed_tokens.weight", "lm_head.weight", "nlp_emb.weight"))}
    ref = O.speechmix_eed_forward(leaves, m["enc_cfg"], m["lm_cfg"], inp["input_values"], labels=inp["labels"], down_scale=2)
    loss = ref["loss"] + 1e-3 * ref["raw_logits"].pow(2).mean() + 1e-2 * ref["inputs_embeds"].pow(2).mean()
    loss.backward()
    assert abs(oe["loss"].item() - loss.item()) < 1e-4
    ne = dict(extra.named_parameters())
    for name in ("enc_to_dec_proj.weight", "decoder_model.model.shared.weight", "length_adapters.0.weight",
                 "encoder_model.encoder.layers.1.attention.q_proj.weight", "decoder_model.model.decoder.layers.1.fc2.bias"):
        g = leaves[name].grad
        e = _err(ne[name].grad, g)
        assert e <= 3e-3 * max(g.abs().max().item(), 1e-3), (name, e)


def test_autograd_param_inputs_gradients_equal_flat_buffer_gradients():
    """INTEGRATION.md's HF-Trainer / DDP mode: every parameter is an input of the step's autograd node, gradients arrive
    through autograd (accumulating into .grad like any torch module) and must equal the flat-buffer gradients."""
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    flat = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32").eval()
    flat.load_state_dict(sd, strict=False)
    flat(inp["input_values"], labels=inp["labels"])["loss"].backward()
    ag = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32", autograd_param_inputs=True).eval()
    ag.load_state_dict(sd, strict=False)
    hooks = []
    p0 = ag.enc_to_dec_proj.weight
    p0.register_hook(lambda g: hooks.append(g.shape))                 # what DDP's reducer relies on
    loss = ag(inp["input_values"], labels=inp["labels"])["loss"]
    grads = torch.autograd.grad(loss, [p for p in ag.parameters() if p.requires_grad], allow_unused=True)
    names = [n for n, p in ag.named_parameters() if p.requires_grad]
    fn = dict(flat.named_parameters())
    n_checked = 0
    for n, g in zip(names, grads):
        if fn[n].grad is None:
            continue
        assert g is not None, n
        assert torch.allclose(g, fn[n].grad, rtol=1e-5, atol=1e-8), n
        n_checked += 1
    assert n_checked > 100
    ag.zero_grad(set_to_none=True)
    ag(inp["input_values"], labels=inp["labels"])["loss"].backward()
    assert hooks, "autograd hooks on parameters must fire in this mode"
    _grads_vs_gold(ag, gold, 3e-3, "autograd_param_inputs")


def test_position_table_overrun_raises_like_hf():
    """positions past max_position_embeddings must raise IndexError (HF does), not read / write the next tensor."""
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    lm_cfg = dict(m["lm_cfg"], max_position_embeddings=10)
    model = SpeechMixEED(m["enc_cfg"], lm_cfg, down_scale=2, compute_dtype="fp32").eval()
    with pytest.raises(IndexError):
        model(inp["input_values"], labels=inp["labels"])                  # S = 12 > 10
    model2 = SpeechMixEED(m["enc_cfg"], dict(m["lm_cfg"], max_position_embeddings=16), down_scale=2, compute_dtype="fp32").eval()
    model2(inp["input_values"], labels=inp["labels"])                     # S = 12, L = 6: fits
    with pytest.raises(IndexError):
        model2.generate(inp["input_values"], max_length=40)
    with pytest.raises(IndexError):
        model2(inp["input_values"], labels=torch.full((2, 6), 500))         # label id >= vocab


def test_allreduce_bucket_c_entry_runs_a_real_rccl_collective():
    """smx_allreduce_bucket(ncclComm_t, buf, n, dtype, stream) with a communicator created the way a
    C host would (ncclGetUniqueId + ncclCommInitRank, one rank on this GPU), on a side stream: the in-place sum over one
    rank must leave the bucket unchanged, for fp32 and bf16 buckets of the flat gradient buffer."""
    import ctypes as C
    from speechmix_amd import _lib as L
    lib = L.lib()
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("librccl.so")

    class UID(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UID()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        fn = lib.smx_allreduce_bucket
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        side = torch.cuda.Stream()
        for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
            x = torch.randn(1 << 20, device="cuda").to(dt)
            ref = x.clone()
            side.wait_stream(torch.cuda.current_stream())
            assert fn(comm, C.c_void_p(x.data_ptr()), x.numel(), code, C.c_void_p(side.cuda_stream)) == 0
            side.synchronize()
            assert torch.equal(x, ref)
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
