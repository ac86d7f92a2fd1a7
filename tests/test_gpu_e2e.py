"""GPU parity of the drop-in classes against the golden fixtures (outputs of the reference itself) and the
CPU oracle.  Tolerances: fp32 compute path - logits within 1e-3 of the reference (north_star), in practice
~1e-5; bf16 compute path - stated separately below (bf16 storage of every activation)."""
import math

import pytest
import torch

from tests.golden_util import load_case

pytestmark = pytest.mark.gpu


def _build(case, dtype, **kw):
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case(case)
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], share_layer_ratio=m.get("share_layer_ratio", 0),
                         down_scale=m["down_scale"], compute_dtype=dtype, **kw)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # (masked_spec_embed: HF 5.x creates it only when mask_time_prob > 0; this model always has it for wav2vec2, like HF 4.x)
    assert all(k in ("weights_sum", "encoder_model.masked_spec_embed") for k in missing), missing
    model.eval()
    return model, inp, gold, m


def _err(a, b):
    return (a.detach().float().cpu() - b.float()).abs().max().item()


@pytest.mark.parametrize("case", ["eed_w2v2_bart", "eed_hubert_mbart"])
# bf16 bounds = 3 x what the path measures on these fixtures on the MI355X (round 2: activations <= 5.5e-2, logits <= 3.9e-3,
# loss <= 1.9e-4, worst gradient 3.5e-2 of its tensor's max); the same rule at the real dimensions: test_gpu_fullsize_parity.py
@pytest.mark.parametrize("dtype,tol_act,tol_logit,tol_grad", [("fp32", 2e-4, 1e-3, 2e-3), ("bf16", 1.6e-1, 1.2e-2, 1.1e-1)])
def test_eed_forward_backward_matches_reference(case, dtype, tol_act, tol_logit, tol_grad):
    model, inp, gold, m = _build(case, dtype)
    out = model(inp["input_values"], labels=inp["labels"], return_model_detail=True)
    assert out["logits"].shape == inp["labels"].shape and out["logits"].dtype == torch.int64
    e_enc = _err(out["encoder_last_hidden_state"], gold["encoder_last_hidden_state"])
    e_emb = _err(out["inputs_embeds"], gold["inputs_embeds"])
    e_lme = _err(out["lm_encoder_last_hidden"], gold["lm_encoder_last_hidden"])
    e_log = _err(out["raw_logits"], gold["raw_logits"])
    e_loss = abs(out["loss"].item() - gold["loss"].item())
    print(f"[{case} {dtype}] enc {e_enc:.3e} emb {e_emb:.3e} lm_enc {e_lme:.3e} logits {e_log:.3e} loss {e_loss:.3e}")
    assert e_enc < tol_act and e_emb < tol_act and e_lme < tol_act
    assert e_log < tol_logit
    assert e_loss < tol_logit
    if dtype == "fp32":
        assert torch.equal(out["logits"].cpu(), gold["logits"])
    out["loss"].backward()
    named = dict(model.named_parameters())
    worst = 0.0
    bad = []
    for k, g in gold.items():
        if not k.startswith("grad::"):
            continue
        name = k[6:]
        got = named[name].grad
        assert got is not None, name
        e = _err(got, g)
        scale = max(g.abs().max().item(), 1e-3)
        print(f"   grad {name}: err {e:.3e} (max {scale:.3e})")
        worst = max(worst, e / scale)
        if e > tol_grad * scale:
            bad.append((name, e, scale))
    assert not bad, bad
    assert worst > 0


def test_eval_no_labels_and_shapes():
    model, inp, gold, m = _build("eed_w2v2_bart", "fp32")
    with torch.no_grad():
        out = model(inp["input_values"], return_model_detail=True)
    assert "loss" not in out
    assert out["logits"].shape == (2, 1)
    T = m["enc_cfg"]["hidden_size"]
    assert tuple(out["shape_before_length_adapter"]) == (2, 24, 64)
    assert tuple(out["shape_before_enc_dec_projector"]) == (2, 12, 64)
    # list-of-1D-tensors input (ref:test/test_hf_model.py:40)
    out2 = model([inp["input_values"][0], inp["input_values"][1]])
    assert torch.equal(out2["logits"], out["logits"])


def test_decoder_model_callable_like_reference_label_loop():
    """ref:train.py:18-34 calls decoder_model(input_ids=..., decoder_input_ids=...).logits"""
    from oracle import speechmix_oracle as O
    model, inp, gold, m = _build("eed_w2v2_bart", "fp32")
    sd, _, _, _ = load_case("eed_w2v2_bart")
    _, lm_sd, _ = O.split_state_dict(sd)
    ids = torch.randint(4, 128, (2, 7))
    dec = torch.randint(4, 128, (2, 3))
    ref, _ = O.lm_forward(lm_sd, m["lm_cfg"], input_ids=ids, decoder_input_ids=dec)
    got = model.decoder_model(input_ids=ids, decoder_input_ids=dec).logits
    assert _err(got, ref) < 1e-3


def test_frozen_lm_gets_no_grads_and_requires_grad_toggle():
    from speechmix_amd.model import SpeechMixFixed
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    model = SpeechMixFixed(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32")
    model.load_state_dict(sd, strict=False)
    assert all(not p.requires_grad for p in model.decoder_model.parameters())
    out = model(inp["input_values"], labels=inp["labels"])
    out["loss"].backward()
    assert model.enc_to_dec_proj.weight.grad is not None
    assert all(p.grad is None for p in model.decoder_model.parameters())
    e = _err(model.enc_to_dec_proj.weight.grad, gold["grad::enc_to_dec_proj.weight"])
    assert e < 2e-3 * gold["grad::enc_to_dec_proj.weight"].abs().max().item() + 1e-6


# bf16: measured logits 2.9e-2, loss 4.8e-3, gradients 1.6e-2 of max -> bounds 3 x
@pytest.mark.parametrize("dtype,tol,tol_grad", [("fp32", 1e-3, 3e-3), ("bf16", 9e-2, 5e-2)])
def test_speechmix_self_t5_losses_and_grads(dtype, tol, tol_grad):
    """SpeechMixSelf (wav2vec2 + T5, share 0.5, ds 4): CE + KLD + MSE and gradients vs the reference's cal_loss."""
    from speechmix_amd.model import SpeechMixSelf
    sd, inp, gold, m = load_case("self_w2v2_t5")
    model = SpeechMixSelf(m["enc_cfg"], m["lm_cfg"], share_layer_ratio=0.5, down_scale=4, compute_dtype=dtype).eval()
    model.load_state_dict(sd, strict=False)
    assert model.speech_encoder_layer == m["speech_encoder_layer"]
    assert len([n for n in model.list_no_grad if n.startswith("decoder_model.")]) > 0
    assert all(not p.requires_grad for p in model.decoder_model.parameters())
    out = model(inp["input_values"], labels=inp["labels"], text_input_ids=inp["text_input_ids"], return_model_detail=True)
    e_emb = _err(out["inputs_embeds"], gold["inputs_embeds"])
    e_log = _err(out["raw_logits"], gold["raw_logits"])
    e_loss = abs(out["loss"].item() - gold["loss"].item())
    e_ce = abs(out["ce_loss"].item() - gold["ce"].item())
    print(f"[self_w2v2_t5 {dtype}] emb {e_emb:.3e} logits {e_log:.3e} loss {e_loss:.3e} ce {e_ce:.3e} "
          f"(kld {out['kld_loss'].item():.4f} mse {out['mse_loss'].item():.4f})")
    assert e_emb < tol and e_log < tol and e_loss < tol and e_ce < tol
    out["loss"].backward()
    named = dict(model.named_parameters())
    for k, g in gold.items():
        if k.startswith("grad::"):
            e = _err(named[k[6:]].grad, g)
            scale = max(g.abs().max().item(), 1e-3)
            print(f"   grad {k[6:]}: err {e:.3e} (max {scale:.3e})")
            assert e <= tol_grad * scale, (k, e, scale)
    assert all(p.grad is None for p in model.decoder_model.parameters())
    # direct cal_loss on reference-provided inputs_embeds (the way the fixture was produced)
    from speechmix_amd.model import shift_tokens_right
    lc = model.decoder_model.config
    dec_in = shift_tokens_right(inp["labels"], lc.pad_token_id, lc.decoder_start_token_id)
    o2 = model.cal_loss(inputs_embeds=gold["inputs_embeds"], text_input_ids=inp["text_input_ids"], decoder_input_ids=dec_in,
                        labels=inp["labels"])
    assert abs(o2["loss"].item() - gold["loss"].item()) < tol


def test_weighted_sum_share_ratio_matches_reference_and_oracle_grads():
    """weighted_sum=True (HF-twin semantics: L+1 softmax weights), share_layer_ratio 0.5, ds 4, no labels -> logits
    vs the reference fixture; then with labels: loss and gradients (incl. weights_sum) vs the CPU oracle."""
    from oracle import speechmix_oracle as O
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case("eed_w2v2_bart_ws")
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], share_layer_ratio=0.5, down_scale=4, weighted_sum=True,
                         compute_dtype="fp32").eval()
    assert model.load_state_dict(sd, strict=False).missing_keys == []
    with torch.no_grad():
        out = model(inp["input_values"], return_model_detail=True)
    assert _err(out["raw_logits"], gold["raw_logits"]) < 1e-4
    assert torch.equal(out["logits"].cpu(), gold["logits"])
    assert abs(out["weighted_sum"].sum().item() - 1.0) < 1e-5 and out["weighted_sum"].numel() == 3
    labels = torch.randint(4, 128, (2, 5))
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()
              if not k.endswith(("embed_tokens.weight", "lm_head.weight", "nlp_emb.weight"))}
    ref = O.speechmix_eed_forward(leaves, m["enc_cfg"], m["lm_cfg"], inp["input_values"], labels=labels, down_scale=4,
                                  weighted_sum=True, num_speech_layers=2)
    ref["loss"].backward()
    out = model(inp["input_values"], labels=labels)
    assert abs(out["loss"].item() - ref["loss"].item()) < 1e-4
    out["loss"].backward()
    named = dict(model.named_parameters())
    for name in ("weights_sum", "enc_to_dec_proj.weight", "encoder_model.encoder.layers.0.attention.q_proj.weight",
                 "encoder_model.feature_projection.projection.weight"):
        g = leaves[name].grad
        e = _err(named[name].grad, g)
        assert e <= 2e-3 * max(g.abs().max().item(), 1e-3), (name, e)


def test_text_prompt_prepended_like_reference():
    """input_text_prompt (ref:speechmix/model.py:168-171): prompt token embeddings prepended to the speech embeddings."""
    from oracle import speechmix_oracle as O
    model, inp, gold, m = _build("eed_w2v2_bart", "fp32")
    sd, _, _, _ = load_case("eed_w2v2_bart")
    prompt = torch.tensor([7, 9, 11])
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()
              if not k.endswith(("embed_tokens.weight", "lm_head.weight", "nlp_emb.weight"))}
    ref = O.speechmix_eed_forward(leaves, m["enc_cfg"], m["lm_cfg"], inp["input_values"], labels=inp["labels"], down_scale=2,
                                  prompt_ids=prompt[None])
    ref["loss"].backward()
    out = model(inp["input_values"], labels=inp["labels"], input_text_prompt=prompt, return_model_detail=True)
    assert _err(out["raw_logits"], ref["raw_logits"]) < 1e-4
    assert abs(out["loss"].item() - ref["loss"].item()) < 1e-5
    out["loss"].backward()
    g = leaves["decoder_model.model.shared.weight"].grad
    assert _err(model.decoder_model.model.shared.weight.grad, g) <= 2e-3 * g.abs().max().item()
    g = leaves["enc_to_dec_proj.weight"].grad
    assert _err(model.enc_to_dec_proj.weight.grad, g) <= 2e-3 * g.abs().max().item()


@pytest.mark.parametrize("case", ["eed_w2v2_bart", "eed_hubert_mbart"])
def test_cached_greedy_decode_equals_recompute_loop(case):
    """KV-cached greedy decoding must emit exactly the tokens of the reference-style loop that
    re-runs the full model on the growing prefix and takes the arg-max of the last position (ref:eval.ipynb cell 6,
    ref:train.py:18-34) - here with the CPU oracle as that loop (pinned to the reference by the golden fixtures)."""
    from oracle import speechmix_oracle as O
    model, inp, gold, m = _build(case, "fp32")
    wave = inp["input_values"]
    B = wave.shape[0]
    lc = model.decoder_model.config
    steps = 6
    first = model.generate(wave, max_length=steps)           # with the real eos id (random-init LMs tend to stop at once)
    lc.eos_token_id = -1                                     # ...so the main comparison runs every step
    got = model.generate(wave, max_length=steps)
    assert all(len(g) == steps for g in got)
    assert all(f == g[:len(f)] for f, g in zip(first, got))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    prefix = torch.full((B, 1), lc.decoder_start_token_id, dtype=torch.int64)
    alive = [True] * B
    want = [[] for _ in range(B)]
    margin = 1e9
    for _ in range(steps):
        ref = O.speechmix_eed_forward(sd, m["enc_cfg"], m["lm_cfg"], wave, decoder_input_ids=prefix, down_scale=m["down_scale"],
                                      num_speech_layers=model.num_speech_encoder_layers)
        last = ref["raw_logits"][:, -1]
        top2 = last.topk(2, dim=-1).values
        nxt = last.argmax(-1)
        for b in range(B):
            if alive[b]:
                margin = min(margin, (top2[b, 0] - top2[b, 1]).item())
                if nxt[b].item() == lc.eos_token_id:
                    alive[b] = False
                else:
                    want[b].append(nxt[b].item())
        prefix = torch.cat([prefix, nxt[:, None]], 1)
        if not any(alive):
            break
    print(f"[{case}] greedy tokens {got} (smallest top-2 logit margin on the path {margin:.3e})")
    assert margin > 1e-4, "fixture has a near-tie: the comparison would be ill-posed"
    assert got == want

    # the cache itself: feed a fixed random continuation through the cached path and compare EVERY step's logits with the
    # oracle's full recompute on the same prefix (a random-init LM's arg-max path is a constant token)
    forced = torch.randint(4, lc.vocab_size, (B, steps), generator=torch.Generator().manual_seed(11))
    eng = model.engine
    model.store.refresh_shadow()
    wv = model._prep_wave(wave)
    x, ssv = eng.speech_fwd(wv, B, wv.shape[1], False)
    e, S, _ = eng.bridge_fwd(x, B, ssv["T"])
    enc = eng.lm_encode(e, None, B, S)
    kept = []
    eng.greedy_decode(enc, B, S, steps, lc.decoder_start_token_id, -1, lc.pad_token_id, forced=forced.to(model.device),
                      keep_logits=kept)
    full = torch.cat([torch.full((B, 1), lc.decoder_start_token_id, dtype=torch.int64), forced[:, :-1]], 1)
    ref = O.speechmix_eed_forward(sd, m["enc_cfg"], m["lm_cfg"], wave, decoder_input_ids=full, down_scale=m["down_scale"],
                                  num_speech_layers=model.num_speech_encoder_layers)["raw_logits"]
    worst = max((kept[t].float().cpu() - ref[:, t]).abs().max().item() for t in range(steps))
    print(f"[{case}] cached-step logits vs full recompute: max err {worst:.3e}")
    assert worst < 1e-3

    # LM-only form used for label creation (ref:train.py:18-34)
    ids = torch.randint(4, lc.vocab_size, (2, 7), generator=torch.Generator().manual_seed(5))
    got2 = model.generate_from_text(ids, max_length=5)
    pre = torch.full((2, 1), lc.decoder_start_token_id, dtype=torch.int64)
    want2, alive2 = [[], []], [True, True]
    for _ in range(5):
        logits = model.decoder_model(input_ids=ids, decoder_input_ids=pre).logits.float().cpu()
        nxt = logits[:, -1].argmax(-1)
        for b in range(2):
            if alive2[b]:
                if nxt[b].item() == lc.eos_token_id:
                    alive2[b] = False
                else:
                    want2[b].append(nxt[b].item())
        pre = torch.cat([pre, nxt[:, None]], 1)
    assert got2 == want2
