"""Pin the CPU oracle against outputs of the reference itself (fixtures made by
tests/golden/make_golden.py from ref:speechmix/hf_model.py and ref:speechmix/model.py).
Tolerances (fp32 vs fp32, different summation order): 2e-5 abs on activations/logits, 1e-5 on loss,
grads 2e-5 abs + 1e-4 rel."""
import math

import numpy as np
import pytest
import torch

from oracle import speechmix_oracle as O
from tests.golden_util import load_case, manifest, GOLDEN

ATOL = 2e-5


def _close(a, b, atol=ATOL, rtol=1e-4, what=""):
    a = a.detach().float()
    b = b.detach().float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, atol=atol, rtol=rtol), f"{what}: max abs err {err}"


def _run_with_grads(sd, m, inp, **kw):
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    # tied embeddings: the reference shares one tensor; grads must accumulate on it
    for k in list(sd):
        if k.endswith(("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight")):
            del sd[k]
    trace = {}
    L = None
    if m.get("share_layer_ratio"):
        n = m["enc_cfg"]["num_hidden_layers"]
        L = n - int(n * m["share_layer_ratio"])
    out = O.speechmix_eed_forward(sd, m["enc_cfg"], m["lm_cfg"], inp["input_values"], labels=inp.get("labels"),
                                  down_scale=m["down_scale"], num_speech_layers=L, trace=trace, **kw)
    return sd, out, trace


@pytest.mark.parametrize("case", ["eed_w2v2_bart", "eed_hubert_mbart"])
def test_eed_forward_and_grads(case):
    sd, inp, gold, m = load_case(case)
    sd, out, trace = _run_with_grads(sd, m, inp)
    if "conv0" in gold:
        _close(trace["conv0"], gold["conv0"], what="conv0")
    _close(trace["conv6"], gold["cnn_out"], what="cnn_out")
    _close(trace["feature_projection"], gold["feature_projection"], what="feature_projection")
    _close(out["encoder_last_hidden_state"], gold["encoder_last_hidden_state"], what="encoder_last_hidden_state")
    _close(out["post_adapter"], gold["post_adapter"], what="post_adapter")
    _close(out["inputs_embeds"], gold["inputs_embeds"], what="inputs_embeds")
    _close(out["lm_encoder_last_hidden"], gold["lm_encoder_last_hidden"], what="lm_encoder_last_hidden")
    _close(out["raw_logits"], gold["raw_logits"], what="raw_logits")
    assert torch.equal(out["logits"], gold["logits"])
    assert abs(out["loss"].item() - gold["loss"].item()) < 1e-5
    out["loss"].backward()
    n = 0
    for k, g in gold.items():
        if k.startswith("grad::"):
            name = k[6:]
            _close(sd[name].grad, g, what=k)
            n += 1
    assert n >= 8


def test_eed_hidden_states_all_layers():
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    enc_sd, _, _ = O.split_state_dict(sd)
    _, hidden = O.speech_encoder(enc_sd, m["enc_cfg"], inp["input_values"])
    assert len(hidden) == m["enc_cfg"]["num_hidden_layers"] + 1
    for i, h in enumerate(hidden):
        _close(h, gold[f"enc_hidden_{i}"], what=f"hidden {i}")


def test_weighted_sum_share_ratio_and_no_labels():
    sd, inp, gold, m = load_case("eed_w2v2_bart_ws")
    assert m["speech_encoder_layer"] == 2
    out = O.speechmix_eed_forward(sd, m["enc_cfg"], m["lm_cfg"], inp["input_values"], down_scale=4,
                                  weighted_sum=True, num_speech_layers=2)
    assert out["raw_logits"].shape[1] == 1          # handle_decoder_input_none -> one start token
    _close(out["raw_logits"], gold["raw_logits"], what="ws raw_logits")
    assert torch.equal(out["logits"], gold["logits"])


def test_route2_model_py():
    """ref:speechmix/model.py SpeechMixEED (north-star file) gives the same numbers as the oracle."""
    sd, inp, gold, m = load_case("eed_route2_model_py")
    assert m["n_no_grad"] == 0                        # ref:test/test_model.py:16
    out = O.speechmix_eed_forward(sd, m["enc_cfg"], m["lm_cfg"], inp["input_values"], labels=inp["labels"],
                                  down_scale=2)
    _close(out["raw_logits"], gold["raw_logits"], what="route2 raw_logits")
    assert torch.equal(out["logits"], gold["logits"])
    assert abs(out["loss"].item() - gold["loss"].item()) < 1e-5


def test_self_losses_t5():
    sd, inp, gold, m = load_case("self_w2v2_t5")
    assert m["speech_encoder_layer"] == 2
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    enc_sd, _, rest = O.split_state_dict(sd)
    last, _ = O.speech_encoder(enc_sd, m["enc_cfg"], inp["input_values"], num_layers=2)
    x = O.length_adapters(rest, last, 2)
    emb = x @ rest["enc_to_dec_proj.weight"].t() + rest["enc_to_dec_proj.bias"]
    _close(emb, gold["inputs_embeds"], what="self inputs_embeds")
    dec_in = O.shift_tokens_right(inp["labels"], m["lm_cfg"]["pad_token_id"], m["lm_cfg"]["decoder_start_token_id"])
    r = O.speechmix_self_losses(sd, m["lm_cfg"], emb, inp["text_input_ids"], dec_in, inp["labels"])
    _close(r["raw_logits"], gold["raw_logits"], what="self logits")
    assert abs(r["ce"].item() - gold["ce"].item()) < 1e-5
    assert abs(r["loss"].item() - gold["loss"].item()) < 2e-5
    r["loss"].backward()
    for k, g in gold.items():
        if k.startswith("grad::"):
            _close(sd[k[6:]].grad, g, what=k)


def test_shift_tokens_right_bit_exact():
    z = np.load(f"{GOLDEN}/shift_tokens_right.npz")
    i = 0
    while f"in{i}" in z.files:
        got = O.shift_tokens_right(torch.from_numpy(z[f"in{i}"]), 1, 2)
        assert torch.equal(got, torch.from_numpy(z[f"out{i}"]))
        i += 1
    assert i == 4
    with pytest.raises(AssertionError):
        O.shift_tokens_right(torch.tensor([[1, 2]]), None, 2)


def test_structure_invariants_recorded():
    s = manifest()["structure"]
    # ref:test/test_model.py:18-25 (layer sharing) and :37-53 (frame ratio == down_scale)
    assert [s[f"layers@{r}"] for r in ("0", "0.4", "0.5", "1")] == [4, 3, 2, 0]
    assert s["n_no_grad@0"] == 0
    for ds in (1, 2, 4, 8):
        before, after = s[f"frames@{ds}"]
        assert round(before / after) == ds
        assert O.conv_out_len(8000, [10, 3, 3, 3, 3, 2, 2], [5, 2, 2, 2, 2, 2, 2]) == before


def test_adafactor_restatement_matches_hf():
    """oracle.adafactor_step vs transformers.optimization.Adafactor as HF Trainer configures it for optim="adafactor"
    (ref:train.py:298): three steps on five tensor shapes (tests/golden/make_adafactor_golden.py)."""
    z = np.load(f"{GOLDEN}/adafactor.npz")
    names = sorted({k.split("::")[1] for k in z.files})
    assert len(names) == 5
    for nm in names:
        p = torch.from_numpy(z[f"p0::{nm}"]).clone()
        st = {}
        for step in range(3):
            O.adafactor_step(p, torch.from_numpy(z[f"g{step}::{nm}"]), st, lr=5e-4)
            _close(p, torch.from_numpy(z[f"p{step + 1}::{nm}"]), atol=1e-6, rtol=1e-6, what=f"adafactor {nm} step {step}")
