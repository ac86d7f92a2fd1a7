"""SURVEY.md §8 f2, the length-aware path, CPU part: the oracle's padding-mask arithmetic pinned to the reference
(fixtures of tests/golden/make_golden_r3.py):
  * the LM hook `decoder_model(inputs_embeds=, attention_mask=, ...)` as ref:speechmix/model.py:132-136 calls it;
  * the speech encoder of HFSpeechMixEED under a ragged sample-level `attention_mask`
    (TF:models/wav2vec2/modeling_wav2vec2.py:1041-1060, 1349-1358, 688-697), wav2vec2 (post-LN) and HuBERT-style (stable LN).
2e-5 abs on activations / logits, 1e-5 on the loss, gradients 2e-5 abs + 1e-4 rel (as test_oracle_golden.py).
The GPU part is tests/test_gpu_r3.py."""
import pytest
import torch

from oracle import speechmix_oracle as O
from tests.golden_util import load_case
from tests.test_oracle_golden import _close


def test_oracle_lm_with_attention_mask_matches_reference_hook():
    sd, inp, gold, m = load_case("lm_attention_mask")
    enc_sd, lm_sd, rest = O.split_state_dict(sd)
    lm_sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in lm_sd.items()}
    for k in list(lm_sd):
        if k.endswith(("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight")):
            del lm_sd[k]
    emb = inp["inputs_embeds"].clone().requires_grad_(True)
    cfg = m["lm_cfg"]
    dec = O.shift_tokens_right(inp["labels"], cfg["pad_token_id"], cfg["decoder_start_token_id"])
    logits, enc = O.lm_forward(lm_sd, cfg, inputs_embeds=emb, decoder_input_ids=dec, attention_mask=inp["attention_mask"])
    _close(logits, gold["raw_logits"], what="raw_logits")
    # the reference's encoder output at PADDED positions is whatever the layers make of them: compared everywhere
    _close(enc, gold["lm_encoder_last_hidden"], what="lm_encoder_last_hidden")
    loss = O.cross_entropy(logits, inp["labels"])
    assert abs(loss.item() - gold["loss"].item()) < 1e-5
    loss.backward()
    _close(emb.grad, gold["grad::inputs_embeds"], what="grad::inputs_embeds")
    n = 0
    for k, g in gold.items():
        if k.startswith("grad::decoder_model."):
            _close(lm_sd[k[len("grad::decoder_model."):]].grad, g, what=k)
            n += 1
    assert n == 5
    # and the mask matters: without it the logits move
    l2, _ = O.lm_forward({k: v.detach() for k, v in lm_sd.items()}, cfg, inputs_embeds=emb.detach(), decoder_input_ids=dec)
    assert (l2 - gold["raw_logits"]).abs().max() > 1e-3


@pytest.mark.parametrize("case", ["w2v2_attention_mask", "hubert_attention_mask"])
def test_oracle_speech_encoder_with_attention_mask_matches_reference(case):
    sd, inp, gold, m = load_case(case)
    enc_sd, _, _ = O.split_state_dict(sd)
    fl = O.feature_lengths(m["enc_cfg"], inp["sample_lengths"].tolist())
    assert fl.tolist() == [24, 16, 7]
    with torch.no_grad():
        last, hidden = O.speech_encoder(enc_sd, m["enc_cfg"], inp["input_values"], frame_lengths=fl)
    _close(last, gold["encoder_last_hidden_state"], what="encoder_last_hidden_state")
    for i, h in enumerate(hidden):
        _close(h, gold["hidden_states"][i], what=f"hidden_states[{i}]")
    with torch.no_grad():
        plain, _ = O.speech_encoder(enc_sd, m["enc_cfg"], inp["input_values"])
    assert (plain - gold["encoder_last_hidden_state"]).abs().max() > 1e-3
