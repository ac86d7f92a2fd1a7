"""CPU tests of the host-side mirror of the reference interface: constructor semantics, parameter naming
(state-dict compatibility with the reference's HF twin), flat storage layout, reduction buckets.
Structural invariants are the ones the reference's own tests assert (ref:test/test_model.py:10-53),
checked against values recorded from the reference in tests/golden/manifest.json."""
import math

import pytest
import torch

from tests.golden_util import load_case, manifest


def _model(case="eed_w2v2_bart", cls=None, **kw):
    from speechmix_amd import model as M
    sd, inp, gold, m = load_case(case)
    cls = cls or M.SpeechMixEED
    args = dict(down_scale=m["down_scale"], share_layer_ratio=m.get("share_layer_ratio", 0), compute_dtype="fp32")
    args.update(kw)
    return cls(m["enc_cfg"], m["lm_cfg"], **args), sd, inp, gold, m


@pytest.mark.parametrize("case", ["eed_w2v2_bart", "eed_hubert_mbart", "self_w2v2_t5"])
def test_state_dict_names_match_reference(case):
    model, sd, *_ = _model(case)
    own = model.state_dict()
    missing = [k for k in own if k not in sd and k != "weights_sum"]
    unexpected = [k for k in sd if k not in own]
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    for k, v in sd.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
    res = model.load_state_dict(sd, strict=False)
    assert res.unexpected_keys == [] and res.missing_keys == ["weights_sum"]
    assert torch.equal(model.enc_to_dec_proj.weight.detach().cpu(), sd["enc_to_dec_proj.weight"])


def test_layer_sharing_and_grad_lists_like_reference():
    """ref:test/test_model.py:10-25."""
    s = manifest()["structure"]
    from speechmix_amd.model import SpeechMixEED
    _, _, _, m = load_case("eed_w2v2_bart")
    for ratio in (0, 0.4, 0.5, 1):
        mod = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], share_layer_ratio=ratio, compute_dtype="fp32")
        assert mod.speech_encoder_layer == s[f"layers@{ratio}"]
        assert mod.nlp_encoder_layer == s[f"nlp_layers@{ratio}"]
        assert len(mod.list_no_grad) == s[f"n_no_grad@{ratio}"] == 0
    # the recorded list comes from the HF twin, whose default `fixed_except` has no 'encoder' entry
    # (ref:speechmix/hf_model.py:196-203 vs ref:speechmix/model.py:60-61)
    mod = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], fixed_parameters=True, compute_dtype="fp32",
                       fixed_except=["layer_norm", "encoder_attn", "enc_to_dec_proj", "length_adapter",
                                     "layernorm_embedding", "attention"])
    ours = sorted(n for n in mod.list_grad if n not in ("weights_sum",))
    ref = sorted(n for n in s["fixed_parameters_list_grad"] if not n.startswith("nlp_emb"))
    assert [n for n in ours if not n.startswith("nlp_emb")] == ref


def test_frame_counts_match_down_scale():
    """ref:test/test_model.py:37-53: round(T_before / T_after) == down_scale."""
    from speechmix_amd.configs import load_speech_config
    s = manifest()["structure"]
    _, _, _, m = load_case("eed_w2v2_bart")
    cfg, _ = load_speech_config(m["enc_cfg"])
    T = cfg.frames(8000)
    for ds in (1, 2, 4, 8):
        t = T
        for _ in range(int(math.log(ds, 2))):
            t = (t - 2) // 2 + 1
        assert [T, t] == s[f"frames@{ds}"]


def test_ctor_swallows_cli_kwargs_and_subclasses():
    """train.py splats every CLI option into the ctor (ref:train.py:190-225)."""
    from speechmix_amd import model as M
    _, _, _, m = load_case("eed_w2v2_bart")
    mod = M.SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, SpeechMixEED=True, batch=3, grad_accum=20, lr=4e-5,
                         dataset="librispeech_asr", worker=10, fp16=True, compute_dtype="fp32")
    assert mod.downsize == 2 and mod.downloop == 1 and len(mod.length_adapters) == 1
    fixed = M.SpeechMixFixed(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32")
    assert all(not p.requires_grad for p in fixed.decoder_model.parameters())
    assert all(p.requires_grad for p in fixed.encoder_model.parameters())
    assert fixed.decoder_model.config.decoder_start_token_id == 2 and fixed.decoder_model.config.hidden_size == 64


def test_shift_tokens_right_and_decoder_input_none():
    import numpy as np
    from speechmix_amd.model import handle_decoder_input_none, shift_tokens_right
    from tests.golden_util import GOLDEN
    z = np.load(f"{GOLDEN}/shift_tokens_right.npz")
    for i in range(4):
        assert torch.equal(shift_tokens_right(torch.from_numpy(z[f"in{i}"]), 1, 2), torch.from_numpy(z[f"out{i}"]))
    with pytest.raises(AssertionError):
        shift_tokens_right(torch.tensor([[3, 4]]), None, 2)

    class Cfg:
        decoder_start_token_id = 2
    assert handle_decoder_input_none(Cfg, 3).tolist() == [[2], [2], [2]]


def test_flat_store_layout():
    from speechmix_amd.params import FlatStore
    model, sd, *_ = _model()
    model.load_state_dict(sd, strict=False)
    st = FlatStore(model, "cpu", torch.float32)
    # q|k|v adjacent -> one fused operand view
    p = "encoder_model.encoder.layers.0.attention."
    w = st.cat([p + "q_proj.weight", p + "k_proj.weight", p + "v_proj.weight"])
    assert w.shape == (192, 64)
    assert torch.equal(w[64:128], sd[p + "k_proj.weight"])
    b = st.cat([p + "q_proj.bias", p + "k_proj.bias", p + "v_proj.bias"], "p32")
    assert torch.equal(b[128:], sd[p + "v_proj.bias"])
    # parameters are views of the flat buffer; tied weights stored once
    assert model.enc_to_dec_proj.weight.data_ptr() == st.p32("enc_to_dec_proj.weight").data_ptr()
    assert "decoder_model.lm_head.weight" not in st.offsets and "decoder_model.model.shared.weight" in st.offsets
    for name, (off, n, shape) in st.offsets.items():
        assert off % 8 == 0, name             # 16-B aligned bf16 rows
    # in-place optimizer updates on the Parameter are visible through the store
    with torch.no_grad():
        model.enc_to_dec_proj.bias.add_(1.0)
    assert torch.allclose(st.p32("enc_to_dec_proj.bias"), sd["enc_to_dec_proj.bias"] + 1.0)


def test_reduction_buckets_cover_every_trainable_gradient_once():
    from speechmix_amd.dist import stage_ranges
    from speechmix_amd.params import FlatStore
    from speechmix_amd.trainer import trainable_ranges
    model, *_ = _model()
    st = FlatStore(model, "cpu", torch.float32)
    stages = stage_ranges(st.offsets, model.num_speech_encoder_layers)
    names = [s for s, _ in stages]
    assert names[0] == "lm" and names[1] == "bridge" and names[-1] == "frontend"
    assert names[2:-1] == [f"enc_layer{i}" for i in range(model.num_speech_encoder_layers - 1, -1, -1)]
    covered = torch.zeros(st.total, dtype=torch.int32)
    for _, rs in stages:
        for a, b in rs:
            covered[a:b] += 1
    assert covered.max() == 1
    for name, (off, n, _) in st.offsets.items():
        assert covered[off:off + n].min() == 1, name
    tr = trainable_ranges(st)
    assert tr[0][0] == 0 and tr[-1][1] <= st.total
    for p in model.decoder_model.parameters():
        p.requires_grad = False
    tr2 = trainable_ranges(st)
    lo = min(o for n, (o, _, _) in st.offsets.items() if n.startswith("decoder_model."))
    hi = max(o + k for n, (o, k, _) in st.offsets.items() if n.startswith("decoder_model."))
    assert all(b <= lo or a >= hi for a, b in tr2)


def test_forward_needs_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model, sd, inp, *_ = _model()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(inp["input_values"], labels=inp["labels"])


def test_config_presets_and_errors():
    from speechmix_amd.configs import load_lm_config, load_speech_config
    c, _ = load_speech_config("wav2vec2")
    assert (c.hidden_size, c.num_hidden_layers, c.frames(160000)) == (768, 12, 499)
    c, _ = load_speech_config("hubert_large_ll60k")
    assert c.do_stable_layer_norm and c.feat_extract_norm == "layer" and c.conv_bias and c.hidden_size == 1024
    l, _ = load_lm_config("facebook/bart-base")
    assert (l.d_model, l.vocab_size, l.decoder_start_token_id) == (768, 50265, 2)
    l, _ = load_lm_config("t5-large")
    assert l.model_type == "t5" and l.d_kv == 64 and l.pad_token_id == 0
    with pytest.raises(ValueError):
        load_speech_config("no-such-model")
    with pytest.raises(ValueError):
        load_lm_config({"model_type": "gpt2"})


def test_collator_restates_reference_padding_rules():
    """ref:train.py:90-133: waveform padded with -100, labels padded then masked to -100, shared leading bos cut;
    cross-checked against an HF tokenizer's own pad() when transformers is importable."""
    import types
    from speechmix_amd.data import DataCollatorWithPadding, DevicePrefetcher
    tok = types.SimpleNamespace(pad_token_id=1, bos_token_id=5)
    feats = [dict(input_values=[0.1, 0.2, 0.3], labels=[5, 9, 8, 2], text_input_ids=[5, 9, 8]),
             dict(input_values=[0.5], labels=[5, 7, 2], text_input_ids=[5, 7])]
    b = DataCollatorWithPadding(tok)(feats)
    assert b["input_values"].tolist() == [[pytest.approx(0.1), pytest.approx(0.2), pytest.approx(0.3)], [0.5, -100.0, -100.0]]
    assert b["labels"].tolist() == [[9, 8, 2], [7, 2, -100]]                       # bos column cut, pad -> -100
    assert b["text_input_ids"].tolist() == [[5, 9, 8], [5, 7, 1]]                  # padded with pad id, not masked
    feats[1]["labels"] = [6, 7, 2]                                                 # bos not shared by every row: kept
    assert DataCollatorWithPadding(tok)(feats)["labels"].tolist() == [[5, 9, 8, 2], [6, 7, 2, -100]]
    tok0 = types.SimpleNamespace(pad_token_id=1, bos_token_id=0)                   # bos id 0 is falsy: never cut (as the reference)
    f0 = [dict(input_values=[0.0], labels=[0, 4]), dict(input_values=[0.0], labels=[0, 3])]
    assert DataCollatorWithPadding(tok0)(f0)["labels"].tolist() == [[0, 4], [0, 3]]
    # CPU prefetcher is a pass-through iterator
    out = list(DevicePrefetcher([b, b], "cpu"))
    assert len(out) == 2 and out[0]["labels"] is b["labels"]


def test_local_hf_checkpoint_directories_load_by_key_name(tmp_path):
    """checkpoints written by HuggingFace's own `save_pretrained` (the format of every pretrained
    backbone the reference names) must load by parameter name: every tensor of the HF speech encoder and of the HF LM is
    found in the drop-in model with identical values (incl. the tied embedding and the weight-norm parametrisation keys)."""
    transformers = pytest.importorskip("transformers")
    from transformers import BartConfig, BartForConditionalGeneration, Wav2Vec2Config, Wav2Vec2Model
    from speechmix_amd.model import SpeechMixEED
    torch.manual_seed(0)
    ecfg = Wav2Vec2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                          conv_dim=[32] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, vocab_size=32)
    lcfg = BartConfig(vocab_size=96, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                      decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, max_position_embeddings=64,
                      pad_token_id=1, bos_token_id=0, eos_token_id=2, decoder_start_token_id=2)
    enc, lm = Wav2Vec2Model(ecfg), BartForConditionalGeneration(lcfg)
    enc_dir, lm_dir = str(tmp_path / "wav2vec2-tiny"), str(tmp_path / "bart-tiny")
    enc.save_pretrained(enc_dir, safe_serialization=True)
    lm.save_pretrained(lm_dir, safe_serialization=True)
    with warnings_ignored():
        model = SpeechMixEED(enc_dir, lm_dir, down_scale=2)
    own = model.state_dict()
    checked = 0
    for k, v in enc.state_dict().items():
        if k.startswith("masked_spec_embed") or "weight" in k or "bias" in k:
            name = "encoder_model." + k
            assert name in own, name
            assert torch.equal(own[name].float().cpu(), v.float()), name
            checked += 1
    for k, v in lm.state_dict().items():
        if k in ("final_logits_bias",):
            continue
        name = "decoder_model." + k
        assert name in own, name
        assert torch.equal(own[name].float().cpu(), v.float()), name
        checked += 1
    assert checked > 100


def warnings_ignored():
    import contextlib
    import warnings

    @contextlib.contextmanager
    def cm():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            yield
    return cm()
