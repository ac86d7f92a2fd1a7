"""Generate golden fixtures from the REFERENCE itself (run in the build container only).

    python tests/golden/make_golden.py

Imports ref:speechmix/hf_model.py by file path (SURVEY.md §8c Route 1) and ref:speechmix/model.py
through a container-only `s3prl.hub` stand-in module (Route 2: s3prl/fairseq are not installed; the
stand-in only adapts the HF Wav2Vec2Model to the attribute names model.py touches and is NOT part of
any reference build).  Tiny random-init checkpoints are created with `save_pretrained`, the reference
classes are run on seeded inputs in eval mode, and weights + inputs + outputs are stored as .npz.
Nothing of the reference's source travels: the fixtures are data only.
"""
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_ref_hf():
    spec = importlib.util.spec_from_file_location("ref_hf_model", f"{REF}/speechmix/hf_model.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def save_tokenizer(lm_dir, vocab_size, pad, bos, eos):
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from transformers import PreTrainedTokenizerFast
    vocab = {f"t{i}": i for i in range(vocab_size)}
    tok = Tokenizer(WordLevel(vocab, unk_token="t3"))
    tok.pre_tokenizer = Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tok, pad_token=f"t{pad}", bos_token=f"t{bos}", eos_token=f"t{eos}",
                            unk_token="t3").save_pretrained(lm_dir)


def tiny_speech(kind, d):
    from transformers import HubertConfig, HubertModel, Wav2Vec2Config, Wav2Vec2Model
    common = dict(hidden_size=64, num_hidden_layers=4, num_attention_heads=4, intermediate_size=128,
                  conv_dim=(32,) * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, vocab_size=32)
    if kind == "w2v2":
        cfg = Wav2Vec2Config(**common)
        path = os.path.join(d, "w2v2_tiny")
        Wav2Vec2Model(cfg).save_pretrained(path)
    elif kind == "hubert":
        cfg = HubertConfig(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True,
                           feat_proj_layer_norm=True, **common)
        path = os.path.join(d, "hubert_tiny")
        HubertModel(cfg).save_pretrained(path)
    return path, cfg


def tiny_lm(kind, d):
    from transformers import (BartConfig, BartForConditionalGeneration, MBartConfig, MBartForConditionalGeneration,
                              T5Config, T5ForConditionalGeneration)
    if kind == "bart":
        cfg = BartConfig(vocab_size=128, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                         decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128,
                         max_position_embeddings=256, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                         decoder_start_token_id=2)
        path = os.path.join(d, "bart_tiny")
        m = BartForConditionalGeneration(cfg)
        with torch.no_grad():
            m.final_logits_bias.normal_(0, 0.02)
        m.save_pretrained(path)
    elif kind == "mbart":
        cfg = MBartConfig(vocab_size=160, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128,
                          max_position_embeddings=256, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                          decoder_start_token_id=2, scale_embedding=True, activation_function="relu")
        path = os.path.join(d, "mbart_tiny")
        MBartForConditionalGeneration(cfg).save_pretrained(path)
    elif kind == "t5":
        cfg = T5Config(vocab_size=128, d_model=64, d_kv=16, d_ff=128, num_layers=2, num_decoder_layers=2, num_heads=4,
                       relative_attention_num_buckets=8, relative_attention_max_distance=16,
                       pad_token_id=0, eos_token_id=1, decoder_start_token_id=0)
        path = os.path.join(d, "t5_tiny")
        T5ForConditionalGeneration(cfg).save_pretrained(path)
    save_tokenizer(path, cfg.vocab_size, cfg.pad_token_id, 0, cfg.eos_token_id)
    return path, cfg


def cfg_dict(cfg):
    d = cfg.to_dict()
    keep = {}
    for k, v in d.items():
        if isinstance(v, (int, float, str, bool, list, tuple)) or v is None:
            keep[k] = v
    return keep


def to_np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def capture_eed(model, x, labels, grads_of, **fw):
    """Run the reference class, capturing stage outputs with forward hooks (no reference edits)."""
    cap = {}
    hooks = []

    def hook(name, pick=lambda o: o):
        def f(mod, inp, out):
            cap[name] = pick(out).detach().clone()
        return f
    em = model.encoder_model
    hooks.append(em.feature_extractor.register_forward_hook(hook("cnn_out")))
    hooks.append(em.feature_extractor.conv_layers[0].register_forward_hook(hook("conv0")))
    hooks.append(em.feature_projection.register_forward_hook(
        hook("feature_projection", lambda o: o[0] if isinstance(o, tuple) else o)))
    hooks.append(em.encoder.pos_conv_embed.register_forward_hook(hook("pos_conv")))
    hooks.append(model.length_adapters.register_forward_hook(hook("post_adapter", lambda o: o.transpose(1, 2))))
    hooks.append(model.enc_to_dec_proj.register_forward_hook(hook("inputs_embeds")))
    model.zero_grad()
    # raw logits are hidden by the reference (argmax) -> intercept the LM's output
    lm_out = {}
    hooks.append(model.decoder_model.register_forward_hook(
        lambda mod, inp, out: lm_out.update(logits=out.logits.detach().clone(),
                                            enc=out.encoder_last_hidden_state.detach().clone())))
    out = model(input_values=x, labels=labels, **fw)
    res = {"raw_logits": lm_out["logits"], "lm_encoder_last_hidden": lm_out["enc"],
           "logits": out["logits"].detach(), "loss": out["loss"].detach()}
    out["loss"].backward()
    named = dict(model.named_parameters())
    for g in grads_of:
        res["grad::" + g] = named[g].grad.detach().clone()
    for h in hooks:
        h.remove()
    res.update(cap)
    return res


def main():
    torch.manual_seed(1234)
    ref = load_ref_hf()
    tmp = tempfile.mkdtemp()
    manifest = {}

    # ---------------- case 1: wav2vec2 (group-norm CNN, post-LN) + BART, down_scale 2 -------------
    enc_dir, ecfg = tiny_speech("w2v2", tmp)
    lm_dir, lcfg = tiny_lm("bart", tmp)
    model = ref.HFSpeechMixEED(enc_dir, lm_dir, down_scale=2).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():   # default-inited biases/LN are 0/1: randomise for coverage
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
    x = torch.randn(2, 8000) * 0.1
    labels = torch.randint(4, 128, (2, 6)); labels[1, -1] = -100
    hs = model.encoder_model(x, output_hidden_states=True)
    grads = ["enc_to_dec_proj.weight", "length_adapters.0.weight", "length_adapters.0.bias",
             "encoder_model.encoder.layers.1.attention.q_proj.weight",
             "encoder_model.encoder.layers.0.feed_forward.intermediate_dense.weight",
             "encoder_model.feature_extractor.conv_layers.0.conv.weight",
             "encoder_model.feature_extractor.conv_layers.0.layer_norm.weight",
             "encoder_model.feature_extractor.conv_layers.3.conv.weight",
             "encoder_model.encoder.pos_conv_embed.conv.parametrizations.weight.original0",
             "encoder_model.encoder.pos_conv_embed.conv.parametrizations.weight.original1",
             "encoder_model.feature_projection.layer_norm.weight",
             "decoder_model.model.shared.weight",
             "decoder_model.model.decoder.layers.1.encoder_attn.k_proj.weight",
             "decoder_model.model.encoder.layers.0.self_attn_layer_norm.bias",
             "decoder_model.model.decoder.embed_positions.weight"]
    r = capture_eed(model, x, labels, grads)
    r["encoder_last_hidden_state"] = hs.last_hidden_state.detach()
    for i, h in enumerate(hs.hidden_states):
        r[f"enc_hidden_{i}"] = h.detach()
    np.savez_compressed(f"{OUT}/eed_w2v2_bart.npz", input_values=x.numpy(), labels=labels.numpy(),
                        **{"w::" + k: v for k, v in to_np(model.state_dict()).items()},
                        **{"o::" + k: v.numpy() for k, v in r.items()})
    manifest["eed_w2v2_bart"] = {"enc_cfg": cfg_dict(ecfg), "lm_cfg": cfg_dict(lcfg), "down_scale": 2,
                                 "share_layer_ratio": 0, "route": "hf_model.HFSpeechMixEED"}
    print("case1 loss", float(r["loss"]))

    # no-label path (handle_decoder_input_none) + weighted_sum (HF semantics, L+1 weights) + share 0.5
    model_ws = ref.HFSpeechMixEED(enc_dir, lm_dir, down_scale=4, weighted_sum=True, share_layer_ratio=0.5).eval()
    with torch.no_grad():
        model_ws.weights_sum.copy_(torch.randn_like(model_ws.weights_sum))
        model_ws.enc_to_dec_proj.bias.normal_(0, 0.05)
    cap = {}
    h = model_ws.decoder_model.register_forward_hook(lambda m, i, o: cap.update(logits=o.logits.detach().clone()))
    o = model_ws(input_values=x)
    h.remove()
    # (the HF twin drops its `return_dict` details on return: only logits come back)
    np.savez_compressed(f"{OUT}/eed_w2v2_bart_ws.npz", input_values=x.numpy(),
                        **{"w::" + k: v for k, v in to_np(model_ws.state_dict()).items()},
                        **{"o::raw_logits": cap["logits"].numpy(), "o::logits": o["logits"].numpy()})
    manifest["eed_w2v2_bart_ws"] = {"enc_cfg": cfg_dict(ecfg), "lm_cfg": cfg_dict(lcfg), "down_scale": 4,
                                    "share_layer_ratio": 0.5, "weighted_sum": True,
                                    "speech_encoder_layer": model_ws.speech_encoder_layer,
                                    "route": "hf_model.HFSpeechMixEED"}

    # structural invariants of the ctor (ref:test/test_model.py:18-53)
    struct = {}
    for ratio in (0, 0.4, 0.5, 1):
        mm = ref.HFSpeechMixEED(enc_dir, lm_dir, share_layer_ratio=ratio)
        struct[f"layers@{ratio}"] = mm.speech_encoder_layer
        struct[f"nlp_layers@{ratio}"] = mm.nlp_encoder_layer
        struct[f"n_no_grad@{ratio}"] = len(mm.list_no_grad)
    for ds in (1, 2, 4, 8):
        mm = ref.HFSpeechMixEED(enc_dir, lm_dir, down_scale=ds).eval()
        shp = {}
        hk = mm.length_adapters.register_forward_hook(
            lambda m, i, o: shp.update(before=int(i[0].shape[-1]), after=int(o.shape[-1])))
        mm(input_values=x)
        hk.remove()
        struct[f"frames@{ds}"] = [shp["before"], shp["after"]]
    mm = ref.HFSpeechMixEED(enc_dir, lm_dir, fixed_parameters=True)
    struct["fixed_parameters_list_grad"] = sorted(mm.list_grad)
    manifest["structure"] = struct

    # ---------------- case 2: HuBERT-style (layer-norm CNN, conv bias, stable LN) + mBART, ds 8 ----
    enc_dir2, ecfg2 = tiny_speech("hubert", tmp)
    lm_dir2, lcfg2 = tiny_lm("mbart", tmp)
    model2 = ref.HFSpeechMixEED(enc_dir2, lm_dir2, down_scale=8).eval()
    with torch.no_grad():
        for n, p in model2.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
    x2 = torch.randn(2, 16000) * 0.1
    labels2 = torch.randint(4, 160, (2, 5)); labels2[0, -2:] = -100
    grads2 = ["enc_to_dec_proj.weight", "length_adapters.2.weight",
              "encoder_model.encoder.layers.0.attention.v_proj.weight",
              "encoder_model.feature_extractor.conv_layers.0.conv.weight",
              "encoder_model.feature_extractor.conv_layers.0.conv.bias",
              "encoder_model.feature_extractor.conv_layers.2.layer_norm.weight",
              "encoder_model.encoder.layer_norm.weight",
              "decoder_model.model.shared.weight", "decoder_model.model.decoder.layer_norm.weight"]
    hs2 = model2.encoder_model(x2)
    r2 = capture_eed(model2, x2, labels2, grads2)
    r2["encoder_last_hidden_state"] = hs2.last_hidden_state.detach()
    np.savez_compressed(f"{OUT}/eed_hubert_mbart.npz", input_values=x2.numpy(), labels=labels2.numpy(),
                        **{"w::" + k: v for k, v in to_np(model2.state_dict()).items()},
                        **{"o::" + k: v.numpy() for k, v in r2.items()})
    manifest["eed_hubert_mbart"] = {"enc_cfg": cfg_dict(ecfg2), "lm_cfg": cfg_dict(lcfg2), "down_scale": 8,
                                    "share_layer_ratio": 0, "route": "hf_model.HFSpeechMixEED"}
    print("case2 loss", float(r2["loss"]))

    # ---------------- case 3: SpeechMixSelf losses, wav2vec2 + T5, share 0.5, ds 4 ------------------
    lm_dir3, lcfg3 = tiny_lm("t5", tmp)
    model3 = ref.HFSpeechMixSelf(enc_dir, lm_dir3, down_scale=4, share_layer_ratio=0.5).eval()
    with torch.no_grad():
        for n, p in model3.named_parameters():
            if p.ndim == 1 and "decoder_model" not in n:
                p.add_(torch.randn_like(p) * 0.05)
    x3 = torch.randn(2, 8000) * 0.1
    labels3 = torch.randint(2, 128, (2, 6)); labels3[1, -1] = -100
    text3 = torch.randint(2, 128, (2, 7))
    # forward() of the Self class is broken at this commit: drive cal_loss directly,
    # with inputs_embeds produced by the reference's own pre-LM pipeline.
    feats = model3.encoder_model(x3).last_hidden_state
    emb = model3.enc_to_dec_proj(model3.length_adapters(feats.transpose(1, 2)).transpose(1, 2))
    dec_in = ref.shift_tokens_right(labels3, model3.decoder_model.config.pad_token_id,
                                    model3.decoder_model.config.decoder_start_token_id)
    model3.zero_grad()
    o3 = model3.cal_loss(inputs_embeds=emb, text_input_ids=text3, decoder_input_ids=dec_in, labels=labels3)
    o3["loss"].backward()
    # separate terms recomputed from the reference's own outputs (the class only returns the sum)
    with torch.no_grad():
        plain = model3.decoder_model(inputs_embeds=emb, decoder_input_ids=dec_in, labels=labels3)
    named3 = dict(model3.named_parameters())
    np.savez_compressed(
        f"{OUT}/self_w2v2_t5.npz", input_values=x3.numpy(), labels=labels3.numpy(), text_input_ids=text3.numpy(),
        **{"w::" + k: v for k, v in to_np(model3.state_dict()).items()},
        **{"o::loss": o3["loss"].detach().numpy(), "o::ce": plain.loss.numpy(),
           "o::raw_logits": o3["logits"].detach().numpy(), "o::inputs_embeds": emb.detach().numpy(),
           "o::grad::enc_to_dec_proj.weight": named3["enc_to_dec_proj.weight"].grad.numpy(),
           "o::grad::length_adapters.1.weight": named3["length_adapters.1.weight"].grad.numpy(),
           "o::grad::encoder_model.encoder.layers.1.attention.q_proj.weight":
               named3["encoder_model.encoder.layers.1.attention.q_proj.weight"].grad.numpy()})
    manifest["self_w2v2_t5"] = {"enc_cfg": cfg_dict(ecfg), "lm_cfg": cfg_dict(lcfg3), "down_scale": 4,
                                "share_layer_ratio": 0.5, "speech_encoder_layer": model3.speech_encoder_layer,
                                "n_frozen": len(model3.list_no_grad), "route": "hf_model.HFSpeechMixSelf.cal_loss"}
    print("case3 loss", float(o3["loss"]))

    # ---------------- Route 2: ref:speechmix/model.py SpeechMixEED via an s3prl.hub stand-in -------
    from transformers import Wav2Vec2Model

    class _Upstream(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Wav2Vec2Model.from_pretrained(enc_dir)
            self.model.final_proj = torch.nn.Linear(self.model.config.hidden_size, 8)  # only .in_features is read

        def forward(self, wavs):
            o = self.model(wavs if torch.is_tensor(wavs) else torch.stack(list(wavs)), output_hidden_states=True)
            return {"last_hidden_state": o.last_hidden_state, "hidden_states": o.hidden_states}
    hub = types.ModuleType("s3prl.hub"); hub.wav2vec2 = _Upstream
    pkg = types.ModuleType("s3prl"); pkg.hub = hub
    sys.modules["s3prl"] = pkg; sys.modules["s3prl.hub"] = hub
    spec = importlib.util.spec_from_file_location("ref_model", f"{REF}/speechmix/model.py")
    refm = importlib.util.module_from_spec(spec); spec.loader.exec_module(refm)
    m4 = refm.SpeechMixEED("wav2vec2", lm_dir, down_scale=2).eval()
    cap = {}
    h = m4.decoder_model.register_forward_hook(lambda m, i, o: cap.update(logits=o.logits.detach().clone()))
    o4 = m4(x, labels=labels)
    h.remove()
    sd4 = {}
    for k, v in m4.state_dict().items():
        k = k.replace("encoder_model.model.", "encoder_model.")
        if "final_proj" in k:
            continue
        sd4[k] = v
    np.savez_compressed(f"{OUT}/eed_route2_model_py.npz", input_values=x.numpy(), labels=labels.numpy(),
                        **{"w::" + k: v for k, v in to_np(sd4).items()},
                        **{"o::raw_logits": cap["logits"].numpy(), "o::logits": o4["logits"].numpy(),
                           "o::loss": o4["loss"].detach().numpy()})
    manifest["eed_route2_model_py"] = {"enc_cfg": cfg_dict(ecfg), "lm_cfg": cfg_dict(lcfg), "down_scale": 2,
                                       "share_layer_ratio": 0, "n_no_grad": len(m4.list_no_grad),
                                       "route": "model.SpeechMixEED (s3prl.hub stand-in)"}
    print("route2 loss", float(o4["loss"]))

    # ---------------- integer cases: shift_tokens_right / handle_decoder_input_none ----------------
    cases = [torch.tensor([[5, 6, 7, 2]]), torch.tensor([[5, -100, -100, -100], [9, 8, 7, -100]]),
             torch.tensor([[-100]]), torch.randint(0, 50000, (4, 33))]
    ints = {}
    for i, c in enumerate(cases):
        ints[f"in{i}"] = c.numpy()
        ints[f"out{i}"] = ref.shift_tokens_right(c, 1, 2).numpy()
    np.savez_compressed(f"{OUT}/shift_tokens_right.npz", **ints)

    with open(f"{OUT}/manifest.json", "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
