"""Round-4 fixtures: the REFERENCE itself at the FULL model dimensions (run in the build container only).

    python tests/golden/make_golden_r4.py [case ...]

Rounds 1-3 pinned the oracle to the reference on tiny configurations only (hidden 64, 4 heads x 16, pos-conv k = 16 g = 4,
T5 with 8 buckets); full-dimension parity then rested on the restatement.  Here `ref:speechmix/hf_model.py:185-447`
(`HFSpeechMixEED`) and `:505-583` (`HFSpeechMixSelf.cal_loss`) run at the dimensions that are benchmarked:

  full_cfg2_1x10s   wav2vec2-base -> bart-base, down_scale 2, ONE 10 s clip, 32 labels  (BASELINE config 1 / 2's shape)
  full_cfg2_2x3s    same model, 2 clips x 3 s, 8 labels with ignored positions
  full_cfg4_2x2s    hubert-large-ll60k -> mbart-large-50 (d 1024, 24 stable-LN layers, "layer" CNN, V 250 054), down_scale 8
  full_cfg5_2x2s    HFSpeechMixSelf wav2vec2-large (12 of 24 layers) -> t5-large (32 buckets, frozen), CE + KLD + MSE
  adapter_tiny      the reference's SpeechMixAdapter twin with its forward hooks re-registered (the reference's lambdas capture
                    their loop variables late and return a nested tuple: ref:speechmix/hf_model.py:497-500 - the fixture pins the
                    INTENDED math: adapter i applied to the hidden-state output of LM layer i)

Weights are NOT stored (0.9 - 3.7 GB): they come from this repository's seeded initialiser (`SpeechMixEED(..., init_seed=0)`,
the weights `bench.py` trains), are loaded into the reference classes with `load_state_dict`, and the tests regenerate them
from the same seed; a few sampled entries per checked tensor guard the regeneration.  Stored per case: loss, arg-max ids, the
logits at 64 seeded vocabulary positions per token (+ each token's max / lse), 256 seeded entries + L2 norm of every hidden
state, and the L2 norm + absolute maximum + 64 seeded entries of EVERY parameter's gradient.  Data only.
"""
import importlib.util
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)
sys.path.insert(0, OUT)

from make_golden import load_ref_hf, save_tokenizer  # noqa: E402

from tests.full_dim_util import CASES, N_GRAD, N_HID, N_VOC, N_W, build_ours, case_inputs, seeded_idx  # noqa: E402


def hf_dirs(c, ours, tmp):
    """Random-init HF checkpoints of the case's architecture (the reference only loads by `from_pretrained(path)`)."""
    from transformers import (BartConfig, BartForConditionalGeneration, HubertConfig, HubertModel, MBartConfig,
                              MBartForConditionalGeneration, T5Config, T5ForConditionalGeneration, Wav2Vec2Config, Wav2Vec2Model)
    ec, lc = ours.encoder_model.config, ours.decoder_model.config
    ed = ec.to_dict()
    ed.pop("model_type")
    if ec.model_type == "hubert":
        enc_dir = os.path.join(tmp, "hubert_enc")
        HubertModel(HubertConfig(**ed)).save_pretrained(enc_dir)
    else:
        enc_dir = os.path.join(tmp, "w2v2_enc")
        Wav2Vec2Model(Wav2Vec2Config(**ed)).save_pretrained(enc_dir)
    ld = lc.to_dict()
    mt = ld.pop("model_type")
    lm_dir = os.path.join(tmp, mt + "_lm")
    if mt == "t5":
        cfg = T5Config(vocab_size=lc.vocab_size, d_model=lc.d_model, d_kv=lc.d_kv, d_ff=lc.encoder_ffn_dim, num_layers=lc.encoder_layers,
                       num_decoder_layers=lc.decoder_layers, num_heads=lc.encoder_attention_heads,
                       relative_attention_num_buckets=lc.relative_attention_num_buckets,
                       relative_attention_max_distance=lc.relative_attention_max_distance, layer_norm_epsilon=lc.layer_norm_epsilon,
                       feed_forward_proj="relu", tie_word_embeddings=lc.tie_word_embeddings, pad_token_id=lc.pad_token_id,
                       eos_token_id=lc.eos_token_id, decoder_start_token_id=lc.decoder_start_token_id)
        T5ForConditionalGeneration(cfg).save_pretrained(lm_dir)
    else:
        keep = {k: v for k, v in ld.items() if k not in ("d_kv", "relative_attention_num_buckets", "relative_attention_max_distance",
                                                         "layer_norm_epsilon", "is_gated_act", "tie_word_embeddings", "max_length")}
        cfg = (MBartConfig if mt == "mbart" else BartConfig)(**keep)
        (MBartForConditionalGeneration if mt == "mbart" else BartForConditionalGeneration)(cfg).save_pretrained(lm_dir)
    save_tokenizer(lm_dir, 64, lc.pad_token_id, 0, lc.eos_token_id)      # (the compute path never calls it)
    return enc_dir, lm_dir


def load_weights(ref_model, ours):
    sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
    own = ref_model.state_dict()
    use = {k: v for k, v in sd.items() if k in own}
    skipped = sorted(k for k in sd if k not in own)
    res = ref_model.load_state_dict(use, strict=False)
    assert all(k.startswith(("weights_sum", "nlp_emb")) for k in skipped), skipped
    # whatever the reference holds that our state dict lacks must be an alias of a tensor we did load (tied embeddings)
    ptrs = {own[k].data_ptr() for k in use}
    for k in res.missing_keys:
        assert own[k].data_ptr() in ptrs, f"reference tensor {k} received no weights"
    return sd


def summarise(name, t, n, store):
    """Append (name, n seeded entries, L2 norm, absolute maximum) of t to the group `name.split("::")[0]` of the store."""
    grp, key = name.split("::", 1)
    t = t.detach().float().reshape(-1)
    idx = seeded_idx(name, t.numel(), n)
    g = store.setdefault("_" + grp, dict(names=[], idx=[], val=[], norm=[], amax=[]))
    g["names"].append(key)
    g["idx"].append(idx.numpy())
    g["val"].append(t[idx].numpy())
    g["norm"].append(t.double().norm().item())
    g["amax"].append(t.abs().max().item())


def pack_groups(store):
    """{_grp: lists} -> rectangular arrays grp_names / grp_idx / grp_val / grp_norm / grp_amax."""
    for k in [k for k in store if k.startswith("_")]:
        g = store.pop(k)
        grp = k[1:]
        store[grp + "_names"] = np.array(g["names"])
        store[grp + "_idx"] = np.stack(g["idx"]).astype(np.int64)
        store[grp + "_val"] = np.stack(g["val"]).astype(np.float32)
        store[grp + "_norm"] = np.array(g["norm"], dtype=np.float64)
        store[grp + "_amax"] = np.array(g["amax"], dtype=np.float32)


def run_case(name, c, ref):
    print(f"== {name}", flush=True)
    tmp = tempfile.mkdtemp(prefix="smx_r4_")
    try:
        ours = build_ours(c)
        enc_dir, lm_dir = hf_dirs(c, ours, tmp)
        cls = ref.HFSpeechMixSelf if c["kind"] == "self" else ref.HFSpeechMixEED
        model = cls(enc_dir, lm_dir, share_layer_ratio=c["share"], down_scale=c["ds"]).eval()
        sd = load_weights(model, ours)
        lc = ours.decoder_model.config
        wave, labels, text = case_inputs(c, lc.vocab_size)
        store = {}
        # guards for the regenerated weights / inputs
        for k in sorted(sd)[:: max(1, len(sd) // 24)]:
            if sd[k].is_floating_point():
                summarise("w::" + k, sd[k], N_W, store)
        store["wave_head"] = wave[:, :64].numpy()
        store["labels"] = labels.numpy()
        if text is not None:
            store["text_input_ids"] = text.numpy()
        del ours
        model.zero_grad()
        cap = {}
        # (the reference overwrites `outputs["logits"]` with their arg-max in place: keep what the FIRST LM call returned)
        hooks = [model.decoder_model.register_forward_hook(
            lambda m, i, o: cap.setdefault("lm", dict(logits=o.logits.detach().clone(), enc=o.encoder_last_hidden_state.detach().clone()))
            and None)]
        enc_out = model.encoder_model(wave, output_hidden_states=True)
        if c["kind"] == "eed":
            hooks.append(model.length_adapters.register_forward_hook(lambda m, i, o: cap.update(post_adapter=o.transpose(1, 2))))
            hooks.append(model.enc_to_dec_proj.register_forward_hook(lambda m, i, o: cap.update(inputs_embeds=o)))
            out = model(input_values=wave, labels=labels)
            loss = out["loss"]
            lm_out = cap["lm"]
            argmax = out["logits"]
        else:
            # forward() of the Self class is broken at this commit: cal_loss on the reference's own pre-LM pipeline
            feats = model.encoder_model(wave).last_hidden_state
            post = model.length_adapters(feats.transpose(1, 2)).transpose(1, 2)
            emb = model.enc_to_dec_proj(post)
            cap.update(post_adapter=post, inputs_embeds=emb)
            dec_in = ref.shift_tokens_right(labels, lc.pad_token_id, lc.decoder_start_token_id)
            out = model.cal_loss(inputs_embeds=emb, text_input_ids=text, decoder_input_ids=dec_in, labels=labels)
            loss = out["loss"]
            lm_out = cap["lm"]                       # (first LM call: the speech pass)
            argmax = out["logits"].argmax(-1)
            with torch.no_grad():
                store["ce"] = np.float64(model.decoder_model(inputs_embeds=emb, decoder_input_ids=dec_in, labels=labels).loss.item())
        loss.backward()
        for h in hooks:
            h.remove()
        logits = lm_out["logits"].float()                      # [B, L, V]
        B, L, V = logits.shape
        vidx = seeded_idx(name + "::vocab", V, N_VOC)
        store["vocab_idx"] = vidx.numpy().astype(np.int32)
        store["logits_at"] = logits[:, :, vidx].numpy()
        store["logits_max"] = logits.max(-1).values.numpy()
        store["logits_lse"] = torch.logsumexp(logits.double(), -1).numpy()
        top2 = logits.topk(2, -1).values
        store["logits_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
        store["argmax"] = argmax.numpy()
        store["loss"] = np.float64(loss.item())
        for i, h in enumerate(enc_out.hidden_states):
            summarise(f"h::enc_hidden_{i}", h, N_HID, store)
        summarise("h::encoder_last_hidden_state", enc_out.last_hidden_state, N_HID, store)
        summarise("h::post_adapter", cap["post_adapter"], N_HID, store)
        summarise("h::inputs_embeds", cap["inputs_embeds"], N_HID, store)
        summarise("h::lm_encoder_last_hidden", lm_out["enc"], N_HID, store)
        ngrads = 0
        for k, p in model.named_parameters():
            if p.grad is not None:
                summarise("g::" + k, p.grad, N_GRAD, store)
                ngrads += 1
        pack_groups(store)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
        meta = dict(c, loss=float(loss.item()), gradients=ngrads, speech_encoder_layer=model.speech_encoder_layer,
                    route="hf_model." + cls.__name__ + (".cal_loss" if c["kind"] == "self" else ""),
                    params_M=round(sum(p.numel() for p in model.parameters()) / 1e6, 2))
        print("   loss", meta["loss"], "gradients", ngrads, "params (M)", meta["params_M"], flush=True)
        return meta
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def adapter_case(ref):
    """Tiny BART LM with the reference's SpeechMixAdapter twin; hooks re-registered with the loop variables bound per hook."""
    from make_golden import cfg_dict, tiny_lm, tiny_speech, to_np
    tmp = tempfile.mkdtemp(prefix="smx_r4_")
    try:
        torch.manual_seed(77)
        enc_dir, ecfg = tiny_speech("w2v2", tmp)
        lm_dir, lcfg = tiny_lm("bart", tmp)
        model = ref.HFSpeechMixAdapter(enc_dir, lm_dir, down_scale=2).eval()
        base = model.decoder_model.base_model
        stacks = [base.encoder.layers, base.decoder.layers]
        for s in stacks:                                  # drop the reference's late-binding hooks ...
            for layer in s:
                layer._forward_hooks.clear()
        for s_i, s in enumerate(stacks):                  # ... and register what they were meant to be
            for l_i, layer in enumerate(s):
                ad = model.adapters[s_i * len(s) + l_i]

                def hook(m, i, o, ad=ad):
                    return (ad(o[0]),) + tuple(o[1:]) if isinstance(o, tuple) else ad(o)
                layer.register_forward_hook(hook)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.ndim == 1:
                    p.add_(torch.randn_like(p) * 0.05)
        x = torch.randn(2, 8000) * 0.1
        labels = torch.randint(4, 128, (2, 6))
        labels[1, -1] = -100
        cap = {}
        h = model.decoder_model.register_forward_hook(lambda m, i, o: cap.update(logits=o.logits.detach().clone()))
        model.zero_grad()
        out = model(input_values=x, labels=labels)
        h.remove()
        out["loss"].backward()
        named = dict(model.named_parameters())
        grads = ["adapters.0.1.weight", "adapters.1.3.bias", "adapters.3.0.weight", "adapters.2.3.weight", "enc_to_dec_proj.weight",
                 "encoder_model.encoder.layers.1.attention.q_proj.weight"]
        np.savez_compressed(f"{OUT}/adapter_tiny.npz", input_values=x.numpy(), labels=labels.numpy(),
                            **{"w::" + k: v for k, v in to_np(model.state_dict()).items()},
                            **{"o::raw_logits": cap["logits"].numpy(), "o::logits": out["logits"].numpy(),
                               "o::loss": out["loss"].detach().numpy()},
                            **{"o::grad::" + g: named[g].grad.numpy() for g in grads})
        frozen = sorted(n for n, p in model.named_parameters() if not p.requires_grad)
        print("   adapter loss", float(out["loss"]), "frozen tensors", len(frozen))
        return {"enc_cfg": cfg_dict(ecfg), "lm_cfg": cfg_dict(lcfg), "down_scale": 2, "share_layer_ratio": 0,
                "n_frozen": len(frozen), "n_adapters": len(model.adapters),
                "route": "hf_model.HFSpeechMixAdapter (forward hooks re-registered with bound indices)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    torch.set_num_threads(8)
    ref = load_ref_hf()
    want = sys.argv[1:] or list(CASES) + ["adapter_tiny"]
    path = os.path.join(OUT, "manifest_r4.json")
    manifest = json.load(open(path)) if os.path.exists(path) else {}
    for name in want:
        manifest[name] = adapter_case(ref) if name == "adapter_tiny" else run_case(name, CASES[name], ref)
        with open(path, "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", want)


if __name__ == "__main__":
    main()
