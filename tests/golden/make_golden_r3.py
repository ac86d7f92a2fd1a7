"""Round-3 golden fixtures, generated from the REFERENCE itself (run in the build container only).

    python tests/golden/make_golden_r3.py

Train-mode behaviour of ref:speechmix/hf_model.py HFSpeechMixEED (the reference trains under HF Trainer, i.e. in
`.train()`: ref:train.py:315-330, ref:speechmix/hf_model.py:397) with every dropout probability 0, so that the forward is a
deterministic function of the two host random streams HF draws from:
  * eed_train_specaug.npz   - mask_time_prob 0.3 / mask_time_length 4 / layerdrop 0, under np.random.seed(k) + torch.manual_seed(k):
    the boolean SpecAugment mask HF's `_compute_mask_indices` returned (TF:models/wav2vec2/modeling_wav2vec2.py:101-218), the
    per-layer LayerDrop draws, hidden states, logits, loss, gradients (incl. `masked_spec_embed`)
  * eed_train_layerdrop.npz - layerdrop 0.5 / mask_time_prob 0: per-layer keep decisions (TF:...wav2vec2.py:709-723), logits,
    loss, gradients of a kept and a dropped layer
  * lm_attention_mask.npz   - the reference's LM hook with a mask: `decoder_model(inputs_embeds=, attention_mask=, decoder_input_ids=,
    labels=)` exactly as ref:speechmix/model.py:132-136 `cal_loss` calls it, ragged right-padded mask: logits, loss, the text
    encoder's last hidden state, gradients wrt LM weights and wrt inputs_embeds
  * w2v2_attention_mask.npz / hubert_attention_mask.npz - the speech encoder of HFSpeechMixEED called with a ragged sample-level
    `attention_mask` (TF:models/wav2vec2/modeling_wav2vec2.py:1041-1060, 1349-1358, 688-697): last hidden state, all hidden states
  * mask_indices.npz        - `_compute_mask_indices` alone on a table of (shape, prob, length, min_masks, lengths, seed) cases,
    including ragged `attention_mask` lengths (integer work: the port must match bit for bit)
Fixtures are data only: weights, inputs, outputs.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

OUT = HERE
ZERO_DROP_ENC = dict(hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0)
ZERO_DROP_LM = dict(dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, classifier_dropout=0.0)


def tiny_speech_train(d, name, **kw):
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    cfg = Wav2Vec2Config(hidden_size=64, num_hidden_layers=4, num_attention_heads=4, intermediate_size=128, conv_dim=(32,) * 7,
                         num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, vocab_size=32, **ZERO_DROP_ENC, **kw)
    path = os.path.join(d, name)
    Wav2Vec2Model(cfg).save_pretrained(path)
    return path, cfg


def tiny_bart_train(d):
    from transformers import BartConfig, BartForConditionalGeneration
    cfg = BartConfig(vocab_size=128, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                     decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, max_position_embeddings=256,
                     pad_token_id=1, bos_token_id=0, eos_token_id=2, decoder_start_token_id=2, **ZERO_DROP_LM)
    path = os.path.join(d, "bart_tiny_train")
    m = BartForConditionalGeneration(cfg)
    with torch.no_grad():
        m.final_logits_bias.normal_(0, 0.02)
    m.save_pretrained(path)
    return path, cfg


def run_train_case(ref, enc_dir, lm_dir, seed, grads):
    """One train-mode forward + backward of the reference under pinned host streams; records what HF drew."""
    import transformers.models.wav2vec2.modeling_wav2vec2 as W
    model = ref.HFSpeechMixEED(enc_dir, lm_dir, down_scale=2)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
        if hasattr(model.encoder_model, "masked_spec_embed"):
            model.encoder_model.masked_spec_embed.normal_(0, 0.5)
    model.train()
    x = torch.randn(2, 16000) * 0.1
    labels = torch.randint(3, 128, (2, 6)); labels[1, -1] = -100
    rec = {"mask": None, "called": []}
    orig = W._compute_mask_indices

    def spy(*a, **k):
        m = orig(*a, **k)
        rec["mask"] = np.array(m, dtype=bool)
        return m
    W._compute_mask_indices = spy
    hooks = [l.register_forward_hook(lambda mod, i, o, idx=idx: rec["called"].append(idx))
             for idx, l in enumerate(model.encoder_model.encoder.layers)]
    hid = {}
    hooks.append(model.encoder_model.encoder.register_forward_hook(lambda m, i, o: hid.update(enc=o.last_hidden_state.detach().clone())))
    try:
        np.random.seed(seed)
        torch.manual_seed(seed)
        r = G.capture_eed(model, x, labels, grads)
    finally:
        W._compute_mask_indices = orig
        for h in hooks:
            h.remove()
    L = len(model.encoder_model.encoder.layers)
    keep = np.array([i in rec["called"] for i in range(L)], dtype=bool)
    # replay of the LayerDrop stream alone: the draws HF made, for the record
    torch.manual_seed(seed)
    draws = np.array([torch.rand([]).item() for _ in range(L)], dtype=np.float64)
    # gradients of dropped layers are None in the reference: stored as zeros + a flag
    named = dict(model.named_parameters())
    none_grads = [g for g in grads if named[g].grad is None]
    return model, x, labels, r, rec["mask"], keep, draws, hid["enc"], none_grads


def main():
    torch.manual_seed(2468)
    ref = G.load_ref_hf()
    tmp = tempfile.mkdtemp()
    manifest = {}
    lm_dir, lcfg = tiny_bart_train(tmp)

    # monkeypatch-free handling of None grads: capture_eed clones .grad, so give dropped layers a zero grad afterwards
    orig_capture = G.capture_eed

    def capture(model, x, labels, grads_of, **fw):
        named = dict(model.named_parameters())
        real = [g for g in grads_of]
        # run once to find which grads are None
        model.zero_grad()
        out_hooks = {}
        lm_out = {}
        h = model.decoder_model.register_forward_hook(
            lambda mod, inp, out: lm_out.update(logits=out.logits.detach().clone(), enc=out.encoder_last_hidden_state.detach().clone()))
        cap = {}
        hk = [model.enc_to_dec_proj.register_forward_hook(lambda m, i, o: cap.update(inputs_embeds=o.detach().clone())),
              model.encoder_model.feature_projection.register_forward_hook(
                  lambda m, i, o: cap.update(feature_projection=(o[0] if isinstance(o, tuple) else o).detach().clone()))]
        out = model(input_values=x, labels=labels, **fw)
        res = {"raw_logits": lm_out["logits"], "lm_encoder_last_hidden": lm_out["enc"], "logits": out["logits"].detach(),
               "loss": out["loss"].detach()}
        out["loss"].backward()
        for g in real:
            gr = named[g].grad
            res["grad::" + g] = torch.zeros_like(named[g]) if gr is None else gr.detach().clone()
        h.remove()
        for k in hk:
            k.remove()
        res.update(cap)
        return res
    G.capture_eed = capture

    # ---------------- SpecAugment pinned -----------------------------------------------------------------------
    enc_dir, ecfg = tiny_speech_train(tmp, "w2v2_specaug", mask_time_prob=0.3, mask_time_length=4, mask_time_min_masks=2, layerdrop=0.0)
    grads = ["encoder_model.masked_spec_embed", "enc_to_dec_proj.weight", "length_adapters.0.weight",
             "encoder_model.feature_projection.projection.weight", "encoder_model.encoder.layers.1.attention.q_proj.weight",
             "encoder_model.encoder.pos_conv_embed.conv.parametrizations.weight.original1",
             "encoder_model.feature_extractor.conv_layers.0.conv.weight"]
    model, x, labels, r, mask, keep, draws, enc_hidden, none_g = run_train_case(ref, enc_dir, lm_dir, 11, grads)
    assert mask is not None and mask.any() and keep.all() and not none_g
    np.savez_compressed(f"{OUT}/eed_train_specaug.npz", input_values=x.numpy(), labels=labels.numpy(), seed=np.int64(11),
                        spec_mask=mask, layer_keep=keep, layerdrop_draws=draws, encoder_hidden=enc_hidden.numpy(),
                        **{"w::" + k: v for k, v in G.to_np(model.state_dict()).items()},
                        **{"o::" + k: v.numpy() for k, v in r.items()})
    manifest["eed_train_specaug"] = {"enc_cfg": G.cfg_dict(ecfg), "lm_cfg": G.cfg_dict(lcfg), "down_scale": 2, "share_layer_ratio": 0,
                                     "route": "hf_model.HFSpeechMixEED.train()"}
    print("specaug: masked frames", int(mask.sum()), "of", mask.size, "loss", float(r["loss"]))

    # ---------------- LayerDrop pinned ---------------------------------------------------------------------------
    enc_dir2, ecfg2 = tiny_speech_train(tmp, "w2v2_layerdrop", mask_time_prob=0.0, layerdrop=0.5)
    for seed in range(3, 40):       # a seed that keeps some layers and drops others
        torch.manual_seed(seed)
        d = [torch.rand([]).item() < 0.5 for _ in range(4)]
        if any(d) and not all(d):
            break
    dropped = [i for i, v in enumerate(d) if v]
    kept = [i for i, v in enumerate(d) if not v]
    grads2 = ["enc_to_dec_proj.weight", f"encoder_model.encoder.layers.{kept[0]}.feed_forward.intermediate_dense.weight",
              f"encoder_model.encoder.layers.{dropped[0]}.feed_forward.intermediate_dense.weight",
              f"encoder_model.encoder.layers.{dropped[0]}.attention.k_proj.bias",
              "encoder_model.encoder.layer_norm.weight", "encoder_model.feature_projection.projection.weight"]
    model2, x2, labels2, r2, mask2, keep2, draws2, enc_hidden2, none_g2 = run_train_case(ref, enc_dir2, lm_dir, seed, grads2)
    assert mask2 is None or not mask2.any()
    assert list(np.flatnonzero(~keep2)) == dropped, (keep2, dropped)
    np.savez_compressed(f"{OUT}/eed_train_layerdrop.npz", input_values=x2.numpy(), labels=labels2.numpy(), seed=np.int64(seed),
                        layer_keep=keep2, layerdrop_draws=draws2, encoder_hidden=enc_hidden2.numpy(),
                        none_grads=np.array([grads2.index(g) for g in none_g2], dtype=np.int64),
                        **{"w::" + k: v for k, v in G.to_np(model2.state_dict()).items()},
                        **{"o::" + k: v.numpy() for k, v in r2.items()})
    manifest["eed_train_layerdrop"] = {"enc_cfg": G.cfg_dict(ecfg2), "lm_cfg": G.cfg_dict(lcfg), "down_scale": 2, "share_layer_ratio": 0,
                                       "route": "hf_model.HFSpeechMixEED.train()", "grads": grads2}
    print("layerdrop: keep", keep2.tolist(), "seed", seed, "loss", float(r2["loss"]), "None grads:", none_g2)

    # ---------------- attention masks: the LM hook and the speech encoder ------------------------------------------
    torch.manual_seed(77)
    enc_dir3, ecfg3 = G.tiny_speech("w2v2", tmp)
    lm_dir3, lcfg3 = G.tiny_lm("bart", tmp)
    m3 = ref.HFSpeechMixEED(enc_dir3, lm_dir3, down_scale=2).eval()
    with torch.no_grad():
        for n, p in m3.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
    B, S, d = 3, 11, lcfg3.d_model
    emb = (torch.randn(B, S, d) * 0.5).requires_grad_(True)
    lens = [11, 6, 2]
    am = torch.zeros(B, S, dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, :n] = 1
    labels3 = torch.randint(3, 128, (B, 5)); labels3[2, -2:] = -100
    dec3 = ref.shift_tokens_right(labels3, lcfg3.pad_token_id, lcfg3.decoder_start_token_id) if hasattr(ref, "shift_tokens_right") else None
    m3.zero_grad()
    kw = dict(inputs_embeds=emb, attention_mask=am, labels=labels3)
    if dec3 is not None:
        kw["decoder_input_ids"] = dec3
    o3 = m3.decoder_model(**kw)                                  # what cal_loss does (ref:speechmix/model.py:135-136)
    o3.loss.backward()
    named3 = dict(m3.named_parameters())
    g3 = ["decoder_model.model.encoder.layers.0.self_attn.k_proj.weight", "decoder_model.model.decoder.layers.1.encoder_attn.v_proj.weight",
          "decoder_model.model.decoder.layers.0.encoder_attn.q_proj.bias", "decoder_model.model.shared.weight",
          "decoder_model.model.encoder.layers.1.fc1.weight"]
    np.savez_compressed(f"{OUT}/lm_attention_mask.npz", inputs_embeds=emb.detach().numpy(), attention_mask=am.numpy(), labels=labels3.numpy(),
                        **{"w::" + k: v for k, v in G.to_np(m3.state_dict()).items()},
                        **{"o::raw_logits": o3.logits.detach().numpy(), "o::loss": o3.loss.detach().numpy(),
                           "o::lm_encoder_last_hidden": o3.encoder_last_hidden_state.detach().numpy(),
                           "o::grad::inputs_embeds": emb.grad.numpy()},
                        **{"o::grad::" + k: named3[k].grad.numpy() for k in g3})
    manifest["lm_attention_mask"] = {"enc_cfg": G.cfg_dict(ecfg3), "lm_cfg": G.cfg_dict(lcfg3), "down_scale": 2, "share_layer_ratio": 0,
                                     "route": "hf_model.HFSpeechMixEED.decoder_model(inputs_embeds, attention_mask)"}
    print("lm mask loss", float(o3.loss))

    for kind in ("w2v2", "hubert"):
        ed, ec_ = G.tiny_speech(kind, tmp)
        ld, lc_ = G.tiny_lm("bart", tmp)
        mm = ref.HFSpeechMixEED(ed, ld, down_scale=2).eval()
        with torch.no_grad():
            for n, p in mm.named_parameters():
                if p.ndim == 1:
                    p.add_(torch.randn_like(p) * 0.05)
        nsamp = [8000, 5200, 2400]
        xw = torch.zeros(3, 8000)
        sm = torch.zeros(3, 8000, dtype=torch.long)
        for b, n in enumerate(nsamp):
            xw[b, :n] = torch.randn(n) * 0.1
            sm[b, :n] = 1
        with torch.no_grad():
            eo = mm.encoder_model(xw, attention_mask=sm, output_hidden_states=True)
        np.savez_compressed(f"{OUT}/{kind}_attention_mask.npz", input_values=xw.numpy(), attention_mask=sm.numpy(),
                            sample_lengths=np.array(nsamp, dtype=np.int64),
                            **{"w::" + k: v for k, v in G.to_np(mm.state_dict()).items()},
                            **{"o::encoder_last_hidden_state": eo.last_hidden_state.numpy(),
                               "o::hidden_states": torch.stack(eo.hidden_states, 0).numpy()})
        manifest[f"{kind}_attention_mask"] = {"enc_cfg": G.cfg_dict(ec_), "lm_cfg": G.cfg_dict(lc_), "down_scale": 2, "share_layer_ratio": 0,
                                              "route": "hf_model.HFSpeechMixEED.encoder_model(input_values, attention_mask)"}
        print(kind, "masked encoder out", tuple(eo.last_hidden_state.shape))

    # ---------------- _compute_mask_indices table ----------------------------------------------------------------
    import transformers.models.wav2vec2.modeling_wav2vec2 as W
    rows, masks = [], {}
    table = [((4, 499), 0.05, 10, 2, None), ((32, 499), 0.05, 10, 2, None), ((3, 49), 0.3, 4, 2, None), ((2, 24), 0.65, 10, 0, None),
             ((5, 120), 0.5, 7, 1, [120, 90, 33, 8, 6]), ((4, 60), 0.2, 5, 2, [60, 4, 59, 30]), ((1, 10), 0.9, 10, 0, None),
             ((6, 200), 0.0101, 10, 0, None)]
    lens_tab = np.full((len(table) * 3, 32), -1, dtype=np.int64)
    for i, (shape, prob, length, mmin, lens) in enumerate(table):
        for seed in (0, 1, 12345):
            np.random.seed(seed)
            am = None
            if lens is not None:
                am = torch.zeros(shape, dtype=torch.long)
                for b_, n in enumerate(lens):
                    am[b_, :n] = 1
                lens_tab[len(rows), :len(lens)] = lens
            m = W._compute_mask_indices(shape, prob, length, attention_mask=am, min_masks=mmin)
            masks[f"m{len(rows)}"] = np.array(m, dtype=bool)
            # columns: B, T, prob, length, min_masks, seed, the next np.random.rand() after the call (stream position)
            rows.append([shape[0], shape[1], prob, length, mmin, seed, float(np.random.rand())])
    np.savez_compressed(f"{OUT}/mask_indices.npz", cases=np.array(rows, dtype=np.float64), lengths=lens_tab, **masks)
    print("mask table:", len(rows), "cases")

    with open(f"{OUT}/manifest_r3.json", "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
