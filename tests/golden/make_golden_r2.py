"""Round-2 golden fixtures, generated from the REFERENCE itself (run in the build container only).

    python tests/golden/make_golden_r2.py

Same routes and rules as make_golden.py (whose fixtures stay byte-identical: this script only ADDS files):
  * eed_w2v2_t5_trainable.npz  - ref:speechmix/hf_model.py HFSpeechMixEED with a TRAINABLE tiny T5: logits, loss and
    gradients including both relative-position bias tables (TF:models/t5/modeling_t5.py:216-279)
  * eed_ragged_batch.npz       - two clips of different length collated the way ref:train.py:100-133 does
    (`pad_sequence(..., padding_value=-100)`, no mask; labels padded and set to -100) through HFSpeechMixEED
  * eed_route2_prompt.npz      - ref:speechmix/model.py SpeechMixEED.forward(input_text_prompt=...) (batch 1: the
    reference's concatenation only works for one clip, ref:speechmix/model.py:168-171)
  * greedy_labels.npz          - ref:train.py:18-34 `create_self_decoder_input` (the function's own code object, lifted
    from the file with `ast` because train.py imports packages that are not installed) on tiny BART and T5
Fixtures are data only: weights, inputs, outputs.
"""
import ast
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

REF = G.REF
OUT = HERE


def main():
    torch.manual_seed(4321)
    ref = G.load_ref_hf()
    tmp = tempfile.mkdtemp()
    manifest = {}
    enc_dir, ecfg = G.tiny_speech("w2v2", tmp)

    # ---------------- trainable T5 (relative-position bias gradients) -------------------------------------------
    lm_dir, lcfg = G.tiny_lm("t5", tmp)
    model = ref.HFSpeechMixEED(enc_dir, lm_dir, down_scale=4).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
            if "relative_attention_bias" in n:
                p.normal_(0, 0.5)
    x = torch.randn(2, 8000) * 0.1
    labels = torch.randint(2, 128, (2, 7)); labels[1, -2:] = -100
    grads = ["enc_to_dec_proj.weight", "length_adapters.1.weight",
             "decoder_model.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
             "decoder_model.decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
             "decoder_model.shared.weight", "decoder_model.decoder.block.1.layer.1.EncDecAttention.q.weight",
             "decoder_model.encoder.block.1.layer.1.DenseReluDense.wi.weight",
             "decoder_model.decoder.final_layer_norm.weight",
             "encoder_model.encoder.layers.2.attention.k_proj.weight"]
    r = G.capture_eed(model, x, labels, grads)
    np.savez_compressed(f"{OUT}/eed_w2v2_t5_trainable.npz", input_values=x.numpy(), labels=labels.numpy(),
                        **{"w::" + k: v for k, v in G.to_np(model.state_dict()).items()},
                        **{"o::" + k: v.numpy() for k, v in r.items()})
    manifest["eed_w2v2_t5_trainable"] = {"enc_cfg": G.cfg_dict(ecfg), "lm_cfg": G.cfg_dict(lcfg), "down_scale": 4,
                                         "share_layer_ratio": 0, "route": "hf_model.HFSpeechMixEED"}
    print("t5 trainable loss", float(r["loss"]))

    # ---------------- ragged batch, collated like ref:train.py:100-133 -----------------------------------------
    lm_dir_b, lcfg_b = G.tiny_lm("bart", tmp)
    model_b = ref.HFSpeechMixEED(enc_dir, lm_dir_b, down_scale=2).eval()
    with torch.no_grad():
        for n, p in model_b.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
    clips = [torch.randn(8000) * 0.1, torch.randn(5200) * 0.1, torch.randn(6731) * 0.1]
    lab_rows = [[17, 5, 99, 42, 2], [8, 2], [64, 64, 7, 2]]
    xb = pad_sequence(clips, batch_first=True, padding_value=-100)                       # the collator's waveform padding
    n = max(len(r_) for r_ in lab_rows)
    lb = torch.full((len(lab_rows), n), lcfg_b.pad_token_id, dtype=torch.int64)
    mask = torch.zeros_like(lb)
    for i, r_ in enumerate(lab_rows):
        lb[i, :len(r_)] = torch.tensor(r_); mask[i, :len(r_)] = 1
    lb = lb.masked_fill(mask.ne(1), -100)                                                # labels -> -100 on the padding
    grads_b = ["enc_to_dec_proj.weight", "length_adapters.0.weight",
               "encoder_model.feature_extractor.conv_layers.0.conv.weight",
               "encoder_model.feature_extractor.conv_layers.0.layer_norm.weight",
               "encoder_model.encoder.layers.3.feed_forward.output_dense.weight", "decoder_model.model.shared.weight"]
    hs = model_b.encoder_model(xb)
    rb = G.capture_eed(model_b, xb, lb, grads_b)
    rb["encoder_last_hidden_state"] = hs.last_hidden_state.detach()
    np.savez_compressed(f"{OUT}/eed_ragged_batch.npz", input_values=xb.numpy(), labels=lb.numpy(),
                        lengths=np.array([len(c) for c in clips]),
                        **{f"clip{i}": c.numpy() for i, c in enumerate(clips)},
                        **{"w::" + k: v for k, v in G.to_np(model_b.state_dict()).items()},
                        **{"o::" + k: v.numpy() for k, v in rb.items()})
    manifest["eed_ragged_batch"] = {"enc_cfg": G.cfg_dict(ecfg), "lm_cfg": G.cfg_dict(lcfg_b), "down_scale": 2,
                                    "share_layer_ratio": 0, "label_rows": lab_rows,
                                    "route": "hf_model.HFSpeechMixEED on the ref:train.py collator's padding"}
    print("ragged loss", float(rb["loss"]))

    # ---------------- Route 2: text prompt through ref:speechmix/model.py -----------------------------------------
    from transformers import Wav2Vec2Model

    class _Upstream(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Wav2Vec2Model.from_pretrained(enc_dir)
            self.model.final_proj = torch.nn.Linear(self.model.config.hidden_size, 8)

        def forward(self, wavs):
            o = self.model(wavs if torch.is_tensor(wavs) else torch.stack(list(wavs)), output_hidden_states=True)
            return {"last_hidden_state": o.last_hidden_state, "hidden_states": o.hidden_states}
    hub = types.ModuleType("s3prl.hub"); hub.wav2vec2 = _Upstream
    pkg = types.ModuleType("s3prl"); pkg.hub = hub
    sys.modules["s3prl"] = pkg; sys.modules["s3prl.hub"] = hub
    spec = importlib.util.spec_from_file_location("ref_model", f"{REF}/speechmix/model.py")
    refm = importlib.util.module_from_spec(spec); spec.loader.exec_module(refm)
    m4 = refm.SpeechMixEED("wav2vec2", lm_dir_b, down_scale=2).eval()
    prompt = "t9 t33 t101"
    x1 = torch.randn(1, 8000) * 0.1
    l1 = torch.randint(4, 128, (1, 6))
    cap = {}
    h = m4.decoder_model.register_forward_hook(lambda m, i, o: cap.update(logits=o.logits.detach().clone()))
    o4 = m4(x1, input_text_prompt=prompt, labels=l1)
    h.remove()
    prompt_ids = m4.tokenizer(prompt, return_tensors="pt")["input_ids"].reshape(-1)
    sd4 = {}
    for k, v in m4.state_dict().items():
        k = k.replace("encoder_model.model.", "encoder_model.")
        if "final_proj" in k:
            continue
        sd4[k] = v
    np.savez_compressed(f"{OUT}/eed_route2_prompt.npz", input_values=x1.numpy(), labels=l1.numpy(),
                        prompt_ids=prompt_ids.numpy(),
                        **{"w::" + k: v for k, v in G.to_np(sd4).items()},
                        **{"o::raw_logits": cap["logits"].numpy(), "o::logits": o4["logits"].numpy(),
                           "o::loss": o4["loss"].detach().numpy()})
    manifest["eed_route2_prompt"] = {"enc_cfg": G.cfg_dict(ecfg), "lm_cfg": G.cfg_dict(lcfg_b), "down_scale": 2,
                                     "share_layer_ratio": 0, "prompt": prompt,
                                     "route": "model.SpeechMixEED(input_text_prompt=...) (s3prl.hub stand-in)"}
    print("prompt loss", float(o4["loss"]), "prompt ids", prompt_ids.tolist())

    # ---------------- greedy label creation, ref:train.py:18-34 ---------------------------------------------------
    src = open(f"{REF}/train.py").read()
    fn = next(nd for nd in ast.parse(src).body if isinstance(nd, ast.FunctionDef) and nd.name == "create_self_decoder_input")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), f"{REF}/train.py", "exec"), ns)
    create = ns["create_self_decoder_input"]
    from transformers import AutoModelForSeq2SeqLM, AutoTokenizer
    greedy = {}
    for kind, d_ in (("bart", lm_dir_b), ("t5", lm_dir)):
        tok = AutoTokenizer.from_pretrained(d_)
        sent = "t20 t21 t90 t7 t64 t11 t100"
        # an untrained tied-embedding LM greedy-decodes a fixed point (its own start token): weight matrices are scaled up
        # (and BART's learned positions spread) until the loop emits a varied sequence - the fixture is the data that results
        for scale, pos in ((4, 1), (8, 1), (8, 10), (8, 30), (16, 30), (3, 1), (6, 3)):
            torch.manual_seed(7)
            lm = AutoModelForSeq2SeqLM.from_pretrained(d_).eval()
            with torch.no_grad():
                for n_, p in lm.named_parameters():
                    if "relative_attention_bias" in n_:
                        p.normal_(0, 0.5)
                    elif "embed_positions" in n_:
                        p.mul_(pos)
                    elif "shared" not in n_ and "embed" not in n_ and p.ndim >= 2:
                        p.mul_(scale)
            lm.config.max_length = 12
            gen_input, predicted = create(lm, tok, sent, "cpu")
            if len(set(predicted)) >= 4:
                break
        assert len(set(predicted)) >= 4, (kind, predicted)
        greedy[kind] = dict(gen_input=gen_input, predicted=predicted)
        np.savez_compressed(f"{OUT}/greedy_labels_{kind}.npz", gen_input=np.array(gen_input), predicted=np.array(predicted),
                            max_length=np.array(12),
                            **{"w::" + k: v for k, v in G.to_np(lm.state_dict()).items()})
        manifest[f"greedy_labels_{kind}"] = {"lm_cfg": G.cfg_dict(lm.config), "max_length": 12, "sentence": sent,
                                             "route": "train.create_self_decoder_input"}
        print("greedy", kind, gen_input, predicted)

    with open(f"{OUT}/manifest_r2.json", "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote round-2 fixtures to", OUT)


if __name__ == "__main__":
    main()
