"""Shared by tests/golden/make_golden_r4.py (which runs the REFERENCE at the full model dimensions in the build container) and
the tests that replay its fixtures: the case table, the seeded inputs, the seeded sample positions and the comparison.

A fixture (tests/golden/full_cfg*.npz) holds no weights and no full tensors - the weights are this repository's seeded initial
weights (`init_seed=0`), regenerated wherever the test runs - only: loss, arg-max ids, logits at 64 seeded vocabulary positions
per token (+ per-token max, log-sum-exp, top-2 margin), and for every hidden state / every parameter gradient a row of seeded
entries with the tensor's L2 norm and absolute maximum."""
import contextlib
import io
import os
import zlib

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_HID, N_GRAD, N_VOC, N_W = 256, 64, 64, 8

CASES = {
    "full_cfg2_1x10s": dict(kind="eed", enc="facebook/wav2vec2-base", lm="facebook/bart-base", ds=2, share=0.0, B=1, N=160000, L=32),
    "full_cfg2_2x3s": dict(kind="eed", enc="facebook/wav2vec2-base", lm="facebook/bart-base", ds=2, share=0.0, B=2, N=48000, L=8),
    "full_cfg4_2x2s": dict(kind="eed", enc="hubert_large_ll60k", lm="facebook/mbart-large-50", ds=8, share=0.0, B=2, N=32000, L=8),
    "full_cfg5_2x2s": dict(kind="self", enc="wav2vec2_large_960", lm="t5-large", ds=8, share=0.5, B=2, N=32000, L=8),
}


def seeded_idx(name, numel, n):
    """n flat indices into a tensor of `numel` entries, a function of the tensor's name only."""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
    return torch.randint(numel, (n,), generator=g)


def case_inputs(c, vocab):
    g = torch.Generator().manual_seed(2024 + c["N"] + c["B"])
    wave = (torch.randn(c["B"], c["N"], generator=g) * 0.1).clamp_(-1, 1)
    labels = torch.randint(4, vocab, (c["B"], c["L"]), generator=g)
    labels[:, -1] = 2
    if c["B"] > 1:
        labels[c["B"] - 1, -3:] = -100
    text = torch.randint(4, vocab, (c["B"], c["L"] + 1), generator=g) if c["kind"] == "self" else None
    return wave, labels, text


def build_ours(c, dtype="fp32"):
    """This repository's model class with the seeded initial weights (on a GPU box: the HIP engine; on the CPU: a parameter
    container only - the compute path has no CPU fallback)."""
    from speechmix_amd.model import SpeechMixEED, SpeechMixSelf
    cls = SpeechMixSelf if c["kind"] == "self" else SpeechMixEED
    with contextlib.redirect_stdout(io.StringIO()):
        return cls(c["enc"], c["lm"], share_layer_ratio=c["share"], down_scale=c["ds"], compute_dtype=dtype, init_seed=0).eval()


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def check_regenerated(fx, sd, wave, labels, text=None):
    """The weights / inputs regenerated from their seeds are the ones the reference ran on (sampled entries, exact to 1e-6:
    the generator's transcendental functions may differ in the last bit between CPU vector widths)."""
    assert np.array_equal(fx["labels"], labels.numpy())
    assert np.abs(fx["wave_head"] - wave[:, :64].numpy()).max() <= 1e-6
    if text is not None:
        assert np.array_equal(fx["text_input_ids"], text.numpy())
    for name, idx, val in zip(fx["w_names"], fx["w_idx"], fx["w_val"]):
        got = sd[str(name)].detach().float().reshape(-1).cpu()[torch.from_numpy(idx)].numpy()
        assert np.abs(got - val).max() <= 1e-6 * max(1.0, np.abs(val).max()), str(name)


def _rows(fx, grp):
    return {str(n): (fx[grp + "_idx"][i], fx[grp + "_val"][i], float(fx[grp + "_norm"][i]), float(fx[grp + "_amax"][i]))
            for i, n in enumerate(fx[grp + "_names"])}


def compare(fx, got, grads=None, skip_hidden=()):
    """got: loss (float), argmax [B, L], logits [B, L, V] (any float tensor), hidden: {name: tensor} (names of the fixture's `h`
    group); grads: {parameter name: gradient tensor} or None.  -> dict of errors (absolute for values in their natural range,
    relative where stated)."""
    res = {}
    logits = got["logits"].detach().float().cpu()
    vidx = torch.from_numpy(fx["vocab_idx"].astype(np.int64))
    res["logits"] = float((logits[:, :, vidx] - torch.from_numpy(fx["logits_at"])).abs().max())
    res["logits_max"] = float((logits.max(-1).values - torch.from_numpy(fx["logits_max"])).abs().max())
    res["logits_lse"] = float((torch.logsumexp(logits.double(), -1) - torch.from_numpy(fx["logits_lse"])).abs().max())
    res["logits_scale"] = float(np.abs(fx["logits_at"]).max())
    res["loss"] = abs(float(got["loss"]) - float(fx["loss"]))
    res["loss_value"] = float(fx["loss"])
    safe = torch.from_numpy(fx["logits_margin"]) > 2 * max(res["logits"], res["logits_max"])
    res["argmax_checked"] = int(safe.sum())
    res["argmax_equal"] = bool((got["argmax"].cpu()[safe] == torch.from_numpy(fx["argmax"])[safe]).all())
    worst = ("", 0.0)
    for name, (idx, val, norm, amax) in _rows(fx, "h").items():
        if name in skip_hidden or name not in got["hidden"]:
            continue
        t = got["hidden"][name].detach().float().reshape(-1).cpu()
        e = float((t[torch.from_numpy(idx)] - torch.from_numpy(val)).abs().max())
        en = abs(float(t.double().norm()) - norm) / max(norm, 1e-30)
        res["h::" + name] = e
        res["hnorm::" + name] = en
        if e > worst[1]:
            worst = (name, e)
    res["hidden_checked"] = sum(1 for k in res if k.startswith("h::"))
    res["hidden_worst"], res["hidden_worst_name"] = worst[1], worst[0]
    if grads is not None:
        rows = _rows(fx, "g")
        gmax = max(a for _, _, _, a in rows.values())
        w1, w2, n = ("", 0.0), ("", 0.0), 0
        missing = []
        for name, (idx, val, norm, amax) in rows.items():
            g = grads.get(name)
            if g is None:
                missing.append(name)
                continue
            t = g.detach().float().reshape(-1).cpu()
            # sampled entries relative to the tensor's largest entry, floored at 1e-3 of the model's largest gradient entry
            # (key-projection biases have a mathematically zero gradient: what the reference holds there is rounding noise)
            e = float((t[torch.from_numpy(idx)] - torch.from_numpy(val)).abs().max()) / max(amax, 1e-3 * gmax)
            en = abs(float(t.double().norm()) - norm) / max(norm, 1e-3 * gmax * t.numel() ** 0.5)
            n += 1
            if e > w1[1]:
                w1 = (name, e)
            if en > w2[1]:
                w2 = (name, en)
        res["grads_checked"], res["grads_missing"] = n, missing
        res["grad_worst"], res["grad_worst_name"] = w1[1], w1[0]
        res["grad_norm_worst"], res["grad_norm_worst_name"] = w2[1], w2[0]
    return res


def oracle_run(c, ours, wave, labels, text, dt=torch.float32, threads=8):
    """The CPU oracle on the case -> (`got` for compare(), {name: grad})."""
    from oracle import speechmix_oracle as O
    torch.set_num_threads(max(1, min(threads, len(os.sched_getaffinity(0)))))
    ec, lc = ours.encoder_model.config.to_dict(), ours.decoder_model.config.to_dict()
    trainable = {k for k, p in ours.named_parameters() if p.requires_grad}
    leaves = {}
    for k, v in ours.state_dict().items():
        if k.endswith(("embed_tokens.weight", "lm_head.weight", "nlp_emb.weight")):      # aliases of the tied embedding
            continue
        leaves[k] = v.detach().to(dt).clone().requires_grad_(k in trainable) if v.is_floating_point() else v
    nl = ours.num_speech_encoder_layers
    w = wave.to(dt)
    if c["kind"] == "eed":
        out = O.speechmix_eed_forward(leaves, ec, lc, w, labels=labels, down_scale=c["ds"], num_speech_layers=nl)
        enc_sd, _, _ = O.split_state_dict(leaves)
        with torch.no_grad():
            _, hidden = O.speech_encoder(enc_sd, ec, w, num_layers=nl)
    else:
        enc_sd, _, rest = O.split_state_dict(leaves)
        last, hidden = O.speech_encoder(enc_sd, ec, w, num_layers=nl)
        x = O.length_adapters(rest, last, {2: 1, 4: 2, 8: 3}[c["ds"]])
        emb = x @ rest["enc_to_dec_proj.weight"].t() + rest["enc_to_dec_proj.bias"]
        dec_in = O.shift_tokens_right(labels, lc["pad_token_id"], lc["decoder_start_token_id"])
        out = O.speechmix_self_losses(leaves, lc, emb, text, dec_in, labels)
        out.update(encoder_last_hidden_state=last, post_adapter=x, inputs_embeds=emb)
    out["loss"].float().backward()
    hid = {f"enc_hidden_{i}": h for i, h in enumerate(hidden)}
    for k in ("encoder_last_hidden_state", "post_adapter", "inputs_embeds", "lm_encoder_last_hidden"):
        if k in out:
            hid[k] = out[k]
    got = dict(loss=float(out["loss"]), argmax=out["raw_logits"].argmax(-1), logits=out["raw_logits"], hidden=hid)
    grads = {k: v.grad for k, v in leaves.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
    return got, grads


def oracle_loss(c, ours, wave, labels, text, dt=torch.float32, threads=8):
    """Forward-only loss of the CPU oracle in dtype `dt` (the loss-spread yardstick of tests/golden/make_bf16_yardstick_r5.py)."""
    from oracle import speechmix_oracle as O
    torch.set_num_threads(max(1, min(threads, len(os.sched_getaffinity(0)))))
    ec, lc = ours.encoder_model.config.to_dict(), ours.decoder_model.config.to_dict()
    leaves = {k: (v.detach().to(dt) if v.is_floating_point() else v) for k, v in ours.state_dict().items()
              if not k.endswith(("embed_tokens.weight", "lm_head.weight", "nlp_emb.weight"))}
    nl = ours.num_speech_encoder_layers
    with torch.no_grad():
        if c["kind"] == "eed":
            out = O.speechmix_eed_forward(leaves, ec, lc, wave.to(dt), labels=labels, down_scale=c["ds"], num_speech_layers=nl)
        else:
            enc_sd, _, rest = O.split_state_dict(leaves)
            last, _ = O.speech_encoder(enc_sd, ec, wave.to(dt), num_layers=nl)
            x = O.length_adapters(rest, last, {2: 1, 4: 2, 8: 3}[c["ds"]])
            emb = x @ rest["enc_to_dec_proj.weight"].t() + rest["enc_to_dec_proj.bias"]
            dec_in = O.shift_tokens_right(labels, lc["pad_token_id"], lc["decoder_start_token_id"])
            out = O.speechmix_self_losses(leaves, lc, emb, text, dec_in, labels)
    return float(out["loss"])


def hip_run(c, model, wave, labels, text):
    """The HIP path on the case (needs the GPU) -> (`got`, {name: grad})."""
    kw = {"text_input_ids": text} if text is not None else {}
    out = model(wave, labels=labels, return_model_detail=True, **kw)
    out["loss"].backward()
    torch.cuda.synchronize()
    hid = {k: out[k] for k in ("encoder_last_hidden_state", "post_adapter", "inputs_embeds", "lm_encoder_last_hidden") if k in out}
    for i, h in enumerate(out.get("encoder_hidden_states", ())):
        hid[f"enc_hidden_{i}"] = h
    got = dict(loss=float(out["loss"]), argmax=out["logits"], logits=out["raw_logits"], hidden=hid)
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    return got, grads
