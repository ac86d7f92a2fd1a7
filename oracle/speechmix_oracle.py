"""CPU oracle for the SpeechMix fused training step.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the arithmetic executed by the
reference path  SpeechMixEED.forward / SpeechMixSelf.cal_loss  (ref:speechmix/model.py:139-177,
235-266; HF twin ref:speechmix/hf_model.py:378-447, 533-583).  The reference itself contains no
arithmetic: every FLOP lives in the un-vendored third-party dependency `transformers`
(requirements: `transformers>=4.12.3`; the de-facto pin is transformers 5.15.0 as installed in the
build container).  Each function below cites the `transformers` file:line (prefix TF:) whose
published algorithm it restates.

PINNING STATUS: the reference's own tests hold no numeric goldens for this path,
so this oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, generated in the build container
by `tests/golden/make_golden.py` (which imports ref:speechmix/hf_model.py by path and stores
weights/inputs/outputs under tests/golden/*.npz).  `tests/test_oracle_golden.py` replays them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (speechmix_amd/) never does.

No `transformers` import here: the oracle must run on the GPU box where only /root/repo travels.
All functions take a flat `sd` dict (HF state-dict names -> torch tensors) and plain-dict configs.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
def gelu(x: Tensor) -> Tensor:
    """Exact erf GELU.  TF:activations.py (ACT2FN["gelu"] = GELUActivation -> nn.functional.gelu)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def act_fn(name: str):
    if name == "gelu":
        return gelu
    if name == "relu":
        return torch.relu
    if name == "gelu_new":
        return lambda x: 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))
    raise ValueError(f"unsupported activation {name}")


def layer_norm(x: Tensor, w: Tensor, b: Optional[Tensor], eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps) * w
    return y + b if b is not None else y


def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """TF:models/t5/modeling_t5.py:50-72 (T5LayerNorm: no mean subtraction, no bias)."""
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def linear(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    w = sd[prefix + ".weight"]
    y = x @ w.t()
    b = sd.get(prefix + ".bias")
    return y + b if b is not None else y


def shift_tokens_right(input_ids: Tensor, pad_token_id: int, decoder_start_token_id: int) -> Tensor:
    """ref:speechmix/model.py:15-23 (integer work: bit exact)."""
    assert pad_token_id is not None, "self.model.config.pad_token_id has to be defined."
    out = torch.zeros_like(input_ids)
    out[:, 1:] = input_ids[:, :-1]
    out[:, 0] = decoder_start_token_id
    out[out == -100] = pad_token_id
    return out


def handle_decoder_input_none(decoder_start_token_id: int, batch: int = 1) -> Tensor:
    """ref:speechmix/model.py:11-12."""
    return torch.tensor([[decoder_start_token_id]] * batch, dtype=torch.long)


def mha(q: Tensor, k: Tensor, v: Tensor, n_heads: int, scale: float, causal: bool = False,
        bias: Optional[Tensor] = None, key_len: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T * scale + bias [+ causal mask] [+ key padding mask]) v.  TF:integrations/sdpa_attention.py:39-130
    and the eager twin TF:models/bart/modeling_bart.py (eager_attention_forward).  key_len [B]: keys at positions >=
    key_len[b] are padding (a right-padded `attention_mask`, the reference's hook ref:speechmix/model.py:132-136; the
    reference's own forward never passes one, ref:speechmix/model.py:148)."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    hd = D // n_heads
    qh = q.view(B, Tq, n_heads, hd).transpose(1, 2)
    kh = k.view(B, Tk, n_heads, hd).transpose(1, 2)
    vh = v.view(B, Tk, n_heads, hd).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    if causal:
        m = torch.ones(Tq, Tk, dtype=torch.bool).tril(diagonal=Tk - Tq)
        s = s.masked_fill(~m, float("-inf"))
    if key_len is not None:
        pad = torch.arange(Tk)[None, :] >= torch.as_tensor(key_len, dtype=torch.long)[:, None]            # [B, Tk]
        s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = p @ vh
    return o.transpose(1, 2).reshape(B, Tq, D)


# --------------------------------------------------------------------------------------------
# speech encoder (wav2vec2 / HuBERT family)
# --------------------------------------------------------------------------------------------
def conv_out_len(n: int, kernels, strides) -> int:
    """TF:models/wav2vec2/modeling_wav2vec2.py:997-1036 (_get_feat_extract_output_lengths)."""
    for k, s in zip(kernels, strides):
        n = (n - k) // s + 1
    return n


def feature_extractor(sd, cfg, x: Tensor, prefix="feature_extractor.", trace=None) -> Tensor:
    """TF:models/wav2vec2/modeling_wav2vec2.py:254-323, 382-419.  x [B,N] -> [B,C,T]."""
    h = x[:, None, :]
    norm = cfg["feat_extract_norm"]
    for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):
        p = f"{prefix}conv_layers.{i}."
        h = F.conv1d(h, sd[p + "conv.weight"], sd.get(p + "conv.bias"), stride=s)
        if norm == "group" and i == 0:
            # GroupNorm(num_groups=C, num_channels=C): per-(b,c) statistics over time, eps 1e-5
            mu = h.mean(-1, keepdim=True)
            var = ((h - mu) ** 2).mean(-1, keepdim=True)
            h = (h - mu) * torch.rsqrt(var + 1e-5)
            h = h * sd[p + "layer_norm.weight"][None, :, None] + sd[p + "layer_norm.bias"][None, :, None]
        elif norm == "layer":
            h = layer_norm(h.transpose(1, 2), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"],
                           1e-5).transpose(1, 2)
        h = gelu(h)
        if trace is not None:
            trace[f"conv{i}"] = h
    return h


def pos_conv_weight(sd, prefix) -> Tensor:
    """weight_norm(dim=2): w = g * v / ||v||_(dims 0,1).  TF:...wav2vec2.py:326-368."""
    g = sd[prefix + "conv.parametrizations.weight.original0"]  # [1,1,K]
    v = sd[prefix + "conv.parametrizations.weight.original1"]  # [C, C/groups, K]
    n = torch.sqrt((v * v).sum(dim=(0, 1), keepdim=True))
    return v * (g / n)


def pos_conv_embed(sd, cfg, h: Tensor, prefix="encoder.pos_conv_embed.") -> Tensor:
    """TF:...wav2vec2.py:326-379.  h [B,T,d] -> [B,T,d]."""
    K = cfg["num_conv_pos_embeddings"]
    w = pos_conv_weight(sd, prefix)
    y = F.conv1d(h.transpose(1, 2), w, sd[prefix + "conv.bias"], padding=K // 2,
                 groups=cfg["num_conv_pos_embedding_groups"])
    if K % 2 == 0:
        y = y[:, :, :-1]
    return gelu(y).transpose(1, 2)


def w2v2_layer(sd, cfg, h: Tensor, p: str, stable: bool, key_len: Optional[Tensor] = None) -> Tensor:
    """TF:...wav2vec2.py:575-608 (post-LN) / 611-654 (stable LN); attention :466-548; FFN :551-572."""
    eps = cfg["layer_norm_eps"]
    nh = cfg["num_attention_heads"]
    hd = cfg["hidden_size"] // nh

    def attn(x):
        q = linear(x, sd, p + "attention.q_proj")
        k = linear(x, sd, p + "attention.k_proj")
        v = linear(x, sd, p + "attention.v_proj")
        return linear(mha(q, k, v, nh, hd ** -0.5, key_len=key_len), sd, p + "attention.out_proj")

    def ffn(x):
        x = act_fn(cfg["hidden_act"])(linear(x, sd, p + "feed_forward.intermediate_dense"))
        return linear(x, sd, p + "feed_forward.output_dense")

    if not stable:
        h = layer_norm(h + attn(h), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps)
        h = layer_norm(h + ffn(h), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps)
    else:
        h = h + attn(layer_norm(h, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps))
        h = h + ffn(layer_norm(h, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps))
    return h


def feature_lengths(cfg: dict, sample_lengths) -> Tensor:
    """TF:models/wav2vec2/modeling_wav2vec2.py:997-1036 per clip: frames that come from real (unpadded) samples."""
    return torch.tensor([conv_out_len(int(n), cfg["conv_kernel"], cfg["conv_stride"]) for n in sample_lengths], dtype=torch.long)


def speech_encoder(sd: Dict[str, Tensor], cfg: dict, input_values: Tensor, num_layers: Optional[int] = None,
                   trace: Optional[dict] = None, spec_mask: Optional[Tensor] = None, layer_keep=None,
                   frame_lengths: Optional[Tensor] = None):
    """wav2vec2 / HuBERT forward without dropout.  Train-mode decisions are INPUTS: `spec_mask` bool [B,T] = the frames
    SpecAugment replaces with `masked_spec_embed` (TF:models/wav2vec2/modeling_wav2vec2.py:1074-1119, after the feature
    projection), `layer_keep` = per-layer LayerDrop keep flags (TF:...wav2vec2.py:709-723).  `frame_lengths` [B] = valid frames
    per clip when an `attention_mask` is given (TF:...wav2vec2.py:1041-1060): padded frames are zeroed before the positional
    convolution and masked as attention keys (TF:...wav2vec2.py:688-697, 772-781).
    TF:models/wav2vec2/modeling_wav2vec2.py:1319-1375, 667-802; TF:models/hubert/modeling_hubert.py:216-232.
    Returns (last_hidden_state [B,T,d], tuple of L+1 hidden states)."""
    eps = cfg["layer_norm_eps"]
    stable = cfg.get("do_stable_layer_norm", False)
    feats = feature_extractor(sd, cfg, input_values, trace=trace).transpose(1, 2)  # [B,T,C]
    if cfg.get("feat_proj_layer_norm", True):
        feats = layer_norm(feats, sd["feature_projection.layer_norm.weight"],
                           sd["feature_projection.layer_norm.bias"], eps)
    h = linear(feats, sd, "feature_projection.projection")
    if trace is not None:
        trace["feature_projection"] = h
    if spec_mask is not None:
        h = torch.where(torch.as_tensor(spec_mask, dtype=torch.bool)[..., None], sd["masked_spec_embed"].to(h.dtype), h)
    key_len = None
    if frame_lengths is not None:
        key_len = torch.as_tensor(frame_lengths, dtype=torch.long)
        valid = torch.arange(h.shape[1])[None, :] < key_len[:, None]
        h = h * valid[..., None].to(h.dtype)
    h = h + pos_conv_embed(sd, cfg, h)
    if not stable:
        h = layer_norm(h, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps)
    if trace is not None:
        trace["encoder_input"] = h
    L = num_layers if num_layers is not None else cfg["num_hidden_layers"]
    hidden = []
    for i in range(L):
        hidden.append(h)
        if layer_keep is not None and not bool(layer_keep[i]):
            continue
        h = w2v2_layer(sd, cfg, h, f"encoder.layers.{i}.", stable, key_len)
    if stable:
        h = layer_norm(h, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps)
    hidden.append(h)
    return h, tuple(hidden)


# --------------------------------------------------------------------------------------------
# seq2seq language models (text encoder + decoder + head)
# --------------------------------------------------------------------------------------------
def _bart_attn(sd, p, x, kv, nh, causal=False, key_len=None):
    D = x.shape[-1]
    q = linear(x, sd, p + "q_proj")
    k = linear(kv, sd, p + "k_proj")
    v = linear(kv, sd, p + "v_proj")
    return linear(mha(q, k, v, nh, (D // nh) ** -0.5, causal=causal, key_len=key_len), sd, p + "out_proj")


def lm_adapter(adapters: Optional[Dict[str, Tensor]], idx: int, x: Tensor) -> Tensor:
    """SpeechMixAdapter (ref:speechmix/model.py:196-222): every LM encoder / decoder layer's hidden-state output is
    REPLACED by Sequential(LayerNorm(d), Linear(d, d/2), ReLU, Linear(d/2, d)) of it (a forward hook, no residual).
    Adapter index = stack * layers_per_stack + layer, i.e. the value the reference's loop variables hold when each hook
    is registered (its lambdas capture them late, so as written every hook runs the LAST adapter; the intended indexing
    is restated here)."""
    if adapters is None:
        return x
    p = f"adapters.{idx}."
    h = layer_norm(x, adapters[p + "0.weight"], adapters[p + "0.bias"], 1e-5)
    h = torch.relu(h @ adapters[p + "1.weight"].t() + adapters[p + "1.bias"])
    return h @ adapters[p + "3.weight"].t() + adapters[p + "3.bias"]


def bart_like_forward(sd, cfg, inputs_embeds: Optional[Tensor], input_ids: Optional[Tensor],
                      decoder_input_ids: Tensor, trace: Optional[dict] = None, adapters=None, enc_key_len=None):
    """BART (post-LN) and mBART (pre-LN + final LNs) forward, eval mode.
    TF:models/bart/modeling_bart.py:58-98 (learned positions, offset 2), 260-390 (layers), 507-549
    (encoder), 594-676 (decoder), 939-940 (head + final_logits_bias);
    TF:models/mbart/modeling_mbart.py:273-420, 474-590, 591-760.
    Returns (logits [B,L,V], encoder_last_hidden [B,S,d])."""
    pre_ln = cfg["model_type"] == "mbart"
    d = cfg["d_model"]
    scale = math.sqrt(d) if cfg.get("scale_embedding", False) else 1.0
    act = act_fn(cfg["activation_function"])
    emb = sd["model.shared.weight"]

    def ln(x, p):
        return layer_norm(x, sd[p + ".weight"], sd[p + ".bias"], 1e-5)

    # ---- text encoder ----
    if inputs_embeds is None:
        inputs_embeds = emb[input_ids] * scale
    S = inputs_embeds.shape[1]
    h = inputs_embeds + sd["model.encoder.embed_positions.weight"][2:2 + S][None]
    h = ln(h, "model.encoder.layernorm_embedding")
    for i in range(cfg["encoder_layers"]):
        p = f"model.encoder.layers.{i}."
        if not pre_ln:
            h = ln(h + _bart_attn(sd, p + "self_attn.", h, h, cfg["encoder_attention_heads"], key_len=enc_key_len),
                   p + "self_attn_layer_norm")
            h = ln(h + linear(act(linear(h, sd, p + "fc1")), sd, p + "fc2"), p + "final_layer_norm")
        else:
            x = ln(h, p + "self_attn_layer_norm")
            h = h + _bart_attn(sd, p + "self_attn.", x, x, cfg["encoder_attention_heads"], key_len=enc_key_len)
            h = h + linear(act(linear(ln(h, p + "final_layer_norm"), sd, p + "fc1")), sd, p + "fc2")
        h = lm_adapter(adapters, i, h)
    if pre_ln:
        h = ln(h, "model.encoder.layer_norm")
    enc = h
    if trace is not None:
        trace["lm_encoder_last_hidden"] = enc

    # ---- decoder ----
    Ld = decoder_input_ids.shape[1]
    y = emb[decoder_input_ids] * scale + sd["model.decoder.embed_positions.weight"][2:2 + Ld][None]
    y = ln(y, "model.decoder.layernorm_embedding")
    nh = cfg["decoder_attention_heads"]
    for i in range(cfg["decoder_layers"]):
        p = f"model.decoder.layers.{i}."
        if not pre_ln:
            y = ln(y + _bart_attn(sd, p + "self_attn.", y, y, nh, causal=True), p + "self_attn_layer_norm")
            y = ln(y + _bart_attn(sd, p + "encoder_attn.", y, enc, nh, key_len=enc_key_len), p + "encoder_attn_layer_norm")
            y = ln(y + linear(act(linear(y, sd, p + "fc1")), sd, p + "fc2"), p + "final_layer_norm")
        else:
            x = ln(y, p + "self_attn_layer_norm")
            y = y + _bart_attn(sd, p + "self_attn.", x, x, nh, causal=True)
            y = y + _bart_attn(sd, p + "encoder_attn.", ln(y, p + "encoder_attn_layer_norm"), enc, nh, key_len=enc_key_len)
            y = y + linear(act(linear(ln(y, p + "final_layer_norm"), sd, p + "fc1")), sd, p + "fc2")
        y = lm_adapter(adapters, cfg["encoder_layers"] + i, y)     # (index: both stacks of every named LM are equally deep)
    if pre_ln:
        y = ln(y, "model.decoder.layer_norm")
    if trace is not None:
        trace["decoder_last_hidden"] = y
    head = sd.get("lm_head.weight", emb)
    logits = y @ head.t() + sd["final_logits_bias"]
    return logits, enc


def t5_relative_bucket(rel: Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> Tensor:
    """TF:models/t5/modeling_t5.py:216-263 (integer work)."""
    ret = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        ret = ret + (rel > 0).long() * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).long()
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return ret + torch.where(is_small, rel, large)


def t5_position_bias(table: Tensor, q_len: int, k_len: int, bidirectional: bool, num_buckets: int,
                     max_distance: int) -> Tensor:
    """TF:models/t5/modeling_t5.py:265-279.  table [num_buckets, H] -> [1,H,q,k]."""
    ctx = torch.arange(q_len)[:, None]
    mem = torch.arange(k_len)[None, :]
    b = t5_relative_bucket(mem - ctx, bidirectional, num_buckets, max_distance)
    return table[b].permute(2, 0, 1)[None]


def t5_forward(sd, cfg, inputs_embeds: Optional[Tensor], input_ids: Optional[Tensor],
               decoder_input_ids: Tensor, trace: Optional[dict] = None, adapters=None, enc_key_len=None):
    """T5 forward, eval mode.  TF:models/t5/modeling_t5.py:50-94 (RMSNorm, FF), 176-369 (attention:
    no QK scaling, shared bucketed relative bias from block 0), 640-752 (stack), 1044-1054 (logit
    scale d_model^-0.5 when embeddings are tied)."""
    eps = cfg.get("layer_norm_epsilon", 1e-6)
    nh, dkv = cfg["num_heads"], cfg["d_kv"]
    inner = nh * dkv
    nb, md = cfg.get("relative_attention_num_buckets", 32), cfg.get("relative_attention_max_distance", 128)
    gated = cfg.get("is_gated_act", False)
    act = act_fn(cfg.get("dense_act_fn", "relu"))
    emb = sd["shared.weight"]

    def attn(p, x, kv, bias, causal, key_len=None):
        q = x @ sd[p + "q.weight"].t()
        k = kv @ sd[p + "k.weight"].t()
        v = kv @ sd[p + "v.weight"].t()
        return mha(q, k, v, nh, 1.0, causal=causal, bias=bias, key_len=key_len) @ sd[p + "o.weight"].t()

    def ff(p, x):
        if gated:
            h = act(x @ sd[p + "wi_0.weight"].t()) * (x @ sd[p + "wi_1.weight"].t())
        else:
            h = act(x @ sd[p + "wi.weight"].t())
        return h @ sd[p + "wo.weight"].t()

    assert inner == nh * dkv
    if inputs_embeds is None:
        inputs_embeds = emb[input_ids]
    h = inputs_embeds
    S = h.shape[1]
    ebias = t5_position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"],
                             S, S, True, nb, md)
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}."
        x = rms_norm(h, sd[p + "layer.0.layer_norm.weight"], eps)
        h = h + attn(p + "layer.0.SelfAttention.", x, x, ebias, False, enc_key_len)
        h = h + ff(p + "layer.1.DenseReluDense.", rms_norm(h, sd[p + "layer.1.layer_norm.weight"], eps))
        h = lm_adapter(adapters, i, h)
    enc = rms_norm(h, sd["encoder.final_layer_norm.weight"], eps)
    if trace is not None:
        trace["lm_encoder_last_hidden"] = enc

    y = emb[decoder_input_ids]
    Ld = y.shape[1]
    dbias = t5_position_bias(sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"],
                             Ld, Ld, False, nb, md)
    for i in range(cfg.get("num_decoder_layers") or cfg["num_layers"]):
        p = f"decoder.block.{i}."
        x = rms_norm(y, sd[p + "layer.0.layer_norm.weight"], eps)
        y = y + attn(p + "layer.0.SelfAttention.", x, x, dbias, True)
        x = rms_norm(y, sd[p + "layer.1.layer_norm.weight"], eps)
        y = y + attn(p + "layer.1.EncDecAttention.", x, enc, None, False, enc_key_len)
        y = y + ff(p + "layer.2.DenseReluDense.", rms_norm(y, sd[p + "layer.2.layer_norm.weight"], eps))
        y = lm_adapter(adapters, cfg["num_layers"] + i, y)
    y = rms_norm(y, sd["decoder.final_layer_norm.weight"], eps)
    if trace is not None:
        trace["decoder_last_hidden"] = y
    if cfg.get("tie_word_embeddings", True):
        y = y * (cfg["d_model"] ** -0.5)
        head = emb
    else:
        head = sd["lm_head.weight"]
    return y @ head.t(), enc


def lm_forward(sd, cfg, inputs_embeds=None, input_ids=None, decoder_input_ids=None, trace=None, adapters=None,
               attention_mask=None):
    """attention_mask [B,S] of ones then zeros (right padding): encoder keys beyond each row's length are masked in the
    text encoder's self-attention and in the decoder's cross-attention (TF:models/bart/modeling_bart.py:741-760, 1010-1016)."""
    mt = cfg["model_type"]
    kl = None if attention_mask is None else torch.as_tensor(attention_mask).long().sum(-1)
    if mt in ("bart", "mbart"):
        return bart_like_forward(sd, cfg, inputs_embeds, input_ids, decoder_input_ids, trace, adapters, kl)
    if mt == "t5":
        return t5_forward(sd, cfg, inputs_embeds, input_ids, decoder_input_ids, trace, adapters, kl)
    raise ValueError(mt)


def greedy_labels(lm_sd, cfg, gen_input, max_length: int, adapters=None):
    """ref:train.py:18-34 `create_self_decoder_input`: greedy decoding of the LM on token ids, re-running the whole model
    for every token (the reference's loop), at most max(max_length, len(input)) steps, stopping BEFORE an eos is appended.
    -> predicted ids without the start token."""
    predicted = [cfg["decoder_start_token_id"]]
    ids = torch.tensor([list(gen_input)])
    with torch.no_grad():
        for _ in range(max(max_length, len(gen_input))):
            logits, _ = lm_forward(lm_sd, cfg, input_ids=ids, decoder_input_ids=torch.tensor([predicted]), adapters=adapters)
            nxt = int(logits.argmax(-1)[0, -1])
            if nxt == cfg["eos_token_id"]:
                break
            predicted.append(nxt)
    return predicted[1:]


def cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """nn.CrossEntropyLoss(ignore_index=-100, reduction='mean').  TF:models/bart/modeling_bart.py:942-946."""
    lp = torch.log_softmax(logits.reshape(-1, logits.shape[-1]).float(), dim=-1)
    lab = labels.reshape(-1)
    valid = lab != -100
    picked = lp[torch.arange(lab.numel()), lab.clamp(min=0)]
    return -(picked * valid).sum() / valid.sum()


def lm_token_embedding(sd, cfg, ids: Tensor) -> Tensor:
    """decoder_model.get_input_embeddings()(ids) (ref:speechmix/model.py:124, 169-170): includes
    embed_scale for BART/mBART (TF:models/bart/modeling_bart.py:101-113)."""
    if cfg["model_type"] == "t5":
        return sd["shared.weight"][ids]
    s = math.sqrt(cfg["d_model"]) if cfg.get("scale_embedding", False) else 1.0
    return sd["model.shared.weight"][ids] * s


# --------------------------------------------------------------------------------------------
# the SpeechMix glue (the hot path proper)
# --------------------------------------------------------------------------------------------
def split_state_dict(sd: Dict[str, Tensor]):
    enc = {k[len("encoder_model."):]: v for k, v in sd.items() if k.startswith("encoder_model.")}
    lm = {k[len("decoder_model."):]: v for k, v in sd.items() if k.startswith("decoder_model.")}
    rest = {k: v for k, v in sd.items() if not k.startswith(("encoder_model.", "decoder_model."))}
    return enc, lm, rest


def length_adapters(rest: Dict[str, Tensor], x: Tensor, downloop: int) -> Tensor:
    """ref:speechmix/model.py:92-96, 162: log2(down_scale) x Conv1d(d,d,k=2,s=2), no activation."""
    h = x.transpose(1, 2)
    for i in range(downloop):
        h = F.conv1d(h, rest[f"length_adapters.{i}.weight"], rest[f"length_adapters.{i}.bias"], stride=2)
    return h.transpose(1, 2)


def speechmix_eed_forward(sd: Dict[str, Tensor], enc_cfg: dict, lm_cfg: dict, input_values: Tensor,
                          labels: Optional[Tensor] = None, decoder_input_ids: Optional[Tensor] = None,
                          down_scale: int = 8, weighted_sum: bool = False, num_speech_layers: Optional[int] = None,
                          prompt_ids: Optional[Tensor] = None, trace: Optional[dict] = None,
                          spec_mask: Optional[Tensor] = None, layer_keep=None, sample_lengths=None,
                          lm_attention_mask: Optional[Tensor] = None) -> dict:
    """ref:speechmix/model.py:139-177 with HF-twin semantics where the two differ
    (weights_sum has L+1 entries, ref:speechmix/hf_model.py:268-270, 411-423).  spec_mask / layer_keep: train-mode decisions
    handed in (see speech_encoder).  sample_lengths [B]: unpadded samples per clip = an `attention_mask` into the speech
    encoder; lm_attention_mask [B,S] (right-padded ones): the mask the reference's `cal_loss` hook forwards to the LM
    (ref:speechmix/model.py:132-136)."""
    enc_sd, lm_sd, rest = split_state_dict(sd)
    B = input_values.shape[0]
    if decoder_input_ids is None and labels is None:
        decoder_input_ids = handle_decoder_input_none(lm_cfg["decoder_start_token_id"], B)
    elif decoder_input_ids is None:
        decoder_input_ids = shift_tokens_right(labels, lm_cfg["pad_token_id"], lm_cfg["decoder_start_token_id"])
    out = {}
    fl = feature_lengths(enc_cfg, sample_lengths) if sample_lengths is not None else None
    last, hidden = speech_encoder(enc_sd, enc_cfg, input_values, num_speech_layers, trace, spec_mask=spec_mask,
                                  layer_keep=layer_keep, frame_lengths=fl)
    out["encoder_last_hidden_state"] = last
    x = last
    if weighted_sum:
        w = torch.softmax(rest["weights_sum"], dim=-1)
        out["weighted_sum"] = w
        x = (w[:, None, None, None] * torch.stack(hidden, 0)).sum(0)
    out["shape_before_length_adapter"] = tuple(x.shape)
    downloop = int(math.log(down_scale, 2))
    if down_scale > 1:
        x = length_adapters(rest, x, downloop)
    out["shape_before_enc_dec_projector"] = tuple(x.shape)
    out["post_adapter"] = x
    x = x @ rest["enc_to_dec_proj.weight"].t() + rest["enc_to_dec_proj.bias"]
    out["shape_after_enc_dec_projector"] = tuple(x.shape)
    if prompt_ids is not None:
        pe = lm_token_embedding(lm_sd, lm_cfg, prompt_ids)
        x = torch.cat((pe.expand(B, -1, -1), x), 1)
    out["inputs_embeds"] = x
    adapters = {k: v for k, v in rest.items() if k.startswith("adapters.")} or None          # SpeechMixAdapter
    logits, enc_last = lm_forward(lm_sd, lm_cfg, inputs_embeds=x, decoder_input_ids=decoder_input_ids, trace=trace,
                                  adapters=adapters, attention_mask=lm_attention_mask)
    out["lm_encoder_last_hidden"] = enc_last
    out["raw_logits"] = logits
    out["logits"] = logits.argmax(-1)
    if labels is not None:
        out["loss"] = cross_entropy(logits, labels)
    return out


def speechmix_self_losses(sd, lm_cfg: dict, inputs_embeds: Tensor, text_input_ids: Tensor,
                          decoder_input_ids: Tensor, labels: Tensor) -> dict:
    """SpeechMixSelf.cal_loss, ref:speechmix/model.py:235-266: CE + KLD(batchmean) + MSE, with the
    reinterpreting `.view(B, d, -1)` of the speech hidden state reproduced exactly."""
    _, lm_sd, _ = split_state_dict(sd)
    d = lm_cfg["d_model"]
    z_s, h_s = lm_forward(lm_sd, lm_cfg, inputs_embeds=inputs_embeds, decoder_input_ids=decoder_input_ids)
    z_t, h_t = lm_forward(lm_sd, lm_cfg, input_ids=text_input_ids, decoder_input_ids=decoder_input_ids)
    B = h_t.shape[0]
    a = torch.bmm(h_t, h_s.contiguous().view(B, d, -1))
    a = torch.softmax(a / math.sqrt(d), dim=-1)
    proj = torch.bmm(a, h_s)
    mse = ((proj - h_t) ** 2).mean()
    lp_s = torch.log_softmax(z_s, -1)
    p_t = torch.softmax(z_t, -1)
    kld = (torch.xlogy(p_t, p_t) - p_t * lp_s).sum() / z_s.shape[0]  # KLDivLoss(reduction='batchmean')
    ce = cross_entropy(z_s, labels)
    return {"ce": ce, "kld": kld, "mse": mse, "loss": kld + ce + mse, "raw_logits": z_s,
            "logits": z_s.argmax(-1)}


# ---------------------------------------------------------------------------------------------------------------------
# Adafactor as the reference's Trainer runs it (ref:train.py:298 optim="adafactor"; TF:trainer.py get_optimizer_cls_and_kwargs:
# Adafactor with scale_parameter=False, relative_step=False; TF:optimization.py Adafactor.step / _approx_sq_grad / _rms).
# Restated on plain tensors: state = {"step", "row", "col"} (>= 2-D) or {"step", "v"} (1-D).  Pinned against the HF class in
# tests/test_oracle_golden.py::test_adafactor_restatement_matches_hf.
def adafactor_step(p, g, state, lr, decay_rate=-0.8, eps1=1e-30, clip_threshold=1.0):
    """In-place update of p (fp32) with gradient g; returns the update that was subtracted."""
    if not state:
        state["step"] = 0
        if g.dim() >= 2:
            state["row"] = torch.zeros(g.shape[:-1])
            state["col"] = torch.zeros(g.shape[:-2] + g.shape[-1:])
        else:
            state["v"] = torch.zeros_like(g)
    state["step"] += 1
    beta2t = 1.0 - math.pow(state["step"], decay_rate)
    u = g * g + eps1
    if g.dim() >= 2:
        state["row"].mul_(beta2t).add_(u.mean(dim=-1), alpha=1.0 - beta2t)
        state["col"].mul_(beta2t).add_(u.mean(dim=-2), alpha=1.0 - beta2t)
        r = (state["row"] / state["row"].mean(dim=-1, keepdim=True)).rsqrt().unsqueeze(-1)
        c = state["col"].unsqueeze(-2).rsqrt()
        upd = r * c * g
    else:
        state["v"].mul_(beta2t).add_(u, alpha=1.0 - beta2t)
        upd = state["v"].rsqrt() * g
    rms = upd.norm(2) / (upd.numel() ** 0.5)
    upd = upd / torch.clamp(rms / clip_threshold, min=1.0)
    upd = upd * lr
    p.sub_(upd)
    return upd
