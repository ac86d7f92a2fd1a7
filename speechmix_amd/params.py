"""Parameter ownership for the MI355X SpeechMix step.

* `spec_*` functions list every parameter of a backbone with its HuggingFace state-dict name and shape
  (so `state_dict()` / `load_state_dict()` interoperate with HF checkpoints and with the reference's
  `pytorch_model.bin`, ref:eval.py:10; SURVEY.md §5 "Checkpoint / resume").
* `ParamTree` is an `nn.Module` that only *owns* parameters under those names (no forward of its own:
  the arithmetic is the HIP engine's).  `encoder_model.named_parameters()` therefore works for the
  reference's FreezingCallback (ref:speechmix/module/utility.py:14-29).
* `FlatStore` lays all parameters out in ONE fp32 device buffer (q/k/v projections adjacent so the fused
  QKV GEMM reads one [3d, d] operand), with a same-offset bf16 compute copy and a same-offset fp32 gradient
  buffer: the RCCL all-reduce, gradient clipping and the optimizer then run on flat ranges.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
from torch import nn

from .configs import LMConfig, SpeechEncoderConfig

Shape = Tuple[int, ...]


# ------------------------------------------------------------------------------------------------
# parameter specs (HF names)
# ------------------------------------------------------------------------------------------------
def spec_speech_encoder(c: SpeechEncoderConfig, num_layers: int) -> "OrderedDict[str, Shape]":
    """TF:models/wav2vec2/modeling_wav2vec2.py (Wav2Vec2Model) / TF:models/hubert/modeling_hubert.py."""
    s: "OrderedDict[str, Shape]" = OrderedDict()
    d = c.hidden_size
    if c.model_type == "wav2vec2" or c.mask_time_prob > 0:
        s["masked_spec_embed"] = (d,)
    cin = 1
    for i, (co, k) in enumerate(zip(c.conv_dim, c.conv_kernel)):
        p = f"feature_extractor.conv_layers.{i}."
        s[p + "conv.weight"] = (co, cin, k)
        if c.conv_bias:
            s[p + "conv.bias"] = (co,)
        if c.feat_extract_norm == "layer" or (c.feat_extract_norm == "group" and i == 0):
            s[p + "layer_norm.weight"] = (co,)
            s[p + "layer_norm.bias"] = (co,)
        cin = co
    if c.feat_proj_layer_norm:
        s["feature_projection.layer_norm.weight"] = (cin,)
        s["feature_projection.layer_norm.bias"] = (cin,)
    s["feature_projection.projection.weight"] = (d, cin)
    s["feature_projection.projection.bias"] = (d,)
    K, G = c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups
    s["encoder.pos_conv_embed.conv.bias"] = (d,)
    s["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, K)
    s["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (d, d // G, K)
    s["encoder.layer_norm.weight"] = (d,)
    s["encoder.layer_norm.bias"] = (d,)
    for i in range(num_layers):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"attention.{n}.weight"] = (d, d)
            s[p + f"attention.{n}.bias"] = (d,)
        s[p + "layer_norm.weight"] = (d,)
        s[p + "layer_norm.bias"] = (d,)
        s[p + "feed_forward.intermediate_dense.weight"] = (c.intermediate_size, d)
        s[p + "feed_forward.intermediate_dense.bias"] = (c.intermediate_size,)
        s[p + "feed_forward.output_dense.weight"] = (d, c.intermediate_size)
        s[p + "feed_forward.output_dense.bias"] = (d,)
        s[p + "final_layer_norm.weight"] = (d,)
        s[p + "final_layer_norm.bias"] = (d,)
    return s


def spec_lm(c: LMConfig) -> Tuple["OrderedDict[str, Shape]", Dict[str, str], Dict[str, Shape]]:
    """-> (params, tied aliases {alias: canonical}, buffers).  TF:models/bart/modeling_bart.py:801-840,
    TF:models/mbart/modeling_mbart.py:763-800, TF:models/t5/modeling_t5.py:935-960."""
    s: "OrderedDict[str, Shape]" = OrderedDict()
    alias: Dict[str, str] = {}
    buffers: Dict[str, Shape] = {}
    d, V = c.d_model, c.vocab_size
    if c.model_type in ("bart", "mbart"):
        s["model.shared.weight"] = (V, d)
        alias["model.encoder.embed_tokens.weight"] = "model.shared.weight"
        alias["model.decoder.embed_tokens.weight"] = "model.shared.weight"
        alias["lm_head.weight"] = "model.shared.weight"
        buffers["final_logits_bias"] = (1, V)
        for side, nl, ffn in (("encoder", c.encoder_layers, c.encoder_ffn_dim), ("decoder", c.decoder_layers, c.decoder_ffn_dim)):
            pre = f"model.{side}."
            s[pre + "embed_positions.weight"] = (c.max_position_embeddings + 2, d)
            s[pre + "layernorm_embedding.weight"] = (d,)
            s[pre + "layernorm_embedding.bias"] = (d,)
            if c.model_type == "mbart":
                s[pre + "layer_norm.weight"] = (d,)
                s[pre + "layer_norm.bias"] = (d,)
            for i in range(nl):
                p = f"{pre}layers.{i}."
                attns = ["self_attn"] + (["encoder_attn"] if side == "decoder" else [])
                for a in attns:
                    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                        s[p + f"{a}.{n}.weight"] = (d, d)
                        s[p + f"{a}.{n}.bias"] = (d,)
                    s[p + f"{a}_layer_norm.weight"] = (d,)
                    s[p + f"{a}_layer_norm.bias"] = (d,)
                s[p + "fc1.weight"] = (ffn, d)
                s[p + "fc1.bias"] = (ffn,)
                s[p + "fc2.weight"] = (d, ffn)
                s[p + "fc2.bias"] = (d,)
                s[p + "final_layer_norm.weight"] = (d,)
                s[p + "final_layer_norm.bias"] = (d,)
    elif c.model_type == "t5":
        inner = c.encoder_attention_heads * c.d_kv
        s["shared.weight"] = (V, d)
        alias["encoder.embed_tokens.weight"] = "shared.weight"
        alias["decoder.embed_tokens.weight"] = "shared.weight"
        if c.tie_word_embeddings:
            alias["lm_head.weight"] = "shared.weight"
        else:
            s["lm_head.weight"] = (V, d)
        for side, nl in (("encoder", c.encoder_layers), ("decoder", c.decoder_layers)):
            for i in range(nl):
                p = f"{side}.block.{i}.layer."
                subs = [("0.SelfAttention", True)] + ([("1.EncDecAttention", False)] if side == "decoder" else [])
                for name, _ in subs:
                    for n in ("q", "k", "v"):
                        s[p + f"{name}.{n}.weight"] = (inner, d)
                    s[p + f"{name}.o.weight"] = (d, inner)
                    if name.endswith("SelfAttention") and i == 0:
                        s[p + f"{name}.relative_attention_bias.weight"] = (c.relative_attention_num_buckets,
                                                                          c.encoder_attention_heads)
                    s[p + f"{name.split('.')[0]}.layer_norm.weight"] = (d,)
                ff = "2" if side == "decoder" else "1"
                if c.is_gated_act:
                    s[p + f"{ff}.DenseReluDense.wi_0.weight"] = (c.encoder_ffn_dim, d)
                    s[p + f"{ff}.DenseReluDense.wi_1.weight"] = (c.encoder_ffn_dim, d)
                else:
                    s[p + f"{ff}.DenseReluDense.wi.weight"] = (c.encoder_ffn_dim, d)
                s[p + f"{ff}.DenseReluDense.wo.weight"] = (d, c.encoder_ffn_dim)
                s[p + f"{ff}.layer_norm.weight"] = (d,)
            s[f"{side}.final_layer_norm.weight"] = (d,)
    else:
        raise ValueError(c.model_type)
    return s, alias, buffers


# ------------------------------------------------------------------------------------------------
# module tree that owns the parameters
# ------------------------------------------------------------------------------------------------
class ParamTree(nn.Module):
    """Nested parameter container addressed by dotted HF names."""

    def add(self, dotted: str, value, buffer=False):
        head, _, rest = dotted.partition(".")
        if not rest:
            if buffer:
                self.register_buffer(head, value)
            else:
                self.register_parameter(head, value)
            return
        if head not in self._modules:
            self.add_module(head, ParamTree())
        self._modules[head].add(rest, value, buffer)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("ParamTree only owns parameters; the forward pass is the HIP engine's")


def build_tree(spec: "OrderedDict[str, Shape]", alias: Dict[str, str] = None, buffers: Dict[str, Shape] = None,
               device="cpu", tree: ParamTree = None) -> ParamTree:
    tree = tree if tree is not None else ParamTree()
    made = {}
    for name, shape in spec.items():
        p = nn.Parameter(torch.empty(shape, dtype=torch.float32, device=device))
        made[name] = p
        tree.add(name, p)
    for a, canon in (alias or {}).items():
        tree.add(a, made[canon])
    for name, shape in (buffers or {}).items():
        tree.add(name, torch.zeros(shape, dtype=torch.float32, device=device), buffer=True)
    return tree


# ------------------------------------------------------------------------------------------------
# random init following HF's rules (exact values never matter for parity: tests load fixtures)
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def init_speech_encoder(tree: nn.Module, c: SpeechEncoderConfig, gen: torch.Generator):
    """TF:models/wav2vec2/modeling_wav2vec2.py:966-995 (_init_weights)."""
    for name, p in tree.named_parameters():
        cpu = torch.empty(p.shape, dtype=torch.float32)
        if name == "masked_spec_embed":
            cpu.uniform_(generator=gen)
        elif "feature_extractor" in name and name.endswith("conv.weight"):
            fan_in = p.shape[1] * p.shape[2]
            cpu.normal_(0, math.sqrt(2.0 / fan_in), generator=gen)          # kaiming_normal_
        elif name.endswith("original1"):
            cpu.normal_(0, 2 * math.sqrt(1.0 / (p.shape[2] * p.shape[0])), generator=gen)
        elif name.endswith("original0"):
            cpu.fill_(1.0)
        elif "feature_projection.projection.weight" in name:
            k = math.sqrt(1.0 / p.shape[1])
            cpu.uniform_(-k, k, generator=gen)
        elif name.endswith("layer_norm.weight") or name.endswith("final_layer_norm.weight"):
            cpu.fill_(1.0)
        elif name.endswith(".bias"):
            cpu.zero_()
        elif name.endswith(".weight"):
            cpu.normal_(0, c.initializer_range, generator=gen)
        else:
            cpu.zero_()
        p.copy_(cpu)
    # weight_norm g starts at ||v|| so that the effective weight equals v
    sd = dict(tree.named_parameters())
    v = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"].copy_(
        torch.sqrt((v.float() ** 2).sum(dim=(0, 1), keepdim=True)))


@torch.no_grad()
def init_lm(tree: nn.Module, c: LMConfig, gen: torch.Generator):
    """TF:models/bart/modeling_bart.py (BartPreTrainedModel._init_weights: normal(0, init_std), LN 1/0)."""
    for name, p in tree.named_parameters():
        cpu = torch.empty(p.shape, dtype=torch.float32)
        if "layer_norm" in name or "layernorm" in name:
            cpu.fill_(1.0) if name.endswith("weight") else cpu.zero_()
        elif name.endswith(".bias"):
            cpu.zero_()
        else:
            std = c.init_std if c.model_type != "t5" else (1.0 if "shared" in name else p.shape[-1] ** -0.5)
            cpu.normal_(0, std, generator=gen)
            if name.endswith("shared.weight") and c.model_type != "t5" and c.pad_token_id is not None:
                cpu[c.pad_token_id].zero_()
        p.copy_(cpu)


# ------------------------------------------------------------------------------------------------
# flat storage
# ------------------------------------------------------------------------------------------------
_QKV_SETS = (("q_proj", "k_proj", "v_proj"), ("q", "k", "v"))


def _alloc_order(names: List[str]) -> List[str]:
    """q/k/v weights (and biases) of one attention module are made adjacent, in q,k,v order."""
    done, out = set(), []
    nset = set(names)
    for n in names:
        if n in done:
            continue
        placed = False
        for trio in _QKV_SETS:
            for suffix in (".weight", ".bias"):
                for t in trio:
                    tag = f".{t}{suffix}"
                    if n.endswith(tag):
                        base = n[: -len(tag)]
                        group = [f"{base}.{x}{suffix}" for x in trio]
                        if all(g in nset for g in group):
                            for g in group:
                                if g not in done:
                                    out.append(g)
                                    done.add(g)
                            placed = True
                        break
                if placed:
                    break
            if placed:
                break
        if not placed:
            out.append(n)
            done.add(n)
    return out


class FlatStore:
    ALIGN = 64  # elements: 256-B aligned fp32 / 128-B aligned bf16 ranges

    def __init__(self, module: nn.Module, device, compute_dtype: torch.dtype):
        self.module = module
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        named = OrderedDict((n, p) for n, p in module.named_parameters())  # de-duplicated (tied weights once)
        order = _alloc_order(list(named.keys()))
        self.offsets: Dict[str, Tuple[int, int, Shape]] = {}
        off = 0
        for n in order:
            p = named[n]
            numel = p.numel()
            self.offsets[n] = (off, numel, tuple(p.shape))
            off += numel
            # keep q|k|v contiguous: only pad when the next tensor does not continue a q/k/v trio
            if not self._continues_trio(n):
                off = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.master = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.shadow = self.master if compute_dtype == torch.float32 else torch.zeros(
            self.total, dtype=compute_dtype, device=self.device)
        self._fresh = False
        # versions of the compute copies (engine.py keeps transposed copies of the Linear weights for the data gradients): `wver` moves
        # whenever the trainable tensors' copies change (optimizer steps), `hard_ver` when every tensor's may have (a re-cast of the masters)
        self.wver = 0
        self.hard_ver = 0
        # True: parameters may be updated behind our back (torch optimizers, load_state_dict) -> the bf16
        # copies are re-cast at every forward.  The built-in flat optimizer keeps them fresh itself.
        self.external_updates = True
        self.rebind(named)

    @staticmethod
    def _continues_trio(name: str) -> bool:
        for trio in _QKV_SETS:
            for suffix in (".weight", ".bias"):
                for t in trio[:2]:
                    if name.endswith(f".{t}{suffix}"):
                        return True
        return False

    @torch.no_grad()
    def rebind(self, named=None):
        """Point every nn.Parameter at its slice of the flat master buffer (keeping current values)."""
        named = named or OrderedDict((n, p) for n, p in self.module.named_parameters())
        self.params = named
        for n, p in named.items():
            off, numel, shape = self.offsets[n]
            dst = self.master[off:off + numel].view(shape)
            if p.data.data_ptr() != dst.data_ptr():
                dst.copy_(p.data.to(self.device, torch.float32))
                p.data = dst
        self._fresh = False

    # ---- views -----------------------------------------------------------------------------
    def p32(self, name):        # fp32 master view
        off, numel, shape = self.offsets[name]
        return self.master[off:off + numel].view(shape)

    def w(self, name):          # compute-dtype view (bf16 copy, or the master itself on the fp32 path)
        off, numel, shape = self.offsets[name]
        return self.shadow[off:off + numel].view(shape)

    def g(self, name):          # fp32 gradient view
        off, numel, shape = self.offsets[name]
        return self.grad[off:off + numel].view(shape)

    def cat(self, names, which="w"):
        """One view over adjacent tensors (fused q|k|v operand)."""
        offs = [self.offsets[n] for n in names]
        for (o1, n1, _), (o2, _, _) in zip(offs[:-1], offs[1:]):
            if o1 + n1 != o2:
                raise RuntimeError(f"{names} are not adjacent in the flat store")
        start = offs[0][0]
        total = sum(o[1] for o in offs)
        buf = {"w": self.shadow, "p32": self.master, "g": self.grad}[which]
        inner = offs[0][2][1:]
        return buf[start:start + total].view((-1,) + tuple(inner))

    def requires_grad(self, name):
        return self.params[name].requires_grad

    def has_prefix(self, prefix):
        return any(n.startswith(prefix) for n in self.offsets)

    def refresh_shadow(self, force=False):
        """bf16 compute copies follow the fp32 masters (no-op when nothing changed / fp32 path)."""
        if self.shadow is self.master:
            return False
        if self._fresh and not self.external_updates and not force:
            return False
        from . import ops
        ops.cast_from_f32(self.master, self.shadow, self.total, ops.BF16)
        self._fresh = True
        self.wver += 1
        self.hard_ver += 1
        return True

    def mark_shadow_fresh(self):
        """The fused optimizer has written masters and compute copies together."""
        self._fresh = True
        self.wver += 1

    def name_at(self, elem_off):
        """Name of the tensor that holds element `elem_off` of the flat buffers (None: padding / outside)."""
        import bisect
        if not hasattr(self, "_starts"):
            items = sorted((o, n, name) for name, (o, n, _) in self.offsets.items())
            self._starts = [o for o, _, _ in items]
            self._spans = items
        i = bisect.bisect_right(self._starts, elem_off) - 1
        if i < 0:
            return None
        o, n, name = self._spans[i]
        return name if o <= elem_off < o + n else None

    def invalidate(self):
        self._fresh = False

    def publish_grads(self):
        """Expose flat gradient slices as `.grad` of every trainable parameter (PyTorch optimizers / DDP)."""
        for n, p in self.params.items():
            if p.requires_grad:
                p.grad = self.g(n)
            else:
                p.grad = None
