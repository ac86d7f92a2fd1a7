"""Checkpoint formats of the reference's backbones, loaded by key name into the parameter trees.

* HuggingFace directories: `model.safetensors` or `pytorch_model.bin` (+ sharded `*.index.json` forms), the
  latter being what `save_pretrained` wrote before safetensors and what the reference's eval script loads for the
  whole model (ref:eval.py:10: `spm.load_state_dict(torch.load('./pytorch_model.bin'))`).
* fairseq / s3prl speech encoders: the reference's default speech side is `s3prl.hub.wav2vec2()` /
  `hubert_large_ll60k()` (ref:speechmix/model.py:65), whose `.model` is a fairseq `Wav2Vec2Model` / `HubertModel`.
  Its parameter names are mapped onto the HF names used here with the table of HF's own conversion script
  (TF:models/wav2vec2/convert_wav2vec2_original_pytorch_checkpoint_to_pytorch.py MAPPING / load_conv_layer;
  the HuBERT script uses the same table).  A whole-model state dict saved from ref:speechmix/model.py carries those
  names behind `encoder_model.model.`; `convert_speechmix_state_dict` rewrites them.

Host-side plumbing only: nothing here touches the GPU.
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, Optional

import torch

# fairseq module path (inside the wav2vec2 / HuBERT model) -> HF module path; "*" = the layer index
_FAIRSEQ_TO_HF = [
    ("post_extract_proj", "feature_projection.projection"),
    ("encoder.pos_conv.0", "encoder.pos_conv_embed.conv"),
    ("encoder.layers.*.self_attn.k_proj", "encoder.layers.*.attention.k_proj"),
    ("encoder.layers.*.self_attn.v_proj", "encoder.layers.*.attention.v_proj"),
    ("encoder.layers.*.self_attn.q_proj", "encoder.layers.*.attention.q_proj"),
    ("encoder.layers.*.self_attn.out_proj", "encoder.layers.*.attention.out_proj"),
    ("encoder.layers.*.self_attn_layer_norm", "encoder.layers.*.layer_norm"),
    ("encoder.layers.*.fc1", "encoder.layers.*.feed_forward.intermediate_dense"),
    ("encoder.layers.*.fc2", "encoder.layers.*.feed_forward.output_dense"),
    ("encoder.layers.*.final_layer_norm", "encoder.layers.*.final_layer_norm"),
    ("encoder.layer_norm", "encoder.layer_norm"),
    ("layer_norm", "feature_projection.layer_norm"),
    ("mask_emb", "masked_spec_embed"),
]
_FAIRSEQ_PREFIXES = ("w2v_encoder.w2v_model.", "w2v_model.", "model.")
_HF_PREFIXES = ("wav2vec2.", "hubert.")
_WN = {"weight_g": "parametrizations.weight.original0", "weight_v": "parametrizations.weight.original1"}


def _wn(key: str) -> str:
    """Old-style weight-norm names (`conv.weight_g/_v`) -> parametrization names used by current torch / HF."""
    for old, new in _WN.items():
        if key.endswith("pos_conv_embed.conv." + old):
            return key[: -len(old)] + new
    return key


def hf_key_from_fairseq(name: str) -> Optional[str]:
    """fairseq wav2vec2 / HuBERT parameter name -> HF `Wav2Vec2Model` / `HubertModel` name (None: not a parameter of the
    encoder that SpeechMix uses - quantizer, project_q, final_proj, label embeddings ...)."""
    for pre in _FAIRSEQ_PREFIXES:
        if name.startswith(pre):
            name = name[len(pre):]
    m = re.match(r"feature_extractor\.conv_layers\.(\d+)\.(\d+)\.(?:\d+\.)?(weight|bias)$", name)
    if m:       # .{i}.0 = conv, .{i}.2[.1] = GroupNorm (layer 0) / LayerNorm between transposes ("layer" mode)
        i, kind, leaf = m.group(1), int(m.group(2)), m.group(3)
        if kind == 0:
            return f"feature_extractor.conv_layers.{i}.conv.{leaf}"
        if kind == 2:
            return f"feature_extractor.conv_layers.{i}.layer_norm.{leaf}"
        return None
    if name == "mask_emb":
        return "masked_spec_embed"
    for src, dst in _FAIRSEQ_TO_HF:
        pat = "^" + re.escape(src).replace(r"\*", r"(\d+)") + r"\.(weight|bias|weight_g|weight_v)$"
        m = re.match(pat, name)
        if m:
            groups = m.groups()
            leaf = groups[-1]
            out = dst.replace("*", groups[0]) if "*" in dst else dst
            return _wn(f"{out}.{leaf}")
    return None


def normalise_hf_key(key: str) -> str:
    """HF names as stored by task-head checkpoints (`wav2vec2.` / `hubert.` prefix) and by older torch (weight_g / _v)."""
    for pre in _HF_PREFIXES:
        if key.startswith(pre):
            key = key[len(pre):]
    return _wn(key)


def read_state_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    import pickle
    try:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as first:
        # (only an unsupported global is retried: a truncated / unreadable file raises something else and is reported as such)
        # real fairseq .pt files keep an argparse.Namespace (`args`) next to the weights: allow exactly that class, nothing else
        import argparse
        if not hasattr(torch.serialization, "safe_globals"):
            raise RuntimeError(f"{path}: holds objects beyond tensors and this torch has no torch.serialization.safe_globals; "
                               f"convert it first, e.g. torch.save({{'model': ckpt['model']}}, ...) in a trusted environment") from first
        try:
            with torch.serialization.safe_globals([argparse.Namespace]):
                sd = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError:
            raise RuntimeError(f"{path}: not loadable with weights_only=True (objects beyond tensors / argparse.Namespace inside); "
                               f"convert it first, e.g. torch.save({{'model': ckpt['model']}}, ...) in a trusted environment") from first
    # where the weights sit: fairseq nests them under "model" (next to cfg / args / task_state), s3prl-converted upstream
    # checkpoints (what the reference's s3prl.hub.wav2vec2() / hubert_large_ll60k() download) under "model_weight" (next to
    # task_cfg / model_cfg), lightning-style files under "state_dict"
    for key in ("model", "model_weight", "state_dict"):
        if isinstance(sd, dict) and isinstance(sd.get(key), dict):
            inner = {k: v for k, v in sd[key].items() if torch.is_tensor(v)}     # (fairseq keeps non-tensor entries such as `_ema` there)
            if inner:
                sd = inner
                break
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """A checkpoint directory (safetensors or .bin, single file or sharded) or a single weights file."""
    if os.path.isfile(path):
        return read_state_file(path)
    for single in ("model.safetensors", "pytorch_model.bin"):
        f = os.path.join(path, single)
        if os.path.exists(f):
            return read_state_file(f)
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        f = os.path.join(path, index)
        if os.path.exists(f):
            with open(f) as fh:
                files = sorted(set(json.load(fh)["weight_map"].values()))
            out: Dict[str, torch.Tensor] = {}
            for shard in files:
                out.update(read_state_file(os.path.join(path, shard)))
            return out
    raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin (or their sharded indexes) in this directory")


def load_backbone(tree: torch.nn.Module, ckpt: str) -> Dict[str, list]:
    """Load a backbone checkpoint into a parameter tree by name.  Speech encoders may use HF or fairseq / s3prl
    names; LMs use HF names.  -> {"loaded": [...], "missing": [...], "unexpected": [...]}."""
    sd = read_checkpoint(ckpt)
    own = dict(tree.state_dict())
    fixed, unexpected = {}, []
    for k, v in sd.items():
        cands = [normalise_hf_key(k)]
        fk = hf_key_from_fairseq(k)
        if fk is not None:
            cands.append(fk)
        for c in cands:
            if c in own and tuple(own[c].shape) == tuple(v.shape):
                fixed[c] = v
                break
        else:
            unexpected.append(k)
    if not fixed:
        raise RuntimeError(f"{ckpt}: none of its {len(sd)} tensors matches a parameter of {type(tree).__name__} by name and shape "
                           f"(first keys: {sorted(sd)[:3]}) - refusing to train from random initialisation silently")
    res = tree.load_state_dict(fixed, strict=False)
    missing = list(res.missing_keys)
    if len(missing) > len(own) // 2:
        import warnings
        warnings.warn(f"{ckpt}: only {len(fixed)} of {len(own)} tensors of {type(tree).__name__} were found "
                      f"(missing e.g. {missing[:3]}); the rest keep their initialisation")
    return {"loaded": sorted(fixed), "missing": missing, "unexpected": unexpected}


def convert_speechmix_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Whole-model state dict of the reference -> this package's names.  ref:speechmix/hf_model.py (HF twin) already
    uses HF names; ref:speechmix/model.py keeps the s3prl upstream under `encoder_model.model.` with fairseq names."""
    out = {}
    for k, v in sd.items():
        if k.startswith("encoder_model.model."):
            inner = k[len("encoder_model.model."):]
            hk = hf_key_from_fairseq(inner)
            if hk is None and inner.split(".")[0] in ("feature_extractor", "feature_projection", "encoder", "masked_spec_embed"):
                hk = normalise_hf_key(inner)          # an HF model wrapped the same way (tests / stand-ins)
            if hk is not None:
                out["encoder_model." + hk] = v
            continue
        if k.startswith("encoder_model."):
            out["encoder_model." + normalise_hf_key(k[len("encoder_model."):])] = v
            continue
        out[k] = v
    return out
