"""Backbone configurations for the SpeechMix hot path.

The reference loads its backbones by name (ref:speechmix/model.py:65-66: `getattr(s3prl.hub, name)()` and
`AutoModelForSeq2SeqLM.from_pretrained(name)`).  Here a backbone is described by a plain dataclass that can
be built from (a) a local HF directory containing `config.json`, (b) a HF config object / dict, or (c) a
small table of well-known names (s3prl upstream names used by the reference + the HF ids in
BASELINE.json).  Table values that could not be verified offline are marked in SURVEY.md §2.4; real
checkpoints always override them through their own `config.json`.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass, field
from typing import Optional, Tuple


@dataclass
class SpeechEncoderConfig:
    """wav2vec2 / HuBERT family (TF:models/wav2vec2/configuration_wav2vec2.py:165-219)."""
    model_type: str = "wav2vec2"
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_act: str = "gelu"
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = False
    feat_extract_norm: str = "group"
    feat_extract_activation: str = "gelu"
    do_stable_layer_norm: bool = False
    feat_proj_layer_norm: bool = True
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    hidden_dropout: float = 0.1
    attention_dropout: float = 0.1
    activation_dropout: float = 0.1
    feat_proj_dropout: float = 0.0
    layerdrop: float = 0.1
    apply_spec_augment: bool = True
    mask_time_prob: float = 0.05
    mask_time_length: int = 10
    mask_time_min_masks: int = 2
    mask_feature_prob: float = 0.0
    initializer_range: float = 0.02

    def to_dict(self):
        d = asdict(self)
        for k in ("conv_dim", "conv_kernel", "conv_stride"):
            d[k] = list(d[k])
        return d

    def frames(self, n_samples: int) -> int:
        """TF:models/wav2vec2/modeling_wav2vec2.py:997-1036."""
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n_samples - k) // s + 1
            n_samples = n
        return n_samples


@dataclass
class LMConfig:
    """BART / mBART / T5 seq2seq LM (TF:models/{bart,mbart,t5}/configuration_*.py)."""
    model_type: str = "bart"
    vocab_size: int = 50265
    d_model: int = 768
    encoder_layers: int = 6
    decoder_layers: int = 6
    encoder_attention_heads: int = 12
    decoder_attention_heads: int = 12
    encoder_ffn_dim: int = 3072
    decoder_ffn_dim: int = 3072
    activation_function: str = "gelu"
    max_position_embeddings: int = 1024
    scale_embedding: bool = False
    dropout: float = 0.1
    attention_dropout: float = 0.0
    activation_dropout: float = 0.0
    encoder_layerdrop: float = 0.0
    decoder_layerdrop: float = 0.0
    pad_token_id: Optional[int] = 1
    bos_token_id: Optional[int] = 0
    eos_token_id: Optional[int] = 2
    decoder_start_token_id: Optional[int] = 2
    init_std: float = 0.02
    max_length: int = 20
    # T5 only
    d_kv: int = 64
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    is_gated_act: bool = False
    tie_word_embeddings: bool = True

    @property
    def hidden_size(self):          # ref:speechmix/model.py:102 reads decoder_model.config.hidden_size
        return self.d_model

    def to_dict(self):
        d = asdict(self)
        if self.model_type == "t5":
            d.update(num_layers=self.encoder_layers, num_decoder_layers=self.decoder_layers,
                     num_heads=self.encoder_attention_heads, d_ff=self.encoder_ffn_dim,
                     dense_act_fn=self.activation_function)
        return d


_SPEECH_PRESETS = {
    # s3prl upstream names used by the reference (ref:README.md:32-50) and their HF twins
    "wav2vec2": dict(),
    "wav2vec2_base_960": dict(),
    "facebook/wav2vec2-base": dict(),
    "wav2vec2_large_960": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096),
    "facebook/wav2vec2-large": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                    intermediate_size=4096),
    "wav2vec2_large_ll60k": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                                 feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True),
    "facebook/wav2vec2-large-lv60": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                         intermediate_size=4096, feat_extract_norm="layer", conv_bias=True,
                                         do_stable_layer_norm=True),
    "hubert": dict(model_type="hubert"),
    "hubert_base": dict(model_type="hubert"),
    "facebook/hubert-base-ls960": dict(model_type="hubert"),
    "hubert_large_ll60k": dict(model_type="hubert", hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                               intermediate_size=4096, feat_extract_norm="layer", conv_bias=True,
                               do_stable_layer_norm=True),
    "facebook/hubert-large-ll60k": dict(model_type="hubert", hidden_size=1024, num_hidden_layers=24,
                                        num_attention_heads=16, intermediate_size=4096, feat_extract_norm="layer",
                                        conv_bias=True, do_stable_layer_norm=True),
}

_LM_PRESETS = {
    "facebook/bart-base": dict(),
    "facebook/bart-large": dict(d_model=1024, encoder_layers=12, decoder_layers=12, encoder_attention_heads=16,
                                decoder_attention_heads=16, encoder_ffn_dim=4096, decoder_ffn_dim=4096),
    "facebook/mbart-large-50": dict(model_type="mbart", vocab_size=250054, d_model=1024, encoder_layers=12,
                                    decoder_layers=12, encoder_attention_heads=16, decoder_attention_heads=16,
                                    encoder_ffn_dim=4096, decoder_ffn_dim=4096, scale_embedding=True,
                                    activation_function="relu"),
    "t5-small": dict(model_type="t5", vocab_size=32128, d_model=512, encoder_layers=6, decoder_layers=6,
                     encoder_attention_heads=8, decoder_attention_heads=8, encoder_ffn_dim=2048, decoder_ffn_dim=2048,
                     activation_function="relu", pad_token_id=0, bos_token_id=None, eos_token_id=1,
                     decoder_start_token_id=0),
    "t5-large": dict(model_type="t5", vocab_size=32128, d_model=1024, encoder_layers=24, decoder_layers=24,
                     encoder_attention_heads=16, decoder_attention_heads=16, encoder_ffn_dim=4096, decoder_ffn_dim=4096,
                     activation_function="relu", pad_token_id=0, bos_token_id=None, eos_token_id=1,
                     decoder_start_token_id=0),
}


def _from_hf_dict(d: dict, kind: str):
    if kind == "speech":
        fields = SpeechEncoderConfig.__dataclass_fields__
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in fields}
        mt = d.get("model_type", "wav2vec2")
        if mt not in ("wav2vec2", "hubert"):
            raise ValueError(f"unsupported speech encoder model_type {mt!r} (wav2vec2 / hubert family only)")
        if d.get("conv_pos_batch_norm", False):
            raise ValueError("conv_pos_batch_norm=True is not supported")
        return SpeechEncoderConfig(**kw)
    mt = d.get("model_type", "bart")
    if mt not in ("bart", "mbart", "t5"):
        raise ValueError(f"unsupported LM model_type {mt!r} (bart / mbart / t5)")
    fields = LMConfig.__dataclass_fields__
    kw = {k: v for k, v in d.items() if k in fields}
    if mt == "t5":
        kw.update(encoder_layers=d["num_layers"], decoder_layers=d.get("num_decoder_layers") or d["num_layers"],
                  encoder_attention_heads=d["num_heads"], decoder_attention_heads=d["num_heads"],
                  encoder_ffn_dim=d["d_ff"], decoder_ffn_dim=d["d_ff"],
                  activation_function=d.get("dense_act_fn", "relu"), is_gated_act=d.get("is_gated_act", False),
                  tie_word_embeddings=d.get("tie_word_embeddings", True), scale_embedding=False,
                  dropout=d.get("dropout_rate", 0.1), attention_dropout=d.get("dropout_rate", 0.1),
                  activation_dropout=d.get("dropout_rate", 0.1))
    return LMConfig(**kw)


def _load(spec, kind, presets, cls):
    if isinstance(spec, cls):
        return spec, None
    if isinstance(spec, dict):
        return _from_hf_dict(spec, kind), None
    if hasattr(spec, "to_dict") and not isinstance(spec, str):
        return _from_hf_dict(spec.to_dict(), kind), None
    if isinstance(spec, str):
        if os.path.isdir(spec):
            with open(os.path.join(spec, "config.json")) as f:
                return _from_hf_dict(json.load(f), kind), spec
        if spec in presets:
            return cls(**presets[spec]), None
        raise ValueError(f"unknown {kind} model {spec!r}: pass a local HF directory, a config object/dict, or one of "
                         f"{sorted(presets)}")
    raise TypeError(type(spec))


def load_speech_config(spec):
    """-> (SpeechEncoderConfig, checkpoint_dir or None)"""
    return _load(spec, "speech", _SPEECH_PRESETS, SpeechEncoderConfig)


def load_lm_config(spec):
    """-> (LMConfig, checkpoint_dir or None)"""
    return _load(spec, "lm", _LM_PRESETS, LMConfig)
