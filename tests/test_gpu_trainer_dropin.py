"""The drop-in boundary driven by its REAL caller: `transformers.Trainer` around `speechmix_amd.SpeechMixEED`
the way ref:train.py:287-330 builds it - the reference's collator (restated: speechmix_amd.data), TrainingArguments with
optim="adafactor" / length-grouped sampling / periodic saves, the reference's FreezingCallback - for two epochs on a synthetic
ragged dataset.  What has to hold: Trainer's own loop (`model.train()`, `model(**batch)["loss"]`, `accelerator.backward`,
`clip_grad_norm_` over the `.grad` views of the flat buffer, HF's Adafactor stepping the parameter views in place,
`model.zero_grad()`) trains the HIP path - the loss falls -, a frozen epoch leaves the frozen tensors without a gradient and
unchanged, and the checkpoint Trainer writes reloads through `load_state_dict` into a fresh model that computes the same loss.

transformers 5.x (this image: 5.15) renamed what the reference's pinned `transformers>=4.12.3` called `group_by_length=True` /
`tokenizer=` (now `train_sampling_strategy="group_by_length"`, `processing_class=`) and saves non-`PreTrainedModel` modules with
safetensors, which refuses tied weights under several names: `model.tied_aliases_in_state_dict = False` (model.py) is the one
line a maintainer adds for that; everything else is the reference's call sequence.
"""
import contextlib
import io
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

ENC = dict(model_type="wav2vec2", hidden_size=64, num_hidden_layers=3, num_attention_heads=2, intermediate_size=128,
           conv_dim=[32] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2], num_conv_pos_embeddings=16,
           num_conv_pos_embedding_groups=4, feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False,
           layerdrop=0.0, mask_time_prob=0.0)
LM = dict(model_type="bart", vocab_size=120, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
          decoder_attention_heads=2, encoder_ffn_dim=128, decoder_ffn_dim=128, max_position_embeddings=128, pad_token_id=1,
          bos_token_id=0, eos_token_id=2, decoder_start_token_id=2, dropout=0.0)


def _build(cls_name="SpeechMixEED", **kw):
    import speechmix_amd
    with contextlib.redirect_stdout(io.StringIO()):
        return getattr(speechmix_amd, cls_name)(ENC, LM, down_scale=2, compute_dtype="fp32", init_seed=3, **kw)


def test_transformers_trainer_trains_saves_and_freezes_the_hip_model(tmp_path):
    transformers = pytest.importorskip("transformers")
    datasets = pytest.importorskip("datasets")
    from safetensors.torch import load_file
    from transformers import Trainer, TrainerCallback, TrainingArguments
    from speechmix_amd.data import DataCollatorWithPadding

    g = torch.Generator().manual_seed(0)
    n_items = 16
    lengths = [int(x) for x in torch.randint(4000, 9000, (n_items,), generator=g)]
    rows = dict(input_values=[(torch.randn(n, generator=g) * 0.1).tolist() for n in lengths],
                labels=[[0] + torch.randint(4, LM["vocab_size"], (int(k),), generator=g).tolist() + [2]
                        for k in torch.randint(3, 7, (n_items,), generator=g)],
                length=lengths)
    train_ds = datasets.Dataset.from_dict(rows)

    model = _build(autograd_param_inputs=True)
    model.tied_aliases_in_state_dict = False          # (safetensors-based Trainer._save: see the module docstring)
    # (Trainer saves `data_collator.tokenizer` beside the weights: the stand-in has the two ids the collator reads and that method)
    tok = types.SimpleNamespace(pad_token_id=LM["pad_token_id"], bos_token_id=LM["bos_token_id"], save_pretrained=lambda d: None)
    collator = DataCollatorWithPadding(tokenizer=tok, padding=True, selftype=False)            # ref:train.py:288-289

    class FreezingCallback(TrainerCallback):
        """ref:speechmix/module/utility.py:7-33 in this test's words: in epoch e < freeze_epoch only the LAST
        int(n / freeze_epoch * e) parameters of the speech encoder train (epoch 0: `-0:` releases all of them), afterwards
        all do; saving switches every parameter back on."""

        def __init__(self, trainer, freeze_model, freeze_epoch):
            self.trainer, self.m, self.fe = trainer, freeze_model, freeze_epoch
            self.names = [n for n, _ in freeze_model.named_parameters()]
            self.default = {n: p.requires_grad for n, p in freeze_model.named_parameters()}
            self.per = int(len(self.names) / freeze_epoch)
            self.frozen_now = []

        def on_epoch_begin(self, args, state, control, **kw):
            release = set(self.names[-int(self.per * state.epoch):]) if state.epoch < self.fe else set(self.names)
            for n, p in self.m.named_parameters():
                p.requires_grad = self.default[n] if n in release else False
            self.frozen_now = [n for n in self.names if n not in release]

        def on_save(self, args, state, control, **kw):
            for n, p in self.trainer.model.named_parameters():
                p.requires_grad = True

    class Probe(TrainerCallback):
        """Records, at every step end, which encoder tensors were frozen and whether they moved."""

        def __init__(self, model, freezer):
            self.model, self.freezer, self.snap, self.moved, self.frozen_seen = model, freezer, None, [], 0

        def on_step_begin(self, args, state, control, **kw):
            named = dict(self.model.encoder_model.named_parameters())
            self.snap = {n: named[n].detach().clone() for n in self.freezer.frozen_now[:4]}

        def on_step_end(self, args, state, control, **kw):
            named = dict(self.model.encoder_model.named_parameters())
            for n, before in self.snap.items():
                self.frozen_seen += 1
                if not torch.equal(named[n].detach(), before):
                    self.moved.append(n)

    args = TrainingArguments(output_dir=str(tmp_path / "run"), per_device_train_batch_size=4, gradient_accumulation_steps=1,
                             train_sampling_strategy="group_by_length", optim="adafactor", learning_rate=5e-3, warmup_steps=0,
                             lr_scheduler_type="constant", num_train_epochs=3, save_strategy="steps", save_steps=4, save_total_limit=2,
                             logging_steps=1, report_to="none", dataloader_num_workers=0, bf16=False, fp16=False, seed=0,
                             max_grad_norm=1.0,
                             remove_unused_columns=False)      # (5.x drops the `length` column BEFORE it builds the length-grouped sampler)
    trainer = Trainer(model=model, args=args, train_dataset=train_ds, data_collator=collator)
    freezer = FreezingCallback(trainer, model.encoder_model, 3)
    probe = Probe(model, freezer)
    trainer.add_callback(freezer)                                                                 # ref:train.py:326-328
    trainer.add_callback(probe)
    trainer.train()

    losses = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    print("Trainer losses:", [round(x, 3) for x in losses])
    assert len(losses) == 12 and all(x == x for x in losses)
    assert sum(losses[-3:]) / 3 < sum(losses[:3]) / 3 - 0.2, losses
    # epochs 1 and 2 froze the head of the encoder's parameter list: those tensors were checked and none moved
    assert probe.frozen_seen > 0 and not probe.moved, probe.moved[:4]
    # the checkpoint Trainer wrote (safetensors, tied weights once) reloads into a fresh model: same parameters, same loss
    ckpts = sorted(p for p in os.listdir(tmp_path / "run") if p.startswith("checkpoint-"))
    assert ckpts, os.listdir(tmp_path / "run")
    last = tmp_path / "run" / ckpts[-1]
    sd = load_file(str(last / "model.safetensors"))
    fresh = _build("HFSpeechMixEED").eval()
    res = fresh.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys, res
    if ckpts[-1].endswith("-12"):                      # saved after the last step: the live model and the reload agree exactly
        model.eval()
        batch = collator([train_ds[i] for i in range(4)])
        with torch.no_grad():
            a = model(batch["input_values"], labels=batch["labels"])["loss"].item()
            b = fresh(input_values=batch["input_values"], labels=batch["labels"])["loss"].item()
        assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (a, b)


def test_hf_twin_names_resolve_and_accept_the_twins_forward_keywords():
    """ref:train.py:205-222 dispatches `--HFSpeechMixEED / --HFSpeechMixFixed / --HFSpeechMixSelf / --HFSpeechMixAdapter`;
    ref:speechmix/hf_model.py:378-394 is their forward signature."""
    import speechmix_amd
    for name, base in (("HFSpeechMixEED", "SpeechMixEED"), ("HFSpeechMixFixed", "SpeechMixFixed"), ("HFSpeechMixSelf", "SpeechMixSelf"),
                       ("HFSpeechMixAdapter", "SpeechMixAdapter")):
        assert issubclass(getattr(speechmix_amd, name), getattr(speechmix_amd, base))
    m = _build("HFSpeechMixEED").eval()
    ref = _build("SpeechMixEED").eval()
    g = torch.Generator().manual_seed(1)
    wave = torch.randn(2, 6000, generator=g) * 0.1
    labels = torch.randint(4, LM["vocab_size"], (2, 5), generator=g)
    with torch.no_grad():
        a = m(input_values=wave, labels=labels, text_input_ids=None, return_model_detail=True, use_cache=False)
        b = ref(wave, labels=labels, return_model_detail=True)
    assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["raw_logits"], b["raw_logits"])
    with pytest.raises(NotImplementedError):
        m(input_values=wave, labels=labels, encoder_outputs=object())
