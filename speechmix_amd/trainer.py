"""Native training-step driver: forward + backward + gradient all-reduce + clipping + optimizer update.

What HF `Trainer.training_step` + accelerate/DDP + the optimizer do around the reference model
(TF:trainer.py:1892-1963; clip to max_grad_norm, optimizer step, zero_grad), done here
without autograd and without per-tensor launches: the engine writes gradients into the flat buffer,
`dist.GradReducer` all-reduces stage ranges on a side stream while backward continues, and the update is
one fused launch per contiguous trainable range (which also refreshes the bf16 compute copies).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import os

import torch
import torch.distributed as dist

from . import ops
from .dist import GradReducer, stage_ranges


def trainable_ranges(store, skip_layers=(), layer_of=None) -> List[Tuple[int, int]]:
    """Maximal runs of consecutive parameters that are updated this step in the flat buffer (alignment gaps are merged):
    trainable, and not belonging to a speech-encoder layer in `skip_layers` (LayerDrop left it without a gradient)."""
    items = sorted((o, o + n, store.requires_grad(name) and not (layer_of and layer_of.get(name, -1) in skip_layers))
                   for name, (o, n, _) in store.offsets.items())
    out: List[List[int]] = []
    prev_trainable = False
    for a, b, tr in items:
        if tr:
            if out and prev_trainable:
                out[-1][1] = b
            else:
                out.append([a, b])
        prev_trainable = tr
    return [(a, b) for a, b in out]


def linear_schedule_with_warmup(lr, warmup_steps, total_steps):
    """HF Trainer's default schedule, which the reference trains with (ref:train.py:305-306: learning_rate 5e-4,
    warmup_steps 500; TF:optimization.py get_linear_schedule_with_warmup): step (1-based) -> learning rate."""
    def f(step):
        s = step - 1
        if s < warmup_steps:
            return lr * s / max(1, warmup_steps)
        return lr * max(0.0, (total_steps - s) / max(1, total_steps - warmup_steps))
    return f


class StepRunner:
    def __init__(self, model, lr=4e-5, optimizer="adamw", betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_grad_norm=1.0, momentum=0.0, force_comm=False, grad_accum=1, sync_params=True, seed=None):
        """lr: a float, or a callable step -> lr (`linear_schedule_with_warmup`).
        grad_accum: micro-batches per optimizer step (the reference trains with HF Trainer's
        gradient_accumulation_steps, ref:train.py:159, 295: every micro-batch's loss is divided by it, gradients add up, and
        the all-reduce / clip / optimizer update run on the last one).
        sync_params: with more than one rank, broadcast rank 0's parameters at construction (DDP does the same), so that
        ranks cannot start from different weights.
        seed: private, reproducible host streams for this runner's model - SpecAugment spans and LayerDrop draws from
        `HFHostRNG.seeded(seed)` (HF's draw order; HF's Trainer seeds every rank alike, so the same seed on every rank), dropout
        mask seeds from `seed` and the rank.  None: the process-global np.random / torch streams, as HF uses them."""
        model._need_engine()
        self.grad_accum = max(1, int(grad_accum))
        self._micro = 0
        self._dropped_all = set()
        self.model, self.store, self.engine = model, model.store, model.engine
        if seed is not None:
            import numpy as _np
            from .engine import HFHostRNG
            rank = dist.get_rank() if dist.is_initialized() else 0
            self.engine.host_rng = HFHostRNG.seeded(seed)
            self.engine.drop_rng = _np.random.default_rng([int(seed), rank])
        self.kind = optimizer
        self.lr, self.betas, self.eps, self.wd, self.max_grad_norm, self.momentum = lr, betas, eps, weight_decay, \
            max_grad_norm, momentum
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        n = self.store.total
        dev = self.store.device
        if optimizer not in ("adamw", "sgd", "adafactor"):
            raise ValueError(f"unknown optimizer {optimizer!r}")
        self.m = torch.zeros(n, dtype=torch.float32, device=dev) if (optimizer == "adamw" or (optimizer == "sgd" and momentum > 0)) \
            else None
        self.v = torch.zeros(n, dtype=torch.float32, device=dev) if optimizer == "adamw" else None
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.t = 0
        ep = self.engine.ep
        pre = ep + "encoder.layers."
        # speech-encoder layer index of every parameter (LayerDrop bookkeeping), -1 elsewhere
        self.layer_of = {nm: (int(nm[len(pre):].split(".", 1)[0]) if nm.startswith(pre) else -1) for nm in self.store.offsets}
        self.af = None
        self._af_split, self._opt_stream, self._af_early, self._early_ok = None, None, None, False
        if optimizer == "adafactor":
            # the reference's optimizer (ref:train.py:298 -> HF Trainer: Adafactor(lr, scale_parameter=False,
            # relative_step=False)); one fused multi-tensor step over the flat buffer (csrc/adafactor.hip).  The plan covers
            # EVERY parameter; the per-step `active` mask carries requires_grad (FreezingCallback toggles it while training,
            # ref:speechmix/module/utility.py:14-29) and LayerDrop, so optimizer state survives a freeze / unfreeze like HF's.
            self.af_names = [nm for nm, _ in sorted(self.store.offsets.items(), key=lambda kv: kv[1][0])]
            self.af = ops.AdafactorPlan([(self.store.offsets[nm][0], self.store.offsets[nm][2]) for nm in self.af_names], dev)
            self._af_layer = [self.layer_of[nm] for nm in self.af_names]
            # The optimizer's tail beside the next step's front end (round 6): the statistics pass and the update of the
            # FRONT-END tensors (waveform CNN, feature projection, positional conv, the encoder's input LayerNorm - a contiguous range of the flat order)
            # stay on the compute stream, the update of everything else goes to a second stream and has to be complete only when the next
            # forward reaches its first encoder layer (Engine.wait_params: eager and replayed steps, and every other reader of parameters).
            # SMX_OPT_OVERLAP=0: off.
            front = [nm.startswith(ep) and not nm.startswith(pre) for nm in self.af_names]
            idx = [i for i, f in enumerate(front) if f]
            self._af_split = (idx[0], idx[-1] + 1) if (os.environ.get("SMX_OPT_OVERLAP", "1") != "0" and dev.type == "cuda" and idx
                                                      and len(idx) == idx[-1] + 1 - idx[0] and len(idx) < len(front)) else None
            self._opt_stream = None
            self._af_early, self._early_ok = None, False
        self._flags = None
        self._force_comm = force_comm
        self.use_graphs = True           # (see _step_graphs)
        self._graphs, self._graph_warm, self._graph_failures = None, {}, 0
        self._graph_trial, self._graph_choice, self.graph_trial_ms = None, {}, None
        self._refresh_trainable()
        if self.world > 1 or force_comm:
            # RCCL runs beside backward: one-workgroup-per-CU kernels leave it CUs (ops.PP_BACKWARD_CUS)
            if ops.PP_BACKWARD_CUS == 0:
                ops.PP_BACKWARD_CUS = ops.PP_RESERVED_DEFAULT
        if self.world > 1 and sync_params:
            dist.broadcast(self.store.master, src=0)
        self.store.external_updates = False          # this runner keeps the bf16 compute copies fresh itself
        self.store.refresh_shadow(force=True)

    def _refresh_trainable(self):
        """(Re)derive everything that depends on the requires_grad flags - update ranges, reduction buckets - when they
        change (the reference's FreezingCallback flips them at a step boundary on every rank alike)."""
        flags = tuple(self.store.requires_grad(nm) for nm in self.store.offsets)
        if flags == self._flags:
            return
        self._flags = flags
        self.ranges = trainable_ranges(self.store)
        self.reducer = GradReducer(self.store.grad, stage_ranges(self.store.offsets, self.model.num_speech_encoder_layers,
                                                                 trainable=self.store.requires_grad),
                                   force_comm=self._force_comm)
        self.engine.stage_cb = self._stage_cb
        self.engine.stage_ranges = self.reducer.stages        # (ranges a reported stage hands to the reducer: Engine._stage)

    # ------------------------------------------------------------------ captured steps (graphs.py)
    def _step_graphs(self, wave, dec_ids, labels, text, fwd_kw):
        """-> the graphs.StepGraphs of this step's configuration, or None (eager).  A configuration - shapes, train / eval flags,
        the requires_grad pattern - is captured once it has run `graphs.WARM_STEPS` eager steps in a row (kernel picks made, the
        first-write gradient ranges learned); another configuration drops the captured one (one set of static activations at
        a time: ~11 GB at config 2).  SMX_STEP_GRAPHS=0, gradient accumulation, live profiling (bench.py's instrumented pass,
        stage marks) and weighted-sum models in train mode stay eager."""
        from . import graphs
        eng, st = self.engine, self.store
        self._trial_mode = None          # the mode this step's timing sample belongs to (set below only on the trial's own steps)
        if (not graphs.ENABLED or not self.use_graphs or st.device.type != "cuda" or self.grad_accum != 1
                or ops.GEMM_PROFILE is not None or ops.OP_PROFILE is not None or getattr(eng, "marks", None) is not None
                or (fwd_kw["weighted_sum"] and fwd_kw["training"]) or self._graph_failures >= 2):
            return None
        key = (tuple(wave.shape), tuple(labels.shape), tuple(text.shape) if text is not None else None, bool(fwd_kw["training"]),
               bool(fwd_kw["lm_training"]), self._flags, eng.dt, ops.PP_BACKWARD_CUS)
        G = self._graphs
        if G is not None and G.key != key:
            G = self._graphs = None                    # another configuration: its static buffers go back to the allocator
            self._graph_warm = {}
            self._graph_trial = None
        if self._graph_choice.get(key) == "eager":     # (auto mode measured this configuration: eager was faster)
            return None
        if G is not None:
            tr = self._graph_trial
            if tr is not None:                         # auto mode's trial: TRIAL_STEPS replayed, then TRIAL_STEPS eager, timed
                if len(tr["replay"]) < graphs.TRIAL_STEPS:
                    self._trial_mode = "replay"
                    return G
                if len(tr["eager"]) < graphs.TRIAL_STEPS:
                    self._trial_mode = "eager"
                    return None
                self._graph_decide(key)
                return self._graphs
            return G
        n = self._graph_warm.get(key, 0)
        warm = max(graphs.WARM_STEPS, 5 if self.world > 1 else 0)       # (N > 1: past the pick broadcasts of steps 1 and 3)
        if n < warm:
            self._graph_warm = {key: n + 1}
            return None
        try:
            G = graphs.StepGraphs(self, key).capture(wave, dec_ids.contiguous(), labels.contiguous(), text, fwd_kw)
        except Exception as e:             # (ops.CaptureAbort included) stay eager; a second failure disables capturing
            import warnings
            self._graph_failures += 1
            self._graph_warm = {}
            warnings.warn(f"StepRunner: HIP-graph capture of the step failed ({str(e)[:300]}); running eagerly")
            self.engine.saved = None
            return None
        self._graphs = G
        if graphs.MODE == "auto" and self._graph_choice.get(key) != "replay":          # (a configuration already measured is not trialled again)
            self._graph_trial = dict(replay=[], eager=[])
            self._trial_mode = "replay"
        return G

    def _graph_decide(self, key):
        """End of auto mode's trial: medians of the event-timed forward + backward sections; the slower mode goes."""
        tr, self._graph_trial = self._graph_trial, None
        med = {}
        for mode in ("replay", "eager"):
            ts = []
            for e0, e1 in tr[mode]:
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            med[mode] = ts[len(ts) // 2]
        # (every rank decides for itself, without a collective: a rank whose capture failed never gets here, and the two schedules
        # issue the same kernels and the same reductions in the same order - ranks in different modes stay bit-identical)
        self.graph_trial_ms = med
        # the replayed schedule has to WIN (2 %): on the GPU's clock the two tie within the noise of a few samples, and the eager schedule is
        # the one every parity figure of the bench line was taken with
        if med["eager"] * 0.98 <= med["replay"]:
            self._graph_choice[key] = "eager"
            self._graphs = None                        # (its static activations go back to the allocator)
        else:
            self._graph_choice[key] = "replay"

    def _stage_cb(self, name):
        """Backward reports a finished stage: its gradient ranges go to the reducer; after the LAST encoder layer - only the front end is
        still in backward - Adafactor's statistics pass over everything but the front-end tensors starts on the optimizer's second stream
        (their gradients are final, and reduced once the comm stream has run; SMX_OPT_OVERLAP)."""
        self.reducer.stage_done(name)
        if name == "enc_layer0" and self._early_ok:
            self._early_ok = False
            st = self.store
            if self._opt_stream is None:
                self._opt_stream = torch.cuda.Stream()          # (torch's priority range here is (0, -1): there is no priority below the default)
            lr = float(self.lr(self.t + 1)) if callable(self.lr) else float(self.lr)
            clip = self.max_grad_norm if self.max_grad_norm and self.max_grad_norm > 0 else 0.0
            dropped = set(self.engine.last_dropped) if self.world == 1 else set()
            active = [f and (l not in dropped) for f, l in zip((st.requires_grad(nm) for nm in self.af_names), self._af_layer)]
            sh = None if st.shadow is st.master else st.shadow
            o, nact = self.af.prepare(st.master, st.grad, sh, lr, active=active, grad_scale=1.0 / self.world, max_grad_norm=clip)
            ev = torch.cuda.Event()
            ev.record()
            self._opt_stream.wait_event(ev)
            if self.reducer.comm_stream is not None:
                self._opt_stream.wait_stream(self.reducer.comm_stream)
            with torch.cuda.stream(self._opt_stream):
                self.af.early_stats(o, self._af_split)
            self._af_early = (o, nact, lr)

    def sync_params(self):
        """Make the current stream wait for an optimizer tail that is still running on the second stream (anything that reads parameters
        outside Engine.forward - a checkpoint, an evaluation loop of another object - calls this or synchronises the device)."""
        self.engine.wait_params()

    def current_lr(self):
        return float(self.lr(self.t)) if callable(self.lr) else float(self.lr)

    def step(self, input_values, labels, decoder_input_ids=None, text_input_ids=None):
        """One optimizer step on this rank's shard.  Returns the (device) loss tensor of this rank."""
        from .model import shift_tokens_right
        m, eng, st = self.model, self.engine, self.store
        lc = m.decoder_model.config
        wave = m._prep_wave(input_values)
        labels = labels.to(st.device)
        if decoder_input_ids is None:
            decoder_input_ids = shift_tokens_right(labels, lc.pad_token_id, lc.decoder_start_token_id)
        if self.world > 1 and self._micro == 0 and self.t in (1, 3):
            # after the first steps have tuned: all ranks adopt rank 0's kernel picks (same kernels, same bf16 roundings, same pace)
            from .dist import share_tuner_picks
            share_tuner_picks()
        ga, first = self.grad_accum, self._micro == 0
        last = self._micro == ga - 1
        self._micro = 0 if last else self._micro + 1
        if first:
            self._refresh_trainable()
            self.reducer.begin_step()
        text = text_input_ids.to(st.device).contiguous() if (text_input_ids is not None and m._uses_text_ids) else None
        training = m.training and m.encoder_model.training
        fwd_kw = dict(training=training, weighted_sum=m.weighted_sum, lm_training=m._lm_training(), want_logits=False)
        self._af_early = None
        self._early_ok = (self.af is not None and getattr(self, "_af_split", None) is not None and ga == 1
                          and os.environ.get("SMX_OPT_EARLY_STATS", "0") == "1")          # (off: measured slower, profiles/r06_probes_not_kept.txt item 6)
        G = self._step_graphs(wave, decoder_input_ids, labels, text, fwd_kw)
        trial = self._graph_trial if getattr(self, "_trial_mode", None) else None          # (steps that are eager for another reason are not samples)
        if trial is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if G is not None:
            # the whole forward + backward as a chain of captured HIP graphs (graphs.py): same kernels, same arguments
            out = G.replay(wave, decoder_input_ids, labels, text)
            out = dict(out, loss=out["loss"].clone())
        else:
            out = eng.forward(wave, decoder_input_ids.contiguous(), labels.contiguous(), text_ids=text, **fwd_kw)
            # micro-batches before the last only add their gradient (no stage reports: nothing is reduced or updated yet)
            cb, eng.stage_cb = eng.stage_cb, (eng.stage_cb if last else None)
            try:
                eng.backward(gscale=1.0 / ga, zero_grads=first)
            finally:
                eng.stage_cb = cb
        # LayerDrop: a layer is without a gradient for this update only if every micro-batch dropped it
        self._dropped_all = set(eng.last_dropped) if first else (self._dropped_all & set(eng.last_dropped))
        if not last:
            return out["loss"]
        self.reducer.finish()
        if m.training and m.encoder_model.training and self.grad_accum == 1:
            # the next step's attention-dropout masks, generated beside the (HBM-bound) optimizer step on a second stream
            eng.pregen_attention_masks(wave.shape[0], out["T"])
        self.t += 1
        lr = self.current_lr()
        inv_world = 1.0 / self.world
        clip = self.max_grad_norm if self.max_grad_norm and self.max_grad_norm > 0 else 0.0
        if clip > 0 and self.af is None:          # (Adafactor takes the norm from its own statistics pass)
            ops.sumsq(st.grad, st.total, self.gnorm_sq)
        sh = None if st.shadow is st.master else st.shadow
        # torch / HF optimizers skip parameters whose .grad is None: a LayerDrop-dropped layer's (on one rank; across ranks the
        # all-reduce gives every tensor a gradient) and frozen ones
        dropped = self._dropped_all if self.world == 1 else set()
        if self.af is not None:
            active = [f and (l not in dropped) for f, l in zip((st.requires_grad(nm) for nm in self.af_names), self._af_layer)]
            if self._af_split is not None:
                if self._opt_stream is None:
                    self._opt_stream = torch.cuda.Stream()          # (torch's priority range here is (0, -1): there is no priority below the default)
                early, self._af_early = self._af_early, None
                if early is not None:          # (the statistics of everything but the front end are already running: _stage_cb)
                    early[0].lr = lr
                    eng.param_event = self.af.finish(early[0], early[1], self._af_split, self._opt_stream, early=True)
                else:
                    eng.param_event = self.af.step(st.master, st.grad, sh, lr, active=active, grad_scale=inv_world, max_grad_norm=clip,
                                                   split=self._af_split, tail_stream=self._opt_stream)
            else:
                self.af.step(st.master, st.grad, sh, lr, active=active,
                             grad_scale=inv_world, max_grad_norm=clip)
            st.mark_shadow_fresh()
            if trial is not None:          # (the WHOLE step: forward + backward alone favoured the replay by up to 1.8 ms on steps that then tied)
                ev[1].record()
                trial[self._trial_mode].append(ev)
            return out["loss"]
        ranges = self.ranges if not dropped else trainable_ranges(st, dropped, self.layer_of)
        for a, b in ranges:
            ops.optimizer_step(st.master[a:b], st.grad[a:b], self.m[a:b] if self.m is not None else None,
                               self.v[a:b] if self.v is not None else None, sh[a:b] if sh is not None else None,
                               self.gnorm_sq if clip > 0 else None, b - a, lr, kind=self.kind,
                               beta1=self.betas[0] if self.kind == "adamw" else self.momentum, beta2=self.betas[1],
                               eps=self.eps, weight_decay=self.wd, step=self.t, grad_scale=inv_world, max_grad_norm=clip)
        st.mark_shadow_fresh()
        if trial is not None:
            ev[1].record()
            trial[self._trial_mode].append(ev)
        return out["loss"]
