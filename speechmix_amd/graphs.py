"""A whole training step as a chain of captured HIP graphs (round 5).

Why: a step is ~800 kernel launches and the Python host needs ~25 us for each (ctypes call, parameter struct, kernel choice,
allocator) - 21.6 ms of host time per 31.6-ms step at config 2 (profiles/r04_host_profile.txt), with the LM stages enqueued
barely faster than the GPU executes them.  The step's launches do not depend on the data, only on shapes, so after a few eager
steps (kernel picks made, first-write gradient ranges learned) ONE pass of the ordinary engine code runs under stream capture
and is cut into graphs at the places where a later step must be free to do something else:

    front | enc_fwd0 .. enc_fwd{L-1} | (bridge, LM forward, loss, LM backward) stage:lm | stage:bridge | pre_layers |
    stage:enc_layer{L-1} .. stage:enc_layer0 | stage:frontend | tail

* LayerDrop (TF:models/wav2vec2/modeling_wav2vec2.py:709-723): a dropped layer's two graphs are not replayed; its output
  buffers (hidden state forward, gradient backward) receive a copy of its input and its gradient ranges are zeroed.
* Data parallelism: between two graphs the replay reports the finished stage to dist.GradReducer exactly where the eager
  backward does (Engine._stage), so the all-reduce of a stage still overlaps the rest of backward.
* Per-step randomness: SpecAugment spans and LayerDrop draws are taken from HF's host streams in HF's order and reach the
  device through a fixed-capacity row list / the choice of graphs; dropout masks change through the library's step key
  (Engine.begin_pass), which the replay sets ahead of the first graph - seeds are baked, the key is not.
* Everything a graph touches lives at a fixed address: activations in the capture's private memory pool (shared by all graphs
  of the chain, so a buffer freed in backward is reused exactly as in the eager step), inputs in static tensors.

The captured chain replays the SAME kernels with the SAME arguments as the eager step: gradients are bit-identical
(tests/test_gpu_r5.py).  SMX_STEP_GRAPHS=0 keeps StepRunner eager.
"""
from __future__ import annotations

import gc
import os
import warnings
from typing import Dict, List

import torch

from . import ops

# SMX_STEP_GRAPHS = auto (default) | 1 | 0.  What replay buys is HOST time (config 2: 18 ms of Python per step eager, 2.5 ms replayed);
# on the GPU's clock the replayed step is the same kernels plus ~9 us of idle at each of its ~30 graph boundaries and at every
# cross-queue dependency inside a graph (measured round 5: 32.05 vs 31.4 ms where the host keeps ahead of the GPU anyway).  So `auto`
# MEASURES: after the capture, TRIAL_STEPS replayed and TRIAL_STEPS eager steps are timed with HIP events around the whole step
# and the replayed schedule is kept if it wins by 2 % (else its captured buffers are released).  `1` always replays.
MODE = os.environ.get("SMX_STEP_GRAPHS", "auto")
ENABLED = MODE != "0"
WARM_STEPS = int(os.environ.get("SMX_GRAPH_WARM_STEPS", "3"))      # eager steps of a configuration before its capture
TRIAL_STEPS = int(os.environ.get("SMX_GRAPH_TRIAL_STEPS", "5"))
CAPTURE_MODE = os.environ.get("SMX_CAPTURE_MODE", "thread_local")          # hipStreamCaptureMode of the captures (see StepGraphs._begin)


CaptureAbort = ops.CaptureAbort


class StepGraphs:
    """One captured configuration of `StepRunner.step`: (batch, samples, label length, text length, train / eval flags, the
    requires_grad pattern)."""

    def __init__(self, runner, key):
        self.runner, self.key = runner, key
        self.graphs: List = []                  # [(name, CUDAGraph)] in capture = replay order
        self.carry: Dict[str, torch.Tensor] = {}
        self.pool = None
        self.rows_dev = self.rows_pin = None
        self.rows_ev = None
        self.out = None
        self.static = {}
        self._cur = None

    # ---- called by the engine during the capture pass
    def boundary(self, closed, carry):
        self.carry.update(carry)
        g = self._cur
        g.capture_end()
        self.graphs.append((closed, g))
        self._begin()

    def note(self, carry):
        self.carry.update(carry)

    def spec_rows(self, eng, B, T):
        if self.rows_dev is None:
            raise CaptureAbort("SpecAugment row list was not allocated ahead of the capture")
        return self.rows_dev

    def _alloc_rows(self, eng, B, T):
        """The fixed-capacity SpecAugment row list (device + pinned twin), allocated AHEAD of the capture: a fill inside it would
        be replayed and wipe the rows."""
        ec = eng.ec
        if not (ec.apply_spec_augment and ec.mask_time_prob > 0 and eng.has(eng.ep + "masked_spec_embed")):
            return
        # every clip gets the same number of spans (compute_mask_indices pads the shorter lists): <= int(p T / len + 1) each
        cap = B * max(int(ec.mask_time_prob * T / ec.mask_time_length + 1), ec.mask_time_min_masks) * ec.mask_time_length
        cap = max(min(cap, B * T), 1)
        self.rows_dev = torch.full((cap,), -1, dtype=torch.int32, device=eng.dev)
        self.rows_pin = torch.full((cap,), -1, dtype=torch.int32).pin_memory()

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # thread_local: only THIS thread's calls are checked against the capture.  Under the default ("global") a hipEventQuery from any other
        # thread while the capture is open fails with hipErrorStreamCaptureUnsupported - and torch.distributed's NCCL watchdog thread polls its
        # work events every few hundred milliseconds: the process then dies in the watchdog (seen once in four runs of tests/test_gpu_dist_single.py)
        g.capture_begin(pool=self.pool, capture_error_mode=CAPTURE_MODE)
        self._cur = g

    # ---- capture
    def capture(self, wave, dec_ids, labels, text, fwd_kwargs):
        eng = self.runner.engine
        if os.environ.get("SMX_WGRAD_SIDE", "0") == "1":
            raise CaptureAbort("SMX_WGRAD_SIDE=1 carries a launch across a stage boundary")
        # the graphs read their inputs from tensors this object owns
        self.static = dict(wave=wave.clone(), dec=dec_ids.clone(), labels=labels.clone(), text=text.clone() if text is not None else None)
        wave, dec_ids, labels, text = (self.static[k] for k in ("wave", "dec", "labels", "text"))
        training = bool(fwd_kwargs.get("training"))
        if training:
            self._alloc_rows(eng, wave.shape[0], eng._conv_geom(wave.shape[1])[-1])
        self.uses_premask = training and getattr(eng, "_premask", None) is not None
        eng._ensure_gplan()
        torch.cuda.synchronize()
        # ONE capture stream per engine, not one per capture: torch hands out its 32 pool streams round-robin, so a process that keeps asking
        # for streams is eventually handed the very stream RCCL's process group launches on - and an event last recorded on a capturing stream
        # cannot be queried from any thread, in any capture mode (tools/gpu_capture_query_probe.py)
        side = getattr(eng, "_capture_stream", None)
        if side is None:
            side = eng._capture_stream = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ops.CAPTURING = True
        eng._cap = self
        rng_state = eng.drop_rng.bit_generator.state
        ok = False
        # No garbage collection while the capture is open: the capture pass runs a few thousand lines of ordinary Python, the cyclic collector
        # may fire anywhere in it, and whatever it finalises then - an older StepGraphs with its graphs and pool, a stream, an event - makes a
        # HIP call that is illegal on a capturing thread; the error is raised inside a destructor and the process aborts (seen once in the
        # GPU suite, "Fatal Python error: Aborted / Garbage-collecting" under _gemm_params).  So the collector runs right BEFORE the capture
        # (what torch.cuda.graph does) and is held while it is open.  SMX_CAPTURE_GC_GUARD=hold: held only; 0: no guard.
        # (Round 5 held it without collecting because collecting "corrupted one replayed hidden state" of the weighted-sum model.  Round 6
        # found the cause, and it was not the collector: smx_weighted_sum_bwd cleared its 20-byte scratch with hipMemsetAsync, the capture
        # turned that into a memset node, and the replayed node did not clear it - the first weight's dot product started from whatever the
        # pool block held, which depended on what had been freed before the capture.  The scratch is zeroed by a kernel now: 6 of 6 runs
        # of the test pass with the collect guard, 5 of 5 failed before.)
        gc_was = gc.isenabled()
        gg = os.environ.get("SMX_CAPTURE_GC_GUARD", "collect")
        if gg != "0":
            if gg != "hold":
                gc.collect()
            gc.disable()
        try:
            with torch.cuda.stream(side):
                self._begin()
                out = eng.forward(wave, dec_ids, labels, text_ids=text, **fwd_kwargs)
                eng.backward(gscale=1.0, zero_grads=True)
                g = self._cur
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")          # (nothing may follow the last stage: an empty graph is fine)
                    g.capture_end()
                self.graphs.append(("tail", g))
            ok = True
        finally:
            if gc_was:
                gc.enable()
            eng._cap = None
            ops.CAPTURING = False
            eng.drop_rng.bit_generator.state = rng_state
            if not ok:
                # A pass that stopped half-way (CaptureAbort behind a fork: a missing kernel pick in backward, a new transposed-weight entry)
                # leaves the forked streams IN the capture: ending the capture then fails as "unjoined", the streams stay in capture mode, and
                # the next HIP call that is illegal on a capturing thread - a CUDAGraph destructor - takes the process down (seen once with
                # SMX_TUNE=live, round 6).  Join every stream that is capturing back into the origin first; the graph is discarded anyway.
                for name in ("_side", "_wg_side", "_cs_stream", "_mask_stream"):
                    st2 = getattr(eng, name, None)
                    if st2 is None:
                        continue
                    try:
                        with torch.cuda.stream(st2):
                            forked = torch.cuda.is_current_stream_capturing()
                        if forked:
                            side.wait_stream(st2)
                    except Exception:
                        pass
                try:
                    if self._cur is not None:
                        with torch.cuda.stream(side):          # (the capture ends on the stream it began on - this handler runs outside the `with` above)
                            self._cur.capture_end()
                except Exception:
                    pass
                self._cur = None                               # (the open graph goes first: its destructor ends a capture that is still open)
                self.graphs.clear()
                eng.reset_side_state()          # (a backward that stopped half-way: nothing it queued may reach the next eager step)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.out = dict(loss=out["loss"], argmax=out["argmax"], T=out["T"])
        self.L = eng.L
        self.training = bool(fwd_kwargs.get("training"))
        lmt = fwd_kwargs.get("lm_training")
        self.keyed_pass = self.training or bool(lmt)          # Engine.forward: begin_pass(training or lm_training)
        return self

    # ---- replay
    def replay(self, wave, dec_ids, labels, text):
        r = self.runner
        eng = r.engine
        st = self.static
        for name, src in (("wave", wave), ("dec", dec_ids), ("labels", labels), ("text", text)):
            dst = st[name]
            if dst is not None and src is not dst and src.data_ptr() != dst.data_ptr():
                if src.is_cuda and src.dtype == dst.dtype and src.is_contiguous() and not (src.data_ptr() & 15):
                    ops.copy_bytes(src, dst)
                else:
                    dst.copy_(src, non_blocking=True)
        B = st["wave"].shape[0]
        T = self.out["T"]
        kept = [True] * self.L
        if self.training:
            ec = eng.ec
            # host streams in the eager order: SpecAugment spans first, then one LayerDrop draw per layer
            if self.rows_dev is not None:
                rows = eng._spec_augment_rows(B, T, host_only=True)
                if self.rows_ev is not None:
                    self.rows_ev.synchronize()          # (the previous step's copy has long completed)
                n = rows.size
                if n > self.rows_pin.numel():
                    raise RuntimeError("SpecAugment drew more rows than the captured capacity")
                self.rows_pin[:n] = torch.from_numpy(rows)
                self.rows_pin[n:] = -1
                ops.copy_bytes(self.rows_pin, self.rows_dev)          # (a kernel reading the pinned list: no hipMemcpy)
                self.rows_ev = torch.cuda.Event()
                self.rows_ev.record()
            for i in range(self.L):
                kept[i] = not (eng.host_rng.layerdrop() < ec.layerdrop)
            if self.uses_premask:
                # the layer graphs read the pre-generated attention-dropout bit masks: they must have been generated with the key
                # this pass is about to use (StepRunner generates them beside the previous optimizer step; an evaluation pass or
                # another configuration in between voids them)
                pm = getattr(eng, "_premask", None)
                if pm is None or pm.get("key") is None or pm["key"] != getattr(eng, "_preset_key", None) or (pm["B"], pm["T"]) != (B, T):
                    eng.pregen_attention_masks(B, T)
                    pm = eng._premask
                if pm is None:
                    raise RuntimeError("graph replay: attention dropout masks could not be generated")
        else:
            for i in range(self.L):
                eng.host_rng.layerdrop()                 # HF draws in eval mode too (Engine.speech_fwd)
        if self.keyed_pass:
            # the step key (set on the stream ahead of the first graph) - also with the speech encoder in eval mode and the LM in train
            # mode: the captured LM dropout sites hash with the key, and a replay that never refreshed it drew the same masks every step
            eng.begin_pass(True)
        if self.training and self.uses_premask and pm.get("ev") is not None:
            torch.cuda.current_stream().wait_event(pm["ev"])
            pm["ev"] = None
        cb = eng.stage_cb
        sr = getattr(eng, "stage_ranges", None) or {}
        c = self.carry
        tr = getattr(self, "trace", None)            # tools/gpu_stage_compare.py: (name, HIP event) after every graph of the chain
        if tr is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            tr.append(("start", ev))
        for name, g in self.graphs:
            if name.startswith("enc_fwd"):
                i = int(name[7:])
                if not kept[i]:
                    ops.copy_bytes(c[f"fwd_in{i}"], c[f"fwd_out{i}"])
                    continue
            elif name.startswith("stage:enc_layer"):
                i = int(name[15:])
                if not kept[i]:
                    if c[f"bwd_out{i}"] is not c[f"bwd_in{i}"]:
                        ops.copy_bytes(c[f"bwd_in{i}"], c[f"bwd_out{i}"])
                    for a, b in sr.get(name[6:], ()):
                        eng.st.grad[a:b].zero_()         # a dropped layer contributes zeros (and holds no stale gradient)
                    if cb is not None:
                        cb(name[6:])
                    continue
            g.replay()
            if name == "front":
                eng.wait_params()          # the optimizer's tail of the previous step (trainer.py): complete before the first encoder layer
            if tr is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                tr.append((name, ev))
            if cb is not None and name.startswith("stage:"):
                cb(name[6:])
        eng.last_dropped = [i for i in range(self.L) if not kept[i]]
        eng.note_dropped(True)                           # (a replayed backward zeroes the gradients first, as its capture pass did)
        eng.saved = None
        return self.out
