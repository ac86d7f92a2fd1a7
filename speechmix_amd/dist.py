"""Data-parallel gradient reduction for the flat gradient buffer (RCCL over xGMI on MI355X, gloo on CPU).

The reference gets data parallelism implicitly from HF Trainer -> accelerate -> DistributedDataParallel
(TF:trainer.py:720-737): a bucketed sum-all-reduce of every trainable gradient.  Here the
gradients already live in ONE flat fp32 buffer (params.FlatStore), so a bucket is just a contiguous range of
it.  Buckets are declared per backward *stage* (LM, bridge, each speech-encoder layer, front end); as soon
as the engine reports a stage finished, its ranges are all-reduced on a side HIP stream while the compute
stream keeps running the rest of backward.  Parameters that received no gradient this step (layerdrop,
frozen sets) simply contribute zeros, which keeps every rank's collective sequence identical.
One process per GPU; no collective on the forward path (pure data parallel: each clip is independent).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

# How a bucket is summed across ranks (SMX_ALLREDUCE):
#   sum    one fp32 all-reduce per bucket (default)
#   rs_ag  reduce-scatter + all-gather on the bucket's world-divisible part (two half-volume collectives: on point-to-point xGMI
#          links the ring all-reduce is per-link bound, and the two halves can be scheduled apart); remainder all-reduced
#   bf16   the bucket is cast to bf16, all-reduced, cast back: half the bytes over xGMI, bf16 rounding of the summed gradient
#          (the sum itself runs in bf16: a precision trade the reference's fp32 DDP buckets do not make - opt-in only)
# Every rank must use the same mode.  UNMEASURED on hardware by this builder (single-GPU boxes): switches for the scaling runs.
ALLREDUCE_MODE = os.environ.get("SMX_ALLREDUCE", "sum")


def reduce_bucket(t: torch.Tensor, group=None, mode: Optional[str] = None):
    """Sum `t` (a contiguous fp32 slice of the flat gradient) over the group, in place."""
    mode = mode or ALLREDUCE_MODE
    world = dist.get_world_size(group)
    if mode == "bf16":
        c = t.to(torch.bfloat16)
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
        return
    if mode == "rs_ag" and world > 1:
        n = t.numel() - t.numel() % world
        if n:
            sh = n // world
            r = dist.get_rank(group)
            mine = t[r * sh:(r + 1) * sh]
            if dist.get_backend(group) == "nccl":
                dist.reduce_scatter_tensor(mine, t[:n], op=dist.ReduceOp.SUM, group=group)
                dist.all_gather_into_tensor(t[:n], mine, group=group)
            else:
                # gloo has no reduce-scatter: the same shard arithmetic on collectives it has (the CPU / shared-GPU tests run THIS
                # branch, so the offsets above are exercised somewhere; RCCL's two calls have run on no hardware of this builder)
                full = t[:n].clone()
                dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
                shard = full[r * sh:(r + 1) * sh].clone()                       # what reduce_scatter_tensor would leave in `mine`
                dist.all_gather([t[i * sh:(i + 1) * sh] for i in range(world)], shard, group=group)
        if n < t.numel():
            dist.all_reduce(t[n:], op=dist.ReduceOp.SUM, group=group)
        return
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def share_tuner_picks(group=None, src: int = 0):
    """Every rank adopts rank `src`'s kernel picks (speechmix_amd.ops tuner state): ranks that tuned independently can
    launch different kernel variants for the same shape - different K splits, hence different bf16 roundings and different
    per-rank step times.  Picks a rank made that `src` has not are kept."""
    from . import ops
    box = [ops.tuner_state() if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    ops.load_tuner_state(box[0])


def stage_ranges(offsets: Dict[str, Tuple[int, int, tuple]], num_speech_layers: int, max_bucket_elems: int = 64 << 20,
                 trainable=None):
    """Contiguous flat ranges per backward stage, in the order backward completes them.  trainable(name) -> bool: only
    parameters that can receive a gradient are reduced (the flags are the same on every rank, so the bucket list is too; a
    frozen LM - SpeechMixSelf / SpeechMixFixed - would otherwise all-reduce ~560 MB of zeros per step)."""
    def span(pred):
        sel = [(o, o + n) for name, (o, n, _) in offsets.items() if pred(name) and (trainable is None or trainable(name))]
        if not sel:
            return []
        sel.sort()
        out = [list(sel[0])]
        for a, b in sel[1:]:
            if a - out[-1][1] <= 64:         # merge across alignment padding (FlatStore.ALIGN), never across a frozen tensor
                out[-1][1] = max(out[-1][1], b)
            else:
                out.append([a, b])
        chunks = []
        for a, b in out:
            while b - a > max_bucket_elems:
                chunks.append((a, a + max_bucket_elems))
                a += max_bucket_elems
            chunks.append((a, b))
        return chunks

    stages = [("lm", span(lambda n: n.startswith(("decoder_model.", "adapters."))))]
    stages.append(("bridge", span(lambda n: n.startswith(("length_adapters.", "enc_to_dec_proj.", "weights_sum")))))
    for i in range(num_speech_layers - 1, -1, -1):
        pre = f"encoder_model.encoder.layers.{i}."
        stages.append((f"enc_layer{i}", span(lambda n, pre=pre: n.startswith(pre))))
    stages.append(("frontend", span(lambda n: n.startswith("encoder_model.") and ".encoder.layers." not in n
                                    and not n.startswith("encoder_model.encoder.layers."))))
    return stages


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, stages, group=None, force_comm: bool = False):
        """force_comm: issue the collectives even in a 1-rank group (exercises the RCCL / side-stream path on one GPU)."""
        self.g = flat_grad
        self.stages = dict(stages)
        self.order = [s for s, _ in stages]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force_comm = force_comm
        self.active = self.world > 1 or (force_comm and dist.is_initialized())
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda and self.active else None
        self._done = set()
        # profile=True (bench.py's instrumented pass): HIP events around every stage's collectives on the comm stream and around
        # the compute stream's final wait -> comm_stats(): how long the reductions ran and how much of that backward did NOT hide
        self.profile = False
        self._ev, self._wait_ev, self._bytes, self._steps = [], [], 0, 0
        covered = sorted(r for rs in self.stages.values() for r in rs)
        for (a0, b0), (a1, b1) in zip(covered[:-1], covered[1:]):
            if b0 > a1:
                raise RuntimeError("overlapping reduction buckets")

    def begin_step(self):
        self._done.clear()

    def stage_done(self, name: str):
        """Called by the engine on the compute stream right after the stage's last gradient kernel."""
        if not self.active or name in self._done or name not in self.stages:
            self._done.add(name)
            return
        self._done.add(name)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if self.profile:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                for a, b in self.stages[name]:
                    reduce_bucket(self.g[a:b], self.group)
                if self.profile:
                    e1.record()
                    self._ev.append((name, e0, e1))
                    self._bytes += 4 * sum(b - a for a, b in self.stages[name])
        else:
            for a, b in self.stages[name]:
                reduce_bucket(self.g[a:b], self.group)

    def finish(self):
        """Reduce whatever has not been reported yet and make the compute stream wait for the side stream."""
        for name in self.order:
            if name not in self._done:
                self.stage_done(name)
        if self.cuda and self.comm_stream is not None:
            if self.profile:
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w0.record()
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            if self.profile:
                w1.record()
                self._wait_ev.append((w0, w1))
                self._steps += 1

    def comm_stats(self):
        """After a device synchronize: per-step figures of the profiled steps - milliseconds the collectives ran on the comm
        stream (sum over stages), milliseconds the compute stream stood waiting for them after backward (the EXPOSED part),
        bytes reduced, collectives issued; per-stage milliseconds."""
        n = max(self._steps, 1)
        per_stage = {}
        for name, e0, e1 in self._ev:
            per_stage[name] = per_stage.get(name, 0.0) + e0.elapsed_time(e1)
        total = sum(per_stage.values())
        return dict(steps=self._steps, allreduce_ms_per_step=total / n,
                    exposed_ms_per_step=sum(w0.elapsed_time(w1) for w0, w1 in self._wait_ev) / n,
                    bytes_per_step=self._bytes / n, collectives_per_step=len(self._ev) / n,
                    lm_stage_ms_per_step=per_stage.get("lm", 0.0) / n, mode=ALLREDUCE_MODE)


def shard_batch(n_items: int, rank: int, world: int):
    """Even split of a global batch across ranks (ref semantics: DistributedSampler without padding)."""
    per = n_items // world
    return rank * per, (rank + 1) * per
