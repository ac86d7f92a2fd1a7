"""Hand-scheduled forward + backward of the SpeechMix training step on HIP kernels.

This is the MI355X-native replacement for what PyTorch autograd + HF modules execute under
`SpeechMixEED.forward` / `loss.backward()` in the reference (ref:speechmix/model.py:139-177):
an explicit stage list (CNN feature extractor -> projection -> positional conv -> encoder layers -> length
adapters -> enc_to_dec_proj -> LM text encoder -> LM decoder -> LM head + CE), each stage with a forward that
saves exactly what its hand-written backward needs.  All arithmetic is in libspeechmix_hip.so; torch is used
for device buffers only.  Data layout: activations are token-major / channels-last `[B*T, C]` in the compute
dtype (bf16, or fp32 on the parity path); every Conv1d is a GEMM over an overlapping row view of that layout
(no im2col, no col2im); parameters are read from the FlatStore's compute copy, gradients are accumulated in
fp32 straight into the FlatStore's flat gradient buffer (the thing RCCL all-reduces).
"""
from __future__ import annotations

import contextlib
import gc
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .configs import LMConfig, SpeechEncoderConfig
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, BF16, F32, view
from .params import FlatStore

_ACT = {"gelu": ACT_GELU, "relu": ACT_RELU}
_WGRAD_GROUP = os.environ.get("SMX_WGRAD_GROUP", "1") != "0"
_COLSUM_SIDE = os.environ.get("SMX_COLSUM_SIDE", "1") != "0"


def _act_id(name):
    if name not in _ACT:
        raise ValueError(f"activation {name!r} is not supported by the HIP kernels (gelu / relu)")
    return _ACT[name]


# Data gradients through K-contiguous weight copies (Engine._wt): OFF by default - the GEMMs get 0.3 ms per step faster and the step does not
# (profiles/r06_probes_not_kept.txt: 31.57 vs 32.14 ms, three same-box runs each); SMX_DGRAD_WT=1 turns it on.
WT_MODE = os.environ.get("SMX_DGRAD_WT", "0") == "1"
WT_MIN_M = int(os.environ.get("SMX_DGRAD_WT_MINM", "4096"))

class HFHostRNG:
    """The host-side random streams HuggingFace draws from in train mode: legacy `np.random` for the SpecAugment spans
    (TF:models/wav2vec2/modeling_wav2vec2.py:139 `np.random.rand(1)`, :183 `np.random.choice`) and torch's CPU generator for
    LayerDrop (TF:...wav2vec2.py:712 `torch.rand([])`).  With no arguments the PROCESS-GLOBAL streams are used, so
    `np.random.seed(k)` / `torch.manual_seed(k)` reproduce HF's indices and keep decisions bit for bit
    (tests/test_train_mode_r3.py); `HFHostRNG.seeded(k)` owns private streams with the same draw order
    (`StepRunner(..., seed=k)` installs one).  HF evaluates `torch.rand([])` once per layer in EVAL mode too (the draw precedes
    `self.training and ...`), so `Engine.speech_fwd` draws unconditionally as well: the stream stays aligned with HF's across
    interleaved evaluation passes."""

    def __init__(self, np_state=None, torch_gen=None):
        self.np, self.tg = np_state, torch_gen

    @classmethod
    def seeded(cls, seed):
        g = torch.Generator()
        g.manual_seed(int(seed))
        return cls(np.random.RandomState(int(seed)), g)

    def rand(self):
        return (self.np.rand(1) if self.np is not None else np.random.rand(1)).item()

    def choice(self, n, k):
        src = self.np if self.np is not None else np.random
        return src.choice(np.arange(n), k, replace=False)

    def layerdrop(self):
        return (torch.rand([], generator=self.tg) if self.tg is not None else torch.rand([])).item()


class RecordedHostRNG:
    """Plays back recorded decisions (fixtures): `mask` = boolean [B, T] SpecAugment mask or None, `keep` = per-layer booleans."""

    def __init__(self, mask=None, keep=None, fallback=None):
        self.mask = None if mask is None else np.asarray(mask, dtype=bool)
        self.keep = None if keep is None else [bool(k) for k in keep]
        self._i = 0
        self.fallback = fallback or HFHostRNG()          # a keep-only recording with mask_time_prob > 0 still draws its spans

    def rand(self):
        return self.fallback.rand()

    def choice(self, n, k):
        return self.fallback.choice(n, k)

    def layerdrop(self):                       # < layerdrop <=> dropped
        if self.keep is None:
            return 2.0
        k = self.keep[self._i % len(self.keep)]
        self._i += 1
        return 2.0 if k else -1.0


def compute_mask_indices(shape, mask_prob, mask_length, rng, lengths=None, min_masks=0):
    """TF:models/wav2vec2/modeling_wav2vec2.py:101-218 `_compute_mask_indices`, draw for draw: one `rand` for the probabilistic
    rounding, then per batch row one `choice` of span starts without replacement over `input_length - (mask_length - 1)`
    positions (rows shorter than the batch maximum pad their start list with their first start; an empty row uses the last
    frame), spans clipped to the sequence end.  -> boolean [B, T]."""
    B, T = shape
    if mask_length < 1:
        raise ValueError("`mask_length` has to be bigger than 0.")
    if mask_length > T:
        raise ValueError(f"`mask_length` has to be smaller than `sequence_length`, but got `mask_length`: {mask_length} and `sequence_length`: {T}`")
    eps = rng.rand()

    def num_spans(n):
        k = max(int(mask_prob * n / mask_length + eps), min_masks)
        if k * mask_length > T:
            k = T // mask_length
        if n - (mask_length - 1) < k:
            k = max(n - (mask_length - 1), 0)
        return k

    lens = [T] * B if lengths is None else [int(x) for x in lengths]
    mask = np.zeros((B, T), dtype=bool)
    kmax = num_spans(T)
    if kmax == 0:
        return mask
    for b, n in enumerate(lens):
        k = num_spans(n)
        starts = rng.choice(n - (mask_length - 1), k)
        dummy = T - 1 if len(starts) == 0 else starts[0]
        starts = np.concatenate([starts, np.ones(kmax - k, dtype=np.int32) * dummy])
        idx = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1)
        idx[idx > T - 1] = T - 1
        mask[b, idx] = True
    return mask


def site_seed(name, n=0):
    """The fixed 32-bit dropout seed of a named site (keyed mode, Engine._dp): crc32 of the name, never 0."""
    import zlib
    return (zlib.crc32(f"{name}#{n}".encode()) & 0xffffffff) or 1


class Engine:
    def __init__(self, store: FlatStore, enc_cfg: SpeechEncoderConfig, lm_cfg: LMConfig, dtype: int,
                 num_speech_layers: int, down_scale: int, enc_prefix="encoder_model.", lm_prefix="decoder_model."):
        self.st = store
        self.ec, self.lc = enc_cfg, lm_cfg
        self.dt = dtype
        self.tdt = ops.torch_dtype(dtype)
        self.dev = store.device
        self.L = num_speech_layers
        self.downloop = int(math.log(down_scale, 2)) if down_scale > 1 else 0
        self.ep, self.lp = enc_prefix, lm_prefix
        self._persist: Dict[str, torch.Tensor] = {}
        self.last_dropped: List[int] = []
        # encoder layers that LayerDrop skipped in EVERY backward since the gradients were last zeroed (gradient accumulation: a layer kept in
        # one micro-batch holds a gradient; FusedAdafactor / StepRunner skip only these - what HF's `grad is None` amounts to)
        self.dropped_since_zero: set = set()
        self.saved = None
        rank = int(os.environ.get("RANK", "0"))
        # SpecAugment spans and LayerDrop decisions: HF's own host streams in HF's draw order (injectable: tests, replays)
        self.host_rng = HFHostRNG()
        self.drop_rng = np.random.default_rng(0x5eed + rank)   # per-site dropout seeds (masks are regenerated in backward)
        self.stage_cb = None      # callable(stage_name): gradient ranges of that stage are final (dist.GradReducer)
        self.lm_adapters = store.has_prefix("adapters.")      # SpeechMixAdapter: bottleneck adapters behind every LM layer

    # ------------------------------------------------------------------ dropout seeds and the step key (round 5)
    # Rounds 1-4 drew a fresh 32-bit seed per dropout site per step and passed it as a kernel argument - which a captured HIP
    # graph would bake.  Now (SMX_STEP_KEY=0: the old scheme) a site's seed is a FIXED function of its name
    # ("enc3/attn", "lmd0/o", ...) and what changes from step to step is the library's step key, one device word that every
    # hashing kernel adds to its seed (csrc/smx_common.h smx_dseed): `begin_pass` draws the key from `drop_rng` (one draw per
    # forward pass) and sets it on the stream ahead of the pass; backward re-sets it only if another pass has set a different
    # key in between (two models alternating).  Eager and graph-replayed steps with the same key draw the same masks.
    def keyed(self):
        return self.dev.type == "cuda" and os.environ.get("SMX_STEP_KEY", "1") != "0"

    def begin_pass(self, training=True):
        """A new forward pass begins (one per model.forward / StepRunner micro-step): in training mode it gets a fresh step key
        (or the one `pregen_attention_masks` already set for it).  -> the key, or None (unkeyed mode / eval)."""
        self._scope, self._site_seen = "", {}
        if not training or not self.keyed():
            self._key = None
            return None
        if self._cap is not None:                  # capture pass: site seeds do not depend on the key, and the replay sets it
            self._key = 1
            return 1
        k = getattr(self, "_preset_key", None)
        self._preset_key = None
        if k is None:
            k = int(self.drop_rng.integers(1, 2 ** 32 - 1))
        self._key = k
        self._ensure_key(k)
        return k

    def _ensure_key(self, key):
        if self._cap is None and key is not None and ops.CURRENT_KEY.get(self.dev.index) != key:
            ops.set_step_key(key, self.dev.index)

    def _dp(self, p, site=""):
        """(p, 32-bit seed) for one dropout site, or None when the site is inactive.  site: the site's name inside the current
        scope (`self._scope`); keyed mode derives the seed from it, the old scheme draws a fresh one."""
        if not (p and p > 0):
            return None
        if getattr(self, "_key", None) is None:
            return (float(p), int(self.drop_rng.integers(1, 2 ** 32 - 1)))
        name = f"{self._scope}/{site}"
        n = self._site_seen.get(name, 0)              # (a name met twice in one pass - a second LM pass - gets its own seed)
        self._site_seen[name] = n + 1
        return (float(p), site_seed(name, n))

    def _dropped(self, dy, drop, n, bias_grad=None, N=0):
        """dy * (the forward mask of a dropout site): gradient entering the dropped branch.  bias_grad [N]: also accumulate
        the column sums of the result (the bias gradient of the Linear whose output was dropped) in the same pass; returns
        (masked dy, True) then, (dy, False) when nothing was fused."""
        if bias_grad is None:
            if drop is None:
                return dy
            out = torch.empty_like(dy)
            ops.dropout(dy, out, n, drop[0], drop[1], self.dt)
            return out
        if drop is None or N <= 0 or (N & 7) or os.environ.get("SMX_FUSE_DROPCOL") == "0":       # (env: A/B switch)
            return self._dropped(dy, drop, n), False
        out = torch.empty_like(dy)
        ops.dropout_colsum(dy, out, n // N, N, drop[0], drop[1], bias_grad, self.dt, folds=self.folds)
        return out, True

    @property
    def folds(self):
        """Queue of deferred second-stage reductions (ops.FoldQueue), or None with SMX_DEFER_FOLDS=0 (A/B switch)."""
        f = getattr(self, "_folds", False)
        if f is False:
            f = self._folds = None if os.environ.get("SMX_DEFER_FOLDS") == "0" else ops.FoldQueue()
        return f

    # ------------------------------------------------------------------ HIP-graph capture of a whole step (graphs.py, round 5)
    # `self._cap` is a graphs.StepGraphs while ITS capture pass runs this engine's forward / backward under stream capture, else
    # None.  The pass closes one graph and opens the next at every `_seg` call (the places where a replayed step must be able to
    # do something else: skip a LayerDrop-dropped layer, hand a finished stage to the gradient reducer), takes no host-RNG draws
    # (every layer kept; SpecAugment rows come from a fixed-capacity device list that the replay refills) and sets no step key
    # (the replay sets it, ahead of the first graph).
    _cap = None

    def _seg(self, closed, **carry):
        """Boundary between two graphs of a captured step: `closed` names the graph that ends here; carry: tensors a replay may
        need by name (a layer's input / output for the LayerDrop copy)."""
        if self._cap is not None:
            self._cap.boundary(closed, carry)

    def _note(self, **carry):
        if self._cap is not None:
            self._cap.note(carry)

    def mark(self, name):
        """Stage boundary marker: with `self.marks` set to a list, records (name, HIP event on the current stream)."""
        marks = getattr(self, "marks", None)
        if marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))

    def _stage(self, name):
        self.mark("bwd:" + name)
        if getattr(self, "_side_active", False):          # join the LM stage's weight-gradient stream
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_active = False
            self._head_ev = None
            ops.GEMM_CONCURRENT = False
        if getattr(self, "_cs_pending", False):           # column sums launched beside a grouped weight gradient (_wg_flush)
            torch.cuda.current_stream().wait_stream(self._cs_stream)
            self._cs_pending = False
        if self.folds is not None:
            self.folds.flush()          # the stage's bias / LayerNorm gradients are complete before it is reported
        if self.stage_cb is not None or self._cap is not None:
            sr = getattr(self, "stage_ranges", None)
            if sr:
                self._zero_fresh_within(sr.get(name))
            if self._cap is not None:
                self._seg("stage:" + name)          # (the replay calls stage_cb between the graphs)
            else:
                self.stage_cb(name)

    # ------------------------------------------------------------------ gradient zeroing (round 4)
    # `optimizer.zero_grad()` + accumulate-into-zeros costs a 942-MB fill per step and a read of every weight gradient's old
    # (zero) value by the pass that writes it.  Instead the FIRST weight-gradient write of a step into a range of the flat
    # buffer STORES (`_gacc`): ranges that were written that way (learned from the steps themselves: `_gstore`) are left out of
    # the step's zeroing, everything else - biases, norm parameters, embedding / position tables, conv kernels, alignment gaps,
    # frozen tensors - is zeroed by ONE launch over a device table of (offset, count) rows (ops.zero_ranges).  A store range
    # nobody wrote in a step (LayerDrop, a frozen layer) is zeroed when backward ends; after 8 such steps it leaves the store set.
    # SMX_LAZY_ZERO=0: the plain fill.  Results are bit-identical either way (0 + x == x).
    def begin_grads(self, zero=True, lazy=True):
        st = self.st
        self._gfresh = {}
        self._gseen = set()
        self._gzeroed = set()
        if not zero:
            return
        if not lazy or os.environ.get("SMX_LAZY_ZERO") == "0" or st.grad.device.type != "cuda":
            st.grad.zero_()
            return
        store = getattr(self, "_gstore", None)
        if store is None:
            store = self._gstore = {}            # offset -> [numel, consecutive steps without a write]
            self._gplan = None
        self._ensure_gplan()
        tab, n = self._gplan
        if n:
            ops.zero_ranges(st.grad, tab, n)
        self._gfresh = {off: v[0] for off, v in store.items()}

    def _ensure_gplan(self):
        """The device table of the step's zeroing launch: the complement of the store ranges in [0, total), cut into rows of
        <= 65536 elements (rebuilt when the store set changed; graphs.StepGraphs builds it ahead of its capture)."""
        st = self.st
        store = getattr(self, "_gstore", None)
        if store is None:
            store = self._gstore = {}
            self._gplan = None
        if getattr(self, "_gplan", None) is not None:
            return
        if self._cap is not None:
            raise ops.CaptureAbort("the gradient-zeroing plan changed inside the capture pass")
        rows, pos = [], 0
        for off in sorted(store):
            if off > pos:
                rows.append((pos, off - pos))
            pos = max(pos, off + store[off][0])
        if pos < st.total:
            rows.append((pos, st.total - pos))
        tab = []
        for a, n in rows:
            for c in range(0, n, 65536):
                tab.append((a + c, min(65536, n - c)))
        self._gplan = (torch.tensor(tab, dtype=torch.int64).to(self.dev) if tab else None, len(tab))

    def _gacc(self, out):
        """-> True when a weight-gradient write into `out` must ADD to what is there, False when it is the step's first write
        into a range the step did not zero (it then stores).  Also learns the ranges (see begin_grads)."""
        fresh = getattr(self, "_gfresh", None)
        if fresh is None or out.dtype != torch.float32:
            return True
        off = (out.data_ptr() - self.st.grad.data_ptr()) // 4
        if not (0 <= off < self.st.total):
            return True                           # not a slice of the flat gradient (conv kernels' tap-major scratch)
        n = out.numel()
        self._gseen.add(off)
        if fresh.get(off) == n:
            del fresh[off]
            return False
        store = getattr(self, "_gstore", None)
        if store is None:
            return True
        known = store.get(off)
        if known is not None and known[0] == n and off not in fresh:
            return True                           # a later write of the step into a range its first write stored
        # slow path (first steps, or a write that does not match what was learned): a still-unwritten store range under this
        # write holds last step's values - zero it now and retire it; a range never seen before is learned for the next step
        hit = [o for o, m in fresh.items() if o < off + n and off < o + m]
        for o in hit:
            self.st.grad[o:o + fresh[o]].zero_()
            del fresh[o]
            store.pop(o, None)
            self._gplan = None
        if not hit and not any(o < off + n and off < o + v[0] for o, v in store.items()):
            store[off] = [n, 0]                   # zeroed this step (not in the plan yet); stores from the next step on
            self._gplan = None
        return True

    def _zero_fresh_within(self, ranges):
        """Zero the store ranges inside `ranges` [(a, b)] that nothing has written this step (a stage about to be reported
        to the gradient reducer: a dropped layer must contribute zeros, not last step's gradient)."""
        fresh = getattr(self, "_gfresh", None)
        if not fresh or not ranges:
            return
        # overlap, not containment: dist.stage_ranges cuts a stage's span into chunks mid-tensor, and a tensor that straddles a cut
        # must still contribute zeros (not last step's gradient) to BOTH chunks' collectives
        for off in [o for o, m in fresh.items() if any(o < b and a < o + m for a, b in ranges)]:
            self.st.grad[off:off + fresh[off]].zero_()
            self._gzeroed.add(off)
            del fresh[off]

    def end_grads(self):
        """Store ranges without a write this step hold last step's values: zero them (dropped / frozen layers)."""
        fresh, self._gfresh = getattr(self, "_gfresh", None), None
        store = getattr(self, "_gstore", None)
        if store is None:
            return
        for off, v in store.items():
            if off in self._gseen:
                v[1] = 0
        for off, n in (fresh or {}).items():
            self.st.grad[off:off + n].zero_()
        for off in list(fresh or {}) + list(self._gzeroed):
            if off in store:
                store[off][1] += 1
                if store[off][1] >= 8:
                    del store[off]
                    self._gplan = None

    # ------------------------------------------------------------------ small host -> device transfers
    def h2d(self, arr, dtype=torch.int32):
        """A small host array (SpecAugment rows, padded-frame rows, key lengths) to the device WITHOUT blocking the host: a
        `.to(device)` from pageable memory is a synchronous copy - the host stands still until the stream has drained (measured
        round 4: ~7 ms per training step at config 2, and the host's lead over the GPU, which the launch-bound LM stages live
        on, is gone afterwards).  Staged through a ring of pinned buffers; a slot is reused only after its copy's event."""
        t = torch.as_tensor(arr).to(dtype).reshape(-1)
        if self.dev.type != "cuda" or os.environ.get("SMX_H2D_BLOCKING") == "1":
            return t.to(self.dev)
        n = t.numel()
        ring = self._persist.get("_h2d_ring")
        if ring is None:
            ring = self._persist["_h2d_ring"] = dict(slots=[None] * 8, ev=[None] * 8, i=0)
        i = ring["i"]
        ring["i"] = (i + 1) % len(ring["slots"])
        buf = ring["slots"][i]
        nbytes = n * t.element_size()
        if ring["ev"][i] is not None:
            ring["ev"][i].synchronize()                  # (eight transfers old: long complete)
        if buf is None or buf.numel() < nbytes:
            buf = ring["slots"][i] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
        host = buf[:nbytes].view(dtype)
        host.copy_(t)
        out = torch.empty(n, dtype=dtype, device=self.dev)
        out.copy_(host, non_blocking=True)
        ring["ev"][i] = torch.cuda.Event()
        ring["ev"][i].record()
        return out

    # ------------------------------------------------------------------ buffers / parameter access
    def new(self, *shape, dt=None):
        return torch.empty(shape, dtype=dt or self.tdt, device=self.dev)

    def zeros(self, *shape, dt=None):
        return torch.zeros(shape, dtype=dt or self.tdt, device=self.dev)

    def persist_zeros(self, key, *shape, dt=None):
        """Zero-initialised buffer that lives across steps (kernels only ever write its interior rows)."""
        t = self._persist.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != (dt or self.tdt):
            t = self.zeros(*shape, dt=dt)
            self._persist[key] = t
        return t

    def workspace(self, key, numel, dt):
        """Grow-only scratch buffer that lives across steps (contents undefined)."""
        t = self._persist.get(key)
        if t is None or t.numel() < numel or t.dtype != dt:
            t = torch.empty(max(numel, 1), dtype=dt, device=self.dev)
            self._persist[key] = t
        return t

    def W(self, n): return self.st.w(n)
    def P(self, n): return self.st.p32(n)
    def G(self, n): return self.st.g(n)
    def has(self, n): return n in self.st.offsets
    def tr(self, *names): return any(self.st.requires_grad(n) for n in names)

    def _split(self, Mo, No, Kred):
        tiles = ((Mo + 127) // 128) * ((No + 127) // 128)
        ksteps = (Kred + 63) // 64
        # 4 workgroups/CU x 256 CUs are resident at once: aim for ~1 full wave of workgroups, keep every split
        # non-empty and at least 8 K-steps long (slab traffic grows with the split count)
        if ksteps <= 16 and tiles >= 96:
            return 1      # short reductions over >= 96 tiles (decoder weight gradients): the slab pass costs more than it returns
        want = int(max(1, min(1024 // max(tiles, 1), ksteps // 8, 32)))       # floor: one resident wave, no tail round
        per = (ksteps + want - 1) // want
        return (ksteps + per - 1) // per

    # ------------------------------------------------------------------ linear building blocks
    def _fsplit(self, Mo, No, Kred, kw):
        """Split factor for a forward / data-gradient GEMM whose output has fewer tiles than the chip has CUs (decoder
        side, LM-head gradient): a lone workgroup per CU is bound by that CU's fetch rate, so spread K over idle CUs."""
        if self.dt != BF16 or kw.get("nbatch", 1) > 1 or kw.get("atomic") or kw.get("split_k", 1) > 1:
            return 1
        tiles = ((Mo + 127) // 128) * ((No + 127) // 128)
        ksteps = (Kred + 63) // 64
        # measured under rocprofv3 (tools/gpu_small_gemm.py trace): a 128x128 launch costs ~8 us + 0.8 us per K step, the slab
        # pass ~7 us more - up to 16 K steps (K <= 1024) one launch wins, beyond that ~8 K steps per split
        # round 4: 257-512 tiles of 128 x 128 (the text encoder's 7 968 x 768 outputs: 126 items of the 192-row kernel on 256 CUs).
        # Two K slices pay from ~96 K steps on (the cross-attention K/V gradients' 7 968 x 768 x 9 216 data gradient: 260 -> 153 us
        # + a 15-us slab epilogue); at K = 2304 / 3072 the GEMMs gain 12-18 us and the epilogue launch gives it back
        # (SMX_FSPLIT_MID=1 splits those too: 32.02 vs 31.96 ms, profiles/r04_ab_runs.txt)
        if 256 < tiles <= 512 and Mo >= 4096 and (ksteps >= 96 or (ksteps >= 32 and os.environ.get("SMX_FSPLIT_MID") == "1")):
            return 2
        if tiles > 256 or ksteps <= 16:
            return 1
        want = int(min(1024 // tiles, ksteps // 8, 24))
        if want < 2:
            return 1
        per = (ksteps + want - 1) // want
        return (ksteps + per - 1) // per

    def _gemm(self, a, b, c, M, N, K, **kw):
        split = self._fsplit(M, N, K, kw)
        if split > 1 and kw.get("cv") is None:
            slabs = self.workspace("fwd_slabs", split * M * ((N + 7) // 8 * 8), torch.float32)
            ops.gemm_splitk(a, b, c, M, N, K, self.dt, split, slabs, **kw)
        else:
            ops.gemm(a, b, c, M, N, K, self.dt, **kw)

    def lin(self, x, w, b, M, N, K, y=None, act=ACT_NONE, resid=None, aux_out=None, av=None, cv=None, ev=None,
            alpha=1.0, out_f32=False, **kw):
        if y is None:
            y = self.new(M, N, dt=torch.float32 if out_f32 else None)
        self._gemm(x, w, y, M, N, K, av=av, cv=cv, ev=ev, bias=b, resid=resid, aux_out=aux_out, act=act,
                   alpha=alpha, out_f32=out_f32, **kw)
        return y

    def dgrad(self, dy, w, dx, M, N, K, resid=None, aux_in=None, act=ACT_NONE, av=None, bv=None, cv=None, ev=None,
              alpha=1.0, **kw):
        """dx[M,K] = dy[M,N] @ w[N,K].  bf16 path, plain weights: the GEMM reads the K-contiguous copy w^T [K, N] (`_wt`), so that both
        operands take the 16-byte fragment reads; otherwise w is read rows-contiguous in place."""
        # (the activation-gradient class - FFN2's data gradient - keeps the in-place read: its saved-derivative epilogues are instantiated for
        # the rows-contiguous layout only)
        # (and only where the GEMM is long enough for the read to matter: at the decoder's M = 1 024 the copy costs what it saves)
        wt = self._wt(w, N, K) if (bv is None and aux_in is None and M >= WT_MIN_M) else None
        if wt is not None:
            self._gemm(dy, wt, dx, M, K, N, av=av, bv=view(N), cv=cv, ev=ev, resid=resid, aux_in=aux_in, act=act, alpha=alpha, **kw)
            return dx
        self._gemm(dy, w, dx, M, K, N, b_rc=True, av=av, bv=bv if bv is not None else view(K), cv=cv, ev=ev,
                   resid=resid, aux_in=aux_in, act=act, alpha=alpha, **kw)
        return dx

    # ---- K-contiguous copies of the Linear weights for the data gradients (round 6) ----
    # One persistent [K, N] tensor per weight view the data gradients have met, re-made by ONE batched transpose launch (per 64 weights)
    # at the first data gradient after the compute copies changed: trainable weights after every optimizer step (FlatStore.wver),
    # frozen ones only after a re-cast of the masters (FlatStore.hard_ver).  SMX_DGRAD_WT=0: off (the rows-contiguous reads of rounds 1-5).
    def _wt(self, w, N, K):
        if not WT_MODE or self.dt != BF16 or (N & 7) or (K & 7) or w.dtype != torch.bfloat16 or not w.is_contiguous():
            return None
        st = self.st
        key = (w.data_ptr(), N, K)
        tab = self.__dict__.setdefault("_wt_tab", {})
        ent = tab.get(key)
        if ent is None:
            off = (w.data_ptr() - st.shadow.data_ptr()) // 2
            if st.shadow is st.master or off < 0 or off + N * K > st.total:
                return None          # not a view of the flat store (packed / temporary operands): read in place
            if ops.CAPTURING:
                raise ops.CaptureAbort("a data gradient met a weight without a transposed copy inside a stream capture")
            name = st.name_at(off)
            ent = tab[key] = {"src": w.view(N, K), "dst": torch.empty(K, N, dtype=w.dtype, device=w.device), "ver": -1,
                              "trainable": name is None or st.requires_grad(name)}
        want = st.wver if ent["trainable"] else st.hard_ver
        if ent["ver"] != want:
            jobs = []
            for e in tab.values():          # every stale copy in the same launch (the first data gradient of a backward pays for all)
                wv = st.wver if e["trainable"] else st.hard_ver
                if e["ver"] != wv:
                    jobs.append((e["src"], e["dst"]))
                    e["ver"] = wv
            ops.transpose_many(jobs)
        return ent["dst"]

    def _slab_key(self):
        """Grow-only slab workspace, one per stream: a buffer that grows is replaced, and the replaced tensor must not go back
        to the allocator of one stream while kernels queued on the other still write it."""
        side = getattr(self, "_side", None)
        on_side = side is not None and torch.cuda.current_stream() == side
        return "wgrad_slabs_side" if on_side else "wgrad_slabs"

    def _wgrad_gemm(self, dy, x, out, No, Ko, Kred, av, bv, alpha, accumulate, cv=None, **kw):
        """out[No,Ko] (+)= dy^T x over Kred rows.  Two candidate launches - the 128x128 kernel with its K split and the
        ping-pong kernel with a split sized for one round of 256x256 items - timed on first use per shape (both write
        slabs, so the trial runs are side-effect free); the slab sum then lands in `out`."""
        n = No * Ko
        if accumulate:
            accumulate = self._gacc(out)
        cands = [(1, self._split(No, Ko, Kred))]
        if self.dt == BF16 and ops.pp_allowed() and cv is None and not getattr(self, "_side_active", False):
            sp = ops.pp_split(No, Ko, Kred)
            if ((No + 255) // 256) * ((Ko + 255) // 256) * sp >= 96 and sp > 1:
                cands.append((8, sp))
                if ops.FR_MODE != "0" and not (av.rows_per_batch > 0 or bv.rows_per_batch > 0):
                    cands.append((12, sp))
        if len(cands) > 1 and ops.PP_MODE == "1":
            cands = cands[1:]
        if len(cands) > 1 and ops.FR_MODE == "1" and cands[-1][0] == 12:
            cands = cands[-1:]

        def run(mode, split):
            if split <= 1:
                ops.gemm(dy, x, out, No, Ko, Kred, self.dt, a_rc=True, b_rc=True, av=av, bv=bv, cv=cv, out_f32=True,
                         atomic=2 if accumulate else 0, alpha=alpha, tr_mode=mode, **kw)
                return
            slabs = self.workspace(self._slab_key(), split * n, torch.float32)
            ops.gemm(dy, x, slabs, No, Ko, Kred, self.dt, a_rc=True, b_rc=True, av=av, bv=bv, cv=cv, out_f32=True, atomic=0,
                     split_k=split, split_stride=n, alpha=alpha, tr_mode=mode, **kw)
            ops.reduce_slabs(slabs, split, n, n, out, accumulate=accumulate)

        choice = cands[0]
        if len(cands) > 1:
            # the pick is stored as the chosen (kernel, K split) itself: the candidate list depends on SMX_GEMM_PP / SMX_GEMM_FR /
            # the stream, so a positional index from a picks file written under other settings could name another kernel
            key = ("wgrad", ops.pp_cus(), No, Ko, Kred, av.rows_per_batch > 0, bv.rows_per_batch > 0, tuple(sorted(kw)))
            choice = ops._tuned_get(key)
            if isinstance(choice, list):
                choice = tuple(choice)
            if choice not in cands:                          # nothing stored, or a pick that is not on offer here: tune
                choice = cands[0]
                if all(sp > 1 for _, sp in cands):          # slab launches only: re-running them changes nothing
                    def once(mode, sp):
                        slabs = self.workspace(self._slab_key(), sp * n, torch.float32)

                        def f():
                            ops.gemm(dy, x, slabs, No, Ko, Kred, self.dt, a_rc=True, b_rc=True, av=av, bv=bv, cv=cv, out_f32=True,
                                     atomic=0, split_k=sp, split_stride=n, alpha=alpha, tr_mode=mode, **kw)
                            ops._reduce_slabs(slabs, sp, n, n, slabs, False)
                        return f
                    self.workspace(self._slab_key(), max(sp for _, sp in cands) * n, torch.float32)      # (grown once, before timing)
                    ts = ops.measure_candidates({c: once(*c) for c in cands})
                    choice = min(ts, key=lambda c: ts[c] * (1.0 if c == cands[0] else 1.03))      # ties go to the 128x128 kernel
                    ops.TUNE_LIVE_KEYS.append(key)
                    if ops.TUNE_LOG is not None:
                        ops.TUNE_LOG.append((key, ts.get(cands[0]), ts.get(cands[1]), choice, None, None,
                                             ts.get(cands[2]) if len(cands) > 2 else None))
                ops._tuned_set(key, choice)
        run(*choice)

    def wgrad(self, dy, x, gw, M, N, K, dyv=None, xv=None, alpha=1.0, gb=None, dy_ld=None, side_ok=True, **kw):
        """gw[N,K] += dy[M,N]^T @ x[M,K];  gb[N] += colsum(dy).  The reduction over M is split across workgroups
        when the output has too few tiles to fill the chip: each split stores its fp32 partial slab and one
        streaming pass sums the slabs into the gradient (no atomics: they serialise in L2)."""
        av = dyv if dyv is not None else view(N)
        bv = xv if xv is not None else view(K)
        side = self._side if (side_ok and getattr(self, "_side_active", False)) else None
        grp = getattr(self, "_wg_group", None)
        rows = getattr(self, "_wg_rows", None)           # LM stage: only problems over exactly this many rows are grouped
        if (grp is not None and (side is None or rows is not None) and not kw and self.dt == BF16 and ops.pp_allowed()
                and (M >= 4096 if rows is None else M == rows) and len(grp) < 8
                and av.rows_per_batch <= 0 and bv.rows_per_batch <= 0 and not ((N | K | av.ld | bv.ld | av.off | bv.off) & 7)):
            # deferred: the layer's weight gradients go out as ONE grouped launch when the layer's backward ends (_wg_flush);
            # dy and x are never written again (every backward output is a fresh tensor, saved activations are read-only)
            grp.append((dy, x, gw, N, K, M, av, bv, alpha))
            csd = getattr(self, "_cs_defer", None)
            if gb is not None and side is None and csd is not None:
                # (speech encoder layers: the bias-gradient column sums go out beside the layer's grouped launch, _wg_flush)
                csd.append((dy, gb, M, N, dy_ld or N, alpha))
            elif gb is not None:
                if side is not None:                     # (LM stage: the column sums stay on the second stream)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        ops.colsum(dy, gb, M, N, dy_ld or N, self.dt, alpha, folds=self.folds)
                    dy.record_stream(side)
                else:
                    ops.colsum(dy, gb, M, N, dy_ld or N, self.dt, alpha, folds=self.folds)
            return
        if side is None:
            self._wgrad_gemm(dy, x, gw, N, K, M, av, bv, alpha, True, **kw)
            if gb is not None:
                ops.colsum(dy, gb, M, N, dy_ld or N, self.dt, alpha, folds=self.folds)
            return
        dq = getattr(self, "_wg_defer", None)
        if dq is not None:
            # deferred to the end of the layer (lm_bwd -> _wg_defer_flush): ONE fork to the second stream per layer instead of one
            # per weight gradient.  In a captured step every fork moves the main chain to another hardware queue, and a
            # cross-queue dependency costs ~8 us of idle GPU (kernel trace, round 5)
            dq.append((dy, x, gw, N, K, M, av, bv, alpha, gb, dy_ld, kw))
            return
        # LM stage (SMX_LM_WGRAD_STREAM=0: off): a weight gradient needs nothing that is still being computed, so it runs on a
        # second stream beside the small-grid kernels of the LM's backward (decoder: 48 workgroups per launch, text encoder:
        # 1.5 per CU) and is joined at the end of the stage.  What keeps that safe: dy and x are never written again (every
        # backward output is a fresh tensor; saved activations are read-only) and record_stream keeps the allocator from
        # recycling them early; the slab workspace is used on that stream only until the join; destinations that main-stream
        # kernels also add to (the tied embedding: side_ok=False) stay on the main stream; bias-gradient partial rows are
        # folded after the join.  The 128x128 kernel is used there (a one-workgroup-per-CU kernel would stall the main stream).
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        with torch.cuda.stream(side):
            self._wgrad_gemm(dy, x, gw, N, K, M, av, bv, alpha, True, **kw)
            if gb is not None:
                ops.colsum(dy, gb, M, N, dy_ld or N, self.dt, alpha, folds=self.folds)
        dy.record_stream(side)
        x.record_stream(side)

    def _wg_defer_begin(self):
        on = getattr(self, "_side_active", False) and (self._cap is not None or os.environ.get("SMX_LM_WGRAD_DEFER", "auto") == "1") \
            and os.environ.get("SMX_LM_WGRAD_DEFER", "auto") != "0"
        self._wg_defer = [] if on else None

    def _wg_defer_flush(self):
        dq, self._wg_defer = getattr(self, "_wg_defer", None), None
        if not dq:
            return
        side = self._side
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for dy, x, gw, N, K, M, av, bv, alpha, gb, dy_ld, kw in dq:
                self._wgrad_gemm(dy, x, gw, N, K, M, av, bv, alpha, True, **kw)
                if gb is not None:
                    ops.colsum(dy, gb, M, N, dy_ld or N, self.dt, alpha, folds=self.folds)
        for dy, x, *_ in dq:
            dy.record_stream(side)
            x.record_stream(side)

    # Grouped weight gradients (SMX_WGRAD_GROUP=0: off).  Launched one by one, each of an encoder layer's four weight
    # gradients has 9-36 output tiles of 256 x 256 and needs 7 K slices to fill the chip: 7 fp32 slabs written and read back
    # per weight.  Together they have 108 tiles, so ONE launch over the concatenated work lists fills the chip with 2 slices:
    # 3.5x less slab traffic, one launch (and one tail) instead of four, and K loops of 125 tiles instead of 36 per item.
    def _wg_side_stream(self):
        # measured round 3 (same box, seeded, 20 steps): 32.74 / 32.82 vs 33.10 / 33.08 ms in one pair of runs, 32.85 / 32.77 vs
        # 32.77 / 32.65 in the next - no reliable gain, so it stays an opt-in switch (gradients identical: tools/gpu_side_check.py)
        if os.environ.get("SMX_WGRAD_SIDE", "0") != "1" or self.st.device.type != "cuda" or self.dt != BF16:
            return None
        if getattr(self, "_wg_side", None) is None:
            self._wg_side = torch.cuda.Stream()
        return self._wg_side

    def _join_wg(self, stage):
        """The grouped weight gradients of `stage` are complete: join their stream, then report the stage."""
        torch.cuda.current_stream().wait_stream(self._wg_side)
        self._stage(stage)

    def _wg_begin(self, rows=None):
        """rows: group only the weight gradients over exactly `rows` rows, also while the LM stage's second stream is active
        (round 4: a decoder layer's seven M = B x L weight gradients - 144 output tiles of 256 x 256, 16 K tiles deep - as one
        launch on that stream instead of seven split-K launches + their slab reductions)."""
        # (the LM-stage form measured 33.10 vs 32.97 ms: a 144-CU persistent launch beside the decoder's small-grid chain takes its
        # CUs away - off unless SMX_LM_WGRAD_GROUP=1)
        on = _WGRAD_GROUP and self.dt == BF16 and (rows is None or os.environ.get("SMX_LM_WGRAD_GROUP", "0") == "1")
        self._wg_group = [] if on else None
        self._wg_rows = rows if on else None
        # Column sums beside the grouped launch (SMX_COLSUM_SIDE=0: off): a layer's grouped weight gradient runs 216 work items on 256 CUs for
        # ~250 us, and the bias gradients of its QKV and FFN1 Linears are two HBM-bound passes over 73 + 98 MB that used to sit in the
        # data-gradient chain (17 + 15 us per layer).  They are collected here and launched on a stream of their own right before the grouped
        # launch (_wg_flush), on the 40 CUs it leaves idle; _stage joins that stream before the stage's partial rows are folded.
        self._cs_defer = [] if (on and rows is None and _COLSUM_SIDE and self.st.device.type == "cuda") else None

    def _wg_flush(self, side=None):
        """side: a HIP stream - the grouped launch (and its slab reduction) go there, behind everything the current stream has
        enqueued so far; the caller joins it before it reports the stage (speech_bwd: one layer later, so that the launch -
        216 work items on 256 CUs - runs beside the next layer's data-gradient chain, which takes the 40 CUs it leaves idle)."""
        grp, self._wg_group = getattr(self, "_wg_group", None), None
        self._wg_rows = None
        cs, self._cs_defer = getattr(self, "_cs_defer", None), None
        if cs:
            beside = bool(grp) and len(grp) > 1 and side is None
            if beside:
                if getattr(self, "_cs_stream", None) is None:
                    self._cs_stream = torch.cuda.Stream()
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self._cs_stream.wait_event(ev)
            with (torch.cuda.stream(self._cs_stream) if beside else contextlib.nullcontext()):
                for dy, gb, M, N, ld, alpha in cs:
                    ops.colsum(dy, gb, M, N, ld, self.dt, alpha, folds=self.folds)
            if beside:
                for dy, *_ in cs:
                    dy.record_stream(self._cs_stream)
                self._cs_pending = True
        if not grp:
            return False
        if side is not None and len(grp) > 1:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            self._wg_group = grp
            with torch.cuda.stream(side):
                self._wg_flush()
            for dy, x, *_ in grp:                 # the allocator must not hand their memory out while the side stream reads it
                dy.record_stream(side)
                x.record_stream(side)
            return True
        if len(grp) == 1:
            dy, x, gw, N, K, M, av, bv, alpha = grp[0]
            self._wgrad_gemm(dy, x, gw, N, K, M, av, bv, alpha, True)
            return
        tiles = sum(((N + 255) // 256) * ((K + 255) // 256) for _, _, _, N, K, _, _, _, _ in grp)
        ksteps = min((M + 63) // 64 for _, _, _, _, _, M, _, _, _ in grp)
        want = max(1, min((ops.pp_cus() or 256) // max(tiles, 1), ksteps // 4))
        total = sum(N * K for _, _, _, N, K, _, _, _, _ in grp)
        probs, outs = [], []
        slabs = self.workspace("wg_group_" + self._slab_key(), (want + 1) * total, torch.float32) if want > 1 else None
        off = 0
        for dy, x, gw, N, K, M, av, bv, alpha in grp:
            n = N * K
            kst = (M + 63) // 64
            per = (kst + want - 1) // want
            sp = (kst + per - 1) // per              # every K slice owns at least one K tile
            kw = dict(a_rc=True, b_rc=True, av=av, bv=bv, out_f32=True, alpha=alpha)
            acc = self._gacc(gw)
            if sp > 1:
                dst = slabs[off:off + sp * n]
                kw.update(atomic=0, split_k=sp, split_stride=n)
                outs.append((dst, sp, n, gw, acc))
                off += sp * n
            else:
                dst = gw
                kw.update(atomic=2 if acc else 0)
            probs.append((dy, x, dst, N, K, M, kw))
        mode = 8
        if ops.FR_MODE == "1":
            mode = 12
        elif ops.FR_MODE != "0":
            key = ("wgrad_group", ops.pp_cus(), tuple((N, K, M) for _, _, _, N, K, M, _, _, _ in grp))
            mode = ops._tuned_get(key)
            if mode is None:
                mode = 8
                if len(outs) == len(probs):            # slab launches only: re-running them changes nothing
                    ts = ops.measure_candidates({m: (lambda m=m: ops.gemm_group(probs, self.dt, mode=m)) for m in (8, 12)})
                    mode = 12 if ts[12] < ts[8] else 8
                    ops.TUNE_LIVE_KEYS.append(key)
                    if ops.TUNE_LOG is not None:
                        ops.TUNE_LOG.append((key, None, ts[8], mode, None, None, ts[12]))
                ops._tuned_set(key, mode)
        ops.gemm_group(probs, self.dt, mode=mode)
        for flag in (True, False):
            sel = [(dst, sp, n, gw) for dst, sp, n, gw, acc in outs if acc == flag]
            if sel:
                ops.reduce_slabs_many(sel, accumulate=flag)

    def ln_fwd(self, x, wname, bname, M, D, eps, rms=False, act=ACT_NONE, pos=None, pos_period=0, pos_offset=0,
               want_sum=False, drop=None):
        y = self.new(M, D)
        mean = None if rms else self.new(M, dt=torch.float32)
        rstd = self.new(M, dt=torch.float32)
        xs = self.new(M, D) if want_sum else None
        ops.norm_fwd(x, y, self.P(wname), self.P(bname) if bname else None, mean, rstd, M, D, self.dt, eps=eps, rms=rms,
                     act=act, pos=pos, pos_period=pos_period, pos_offset=pos_offset, xsum_out=xs, drop=drop)
        return y, (xs if want_sum else x, mean, rstd, drop)

    def ln_bwd(self, dy, saved, wname, bname, M, D, rms=False, act=ACT_NONE, dres=None, dpos=None, pos_period=0,
               pos_offset=0, dx=None, fuse=None):
        """fuse = dict(drop=(p, seed), gb=bias gradient) of the Linear whose dropped output fed this norm (post-LN layers): the
        kernel then also writes dx * mask into fuse["out"] and queues its column sums into gb (see ops.norm_bwd)."""
        x, mean, rstd, drop = saved
        dx = dx if dx is not None else self.new(M, D)
        train = self.tr(wname)
        extra = {}
        if (fuse is not None and train and self.folds is not None and act == ACT_NONE and not (D & 7)
                and os.environ.get("SMX_FUSE_LN_DROPCOL") != "0" and os.environ.get("SMX_NORM_FUSED") != "0"):
            fuse["out"] = self.new(M, D)
            extra = dict(drop2=fuse["drop"], dx_drop=fuse["out"], gb2=fuse["gb"])
        ops.norm_bwd(dy, x, dx, self.P(wname), self.P(bname) if bname else None, mean, rstd,
                     self.G(wname) if train else None, self.G(bname) if (bname and train) else None, M, D, self.dt,
                     rms=rms, act=act, dres=dres, dpos=dpos, pos_period=pos_period, pos_offset=pos_offset, drop=drop,
                     folds=self.folds, **extra)
        return dx

    def _fuse_site(self, drop, wname, bname, N):
        """-> the `fuse` argument of ln_bwd for a dropped Linear (weight wname, bias bname, N outputs), or None when its
        masked gradient / bias gradient cannot ride in the norm's backward (no dropout, no trainable bias)."""
        if drop is None or not bname or not self.tr(wname) or (N & 7):
            return None
        return dict(drop=drop, gb=self.G(bname))

    # ------------------------------------------------------------------ attention block (self or cross)
    def attn_fwd(self, x, kvsrc, B, Tq, Tk, d, H, names, causal, scale, bias=None, drop=None, klen=None, masks=None):
        """names: dict(q,k,v,o -> (weight, bias|None)).  klen: int32 [B] on the device - keys >= klen[b] are padding (stays in the
        descriptor: backward masks the same keys).  Returns (o [B*Tq, d], saved)."""
        Mq, Mk = B * Tq, B * Tk
        hd = d // H
        self_attn = kvsrc is None
        qn, kn, vn = names["q"], names["k"], names["v"]
        if self_attn:
            wqkv = self.st.cat([qn[0], kn[0], vn[0]])
            bqkv = self.st.cat([qn[1], kn[1], vn[1]], "p32") if qn[1] else None
            qkv = self.lin(x, wqkv, bqkv, Mq, 3 * d, d)
            desc = ops.AttnDesc(B, H, Tq, Tk, hd, causal, scale, bias, drop=drop, klen=klen, masks=masks)
            desc.set("Q", qkv, 0, Tq * 3 * d, 3 * d)
            desc.set("K", qkv, d, Tk * 3 * d, 3 * d)
            desc.set("V", qkv, 2 * d, Tk * 3 * d, 3 * d)
            kv = None
        else:
            qkv = self.lin(x, self.W(qn[0]), self.P(qn[1]) if qn[1] else None, Mq, d, d)
            wkv = self.st.cat([kn[0], vn[0]])
            bkv = self.st.cat([kn[1], vn[1]], "p32") if kn[1] else None
            kv = self.lin(kvsrc, wkv, bkv, Mk, 2 * d, d)
            desc = ops.AttnDesc(B, H, Tq, Tk, hd, causal, scale, bias, drop=drop, klen=klen)
            desc.set("Q", qkv, 0, Tq * d, d)
            desc.set("K", kv, 0, Tk * 2 * d, 2 * d)
            desc.set("V", kv, d, Tk * 2 * d, 2 * d)
        o = self.new(Mq, d)
        lse = self.new(B * H * Tq, dt=torch.float32)
        desc.set("O", o, 0, Tq * d, d)
        ops.attention_fwd(desc, lse, self.dt)
        return o, dict(x=x, kvsrc=kvsrc, qkv=qkv, kv=kv, o=o, lse=lse, desc=desc, dims=(B, Tq, Tk, d, H))

    def attn_bwd(self, do, sv, names, dx_resid=None, dkv_accum=None, dx=None, dbias=None):
        """do: grad wrt attention output o (before out_proj).  Returns dx (grad wrt x, + dx_resid).
        Cross attention: grad wrt kvsrc is accumulated into dkv_accum [B*Tk, d] (must be pre-initialised).
        dbias [H, Tq, Tk] fp32: += the batch-summed score gradient (T5 relative-position bias)."""
        B, Tq, Tk, d, H = sv["dims"]
        Mq, Mk = B * Tq, B * Tk
        desc = sv["desc"]
        qn, kn, vn = names["q"], names["k"], names["v"]
        delta = self.new(B * H * Tq, dt=torch.float32)
        desc.set("dO", do, 0, Tq * d, d)
        if sv["kvsrc"] is None:
            dqkv = self.new(Mq, 3 * d)
            desc.set("dQ", dqkv, 0, Tq * 3 * d, 3 * d)
            desc.set("dK", dqkv, d, Tk * 3 * d, 3 * d)
            desc.set("dV", dqkv, 2 * d, Tk * 3 * d, 3 * d)
            ops.attention_bwd(desc, sv["lse"], delta, self.dt, dbias=dbias)
            wn = [qn[0], kn[0], vn[0]]
            if self.tr(*wn):
                self.wgrad(dqkv, sv["x"], self.st.cat(wn, "g"), Mq, 3 * d, d,
                           gb=self.st.cat([qn[1], kn[1], vn[1]], "g") if qn[1] else None)
            dx = dx if dx is not None else self.new(Mq, d)
            self.dgrad(dqkv, self.st.cat(wn), dx, Mq, 3 * d, d, resid=dx_resid)
            return dx
        dq = self.new(Mq, d)
        desc.set("dQ", dq, 0, Tq * d, d)
        cat = dkv_accum[2] if (dkv_accum is not None and len(dkv_accum) > 2) else None
        if cat is not None:
            # all decoder layers' dK | dV side by side in ONE [Mk, L * 2d] buffer: the gradient wrt the encoder output is then a
            # single GEMM over K = L * 2d after the last layer (lm_bwd) instead of L accumulating launches of 378 tiles each
            buf, li, ldc = cat["buf"], cat["layer"], cat["ld"]
            dkv = buf.view(-1)[li * 2 * d:]
            desc.set("dK", buf, li * 2 * d, Tk * ldc, ldc)
            desc.set("dV", buf, li * 2 * d + d, Tk * ldc, ldc)
        else:
            ldc = 2 * d
            dkv = self.new(Mk, 2 * d)
            desc.set("dK", dkv, 0, Tk * 2 * d, 2 * d)
            desc.set("dV", dkv, d, Tk * 2 * d, 2 * d)
        ops.attention_bwd(desc, sv["lse"], delta, self.dt)
        if self.tr(qn[0]):
            self.wgrad(dq, sv["x"], self.G(qn[0]), Mq, d, d, gb=self.G(qn[1]) if qn[1] else None)
        if self.tr(kn[0], vn[0]):
            self.wgrad(dkv, sv["kvsrc"], self.st.cat([kn[0], vn[0]], "g"), Mk, 2 * d, d, dyv=view(ldc), dy_ld=ldc,
                       gb=self.st.cat([kn[1], vn[1]], "g") if kn[1] else None)
        dx = dx if dx is not None else self.new(Mq, d)
        self.dgrad(dq, self.W(qn[0]), dx, Mq, d, d, resid=dx_resid)
        if dkv_accum is not None and cat is None:
            first = dkv_accum[1]
            self.dgrad(dkv, self.st.cat([kn[0], vn[0]]), dkv_accum[0], Mk, 2 * d, d, resid=None if first else dkv_accum[0])
        return dx

    # ------------------------------------------------------------------ transformer layers
    def _pad_ld(self, F):
        """Leading dimension of the FFN's [M, F] intermediates.  A row stride that is a multiple of 8 KB (F = 4096 in bf16: the
        *large* backbones) puts the same column of consecutive rows on ONE HBM channel group: the weight-gradient GEMMs, which
        read those tensors rows-contiguous (a short run of every row per K step), measured 5x slower there.  64 elements of
        padding spread the rows over the channels; the GEMMs take the stride through their row views."""
        es = 2 if self.dt == BF16 else 4
        return F + 64 if (F * es) % 8192 == 0 and os.environ.get("SMX_PAD_FFN") != "0" else F

    def _ffn_fwd(self, h, M, d, F, n1, n2, act, resid, d_act=None, d_out=None):
        """resid + drop_out(fc2(drop_act(act(fc1(h)))))  - both dropouts run inside the GEMM epilogues."""
        Fp = self._pad_ld(F)
        pre = self.new(M, Fp)
        f = self.new(M, Fp)
        fv = view(Fp)
        # bf16: `pre` holds act'(fc1 out) * activation-dropout multiplier - the local derivative the backward GEMM multiplies
        # in - instead of the pre-activation (ops.ACT_SAVE_GRAD: no erf / exp and no dropout hash in the backward epilogue)
        if self.dt == BF16 and act != ACT_NONE and os.environ.get("SMX_SAVE_ACT_GRAD") != "0":
            act = act | ops.ACT_SAVE_GRAD
        self.lin(h, self.W(n1[0]), self.P(n1[1]) if n1[1] else None, M, F, d, y=f, act=act, aux_out=pre, drop=d_act,
                 cv=fv if Fp != F else None)
        y = self.lin(f, self.W(n2[0]), self.P(n2[1]) if n2[1] else None, M, d, F, resid=resid, drop=d_out,
                     av=fv if Fp != F else None)
        return y, (h, pre, f, d_act, d_out, act)

    def _ffn_bwd(self, dy, sv, M, d, F, n1, n2, act, dx_resid, dy_masked=None):
        """dy: grad wrt fc2 output.  Returns grad wrt h (+ dx_resid).  dy_masked: dy * fc2's output-dropout mask with the bias
        gradient already queued (ln_bwd's `fuse`)."""
        h, pre, f, d_act, d_out, act = sv            # (act as the forward used it: may carry ACT_SAVE_GRAD)
        Fp = self._pad_ld(F)
        fv = view(Fp) if Fp != F else None
        gb = self.G(n2[1]) if (n2[1] and self.tr(n2[0])) else None
        if dy_masked is not None:
            dy, fused = dy_masked, True
        else:
            dy, fused = self._dropped(dy, d_out, M * d, bias_grad=gb, N=d) if gb is not None else (self._dropped(dy, d_out, M * d), False)
        if self.tr(n2[0]):
            self.wgrad(dy, f, self.G(n2[0]), M, d, F, gb=None if fused else gb, xv=fv)
        dpre = self.new(M, Fp)
        self.dgrad(dy, self.W(n2[0]), dpre, M, d, F, aux_in=pre, act=act, drop=d_act, cv=fv)
        if self.tr(n1[0]):
            self.wgrad(dpre, h, self.G(n1[0]), M, F, d, gb=self.G(n1[1]) if n1[1] else None, dyv=fv, dy_ld=Fp)
        dh = self.new(M, d)
        self.dgrad(dpre, self.W(n1[0]), dh, M, F, d, resid=dx_resid, av=fv)
        return dh

    def layer_fwd(self, x, B, T, d, H, F, nm, pre_ln, act, eps, causal=False, scale=None, enc=None, Tk=None, rms=False,
                  bias=None, cross_bias=None, drop=None, klen=None, enc_klen=None, premask=None):
        """One transformer layer.  nm: dict with keys attn{q,k,v,o}, ln1, [xattn, lnx], fc1, fc2, ln2.
        drop = (hidden p, attention-probability p, activation p) in training mode, else None.  Sites (identical in
        TF:models/wav2vec2/modeling_wav2vec2.py:575-654, TF:models/bart/modeling_bart.py:260-475, T5 blocks): the
        attention probabilities, the out_proj output, the activation, the fc2 output."""
        M = B * T
        scale = scale if scale is not None else (d // H) ** -0.5
        sv = {}
        ph, pa, pf = drop if drop is not None else (0.0, 0.0, 0.0)
        sv["d_o"], sv["d_xo"] = self._dp(ph, "o"), (self._dp(ph, "xo") if enc is not None else None)
        d_act, d_out = self._dp(pf, "act"), self._dp(ph, "out")
        da, dxa = self._dp(pa, "attn"), (self._dp(pa, "xattn") if enc is not None else None)
        amask = None
        if premask is not None and da is not None and premask["p"] == da[0] and premask.get("key") == getattr(self, "_key", None):
            da, amask = (da[0], premask["seed"]), premask["masks"]      # the bit matrices generated beside the last optimizer step
        if not pre_ln:
            o, sv["a"] = self.attn_fwd(x, None, B, T, T, d, H, nm["attn"], causal, scale, bias, drop=da, klen=klen, masks=amask)
            s1 = self.lin(o, self.W(nm["attn"]["o"][0]), self._b(nm["attn"]["o"][1]), M, d, d, resid=x, drop=sv["d_o"])
            h, sv["ln1"] = self.ln_fwd(s1, nm["ln1"][0], nm["ln1"][1], M, d, eps)
            if enc is not None:
                o2, sv["x"] = self.attn_fwd(h, enc, B, T, Tk, d, H, nm["xattn"], False, scale, cross_bias, drop=dxa, klen=enc_klen)
                s2 = self.lin(o2, self.W(nm["xattn"]["o"][0]), self._b(nm["xattn"]["o"][1]), M, d, d, resid=h,
                              drop=sv["d_xo"])
                h, sv["lnx"] = self.ln_fwd(s2, nm["lnx"][0], nm["lnx"][1], M, d, eps)
            s3, sv["f"] = self._ffn_fwd(h, M, d, F, nm["fc1"], nm["fc2"], act, h, d_act, d_out)
            y, sv["ln2"] = self.ln_fwd(s3, nm["ln2"][0], nm["ln2"][1], M, d, eps)
        else:
            n1, sv["ln1"] = self.ln_fwd(x, nm["ln1"][0], nm["ln1"][1], M, d, eps, rms=rms)
            o, sv["a"] = self.attn_fwd(n1, None, B, T, T, d, H, nm["attn"], causal, scale, bias, drop=da, klen=klen, masks=amask)
            x1 = self.lin(o, self.W(nm["attn"]["o"][0]), self._b(nm["attn"]["o"][1]), M, d, d, resid=x, drop=sv["d_o"])
            if enc is not None:
                nx, sv["lnx"] = self.ln_fwd(x1, nm["lnx"][0], nm["lnx"][1], M, d, eps, rms=rms)
                o2, sv["x"] = self.attn_fwd(nx, enc, B, T, Tk, d, H, nm["xattn"], False, scale, cross_bias, drop=dxa, klen=enc_klen)
                x1 = self.lin(o2, self.W(nm["xattn"]["o"][0]), self._b(nm["xattn"]["o"][1]), M, d, d, resid=x1,
                              drop=sv["d_xo"])
            n2, sv["ln2"] = self.ln_fwd(x1, nm["ln2"][0], nm["ln2"][1], M, d, eps, rms=rms)
            y, sv["f"] = self._ffn_fwd(n2, M, d, F, nm["fc1"], nm["fc2"], act, x1, d_act, d_out)
        sv["dims"] = (B, T, d, H, F)
        return y, sv

    def _b(self, n):
        return self.P(n) if n else None

    def _oproj_bwd(self, dy, sv_attn, names, M, d, drop=None, dy_masked=None):
        """dy: grad wrt the (dropped) out_proj output; returns grad wrt attention output o.  dy_masked: see _ffn_bwd."""
        wn, bn = names["o"]
        gb = self.G(bn) if (bn and self.tr(wn)) else None
        if dy_masked is not None:
            dy, fused = dy_masked, True
        else:
            dy, fused = self._dropped(dy, drop, M * d, bias_grad=gb, N=d) if gb is not None else (self._dropped(dy, drop, M * d), False)
        if self.tr(wn):
            self.wgrad(dy, sv_attn["o"], self.G(wn), M, d, d, gb=None if fused else gb)
        do = self.new(M, d)
        self.dgrad(dy, self.W(wn), do, M, d, d)
        return do

    def layer_bwd(self, dy, sv, nm, pre_ln, act, rms=False, denc=None, dbias=None):
        B, T, d, H, F = sv["dims"]
        M = B * T
        if not pre_ln:
            # post-LN: every norm's input is x + dropout(Linear(...)), so the norm's backward also emits the masked gradient
            # that enters that Linear and its bias gradient (ln_bwd `fuse`; otherwise _dropped does it in a pass of its own)
            fz = self._fuse_site(sv["f"][4], nm["fc2"][0], nm["fc2"][1], d)
            ds3 = self.ln_bwd(dy, sv["ln2"], nm["ln2"][0], nm["ln2"][1], M, d, fuse=fz)
            dh = self._ffn_bwd(ds3, sv["f"], M, d, F, nm["fc1"], nm["fc2"], act, ds3, dy_masked=fz.get("out") if fz else None)
            if "x" in sv:
                fz = self._fuse_site(sv["d_xo"], nm["xattn"]["o"][0], nm["xattn"]["o"][1], d)
                ds2 = self.ln_bwd(dh, sv["lnx"], nm["lnx"][0], nm["lnx"][1], M, d, fuse=fz)
                do2 = self._oproj_bwd(ds2, sv["x"], nm["xattn"], M, d, sv["d_xo"], dy_masked=fz.get("out") if fz else None)
                dh = self.attn_bwd(do2, sv["x"], nm["xattn"], dx_resid=ds2, dkv_accum=denc)
            fz = self._fuse_site(sv["d_o"], nm["attn"]["o"][0], nm["attn"]["o"][1], d)
            ds1 = self.ln_bwd(dh, sv["ln1"], nm["ln1"][0], nm["ln1"][1], M, d, fuse=fz)
            do = self._oproj_bwd(ds1, sv["a"], nm["attn"], M, d, sv["d_o"], dy_masked=fz.get("out") if fz else None)
            return self.attn_bwd(do, sv["a"], nm["attn"], dx_resid=ds1, dbias=dbias)
        # pre-LN: y = x1 + ffn(LN2(x1)); x1 = x(+cross) + attn(LN1(x))
        dn2 = self._ffn_bwd(dy, sv["f"], M, d, F, nm["fc1"], nm["fc2"], act, None)
        dx1 = self.ln_bwd(dn2, sv["ln2"], nm["ln2"][0], nm["ln2"][1] if not rms else None, M, d, rms=rms, dres=dy)
        if "x" in sv:
            do2 = self._oproj_bwd(dx1, sv["x"], nm["xattn"], M, d, sv["d_xo"])
            dnx = self.attn_bwd(do2, sv["x"], nm["xattn"], dkv_accum=denc)
            dx1 = self.ln_bwd(dnx, sv["lnx"], nm["lnx"][0], nm["lnx"][1] if not rms else None, M, d, rms=rms, dres=dx1)
        do = self._oproj_bwd(dx1, sv["a"], nm["attn"], M, d, sv["d_o"])
        dn1 = self.attn_bwd(do, sv["a"], nm["attn"], dbias=dbias)
        return self.ln_bwd(dn1, sv["ln1"], nm["ln1"][0], nm["ln1"][1] if not rms else None, M, d, rms=rms, dres=dx1)

    # ------------------------------------------------------------------ SpeechMixAdapter (ref:speechmix/model.py:196-222)
    def _adapter_names(self, idx):
        p = f"adapters.{idx}."
        return (p + "0.weight", p + "0.bias"), (p + "1.weight", p + "1.bias"), (p + "3.weight", p + "3.bias")

    def adapter_fwd(self, x, idx, M, d):
        """Layer output -> Linear(ReLU(Linear(LayerNorm(x)))), bottleneck d/2, no residual (the reference's forward hook
        replaces the layer's hidden-state output)."""
        ln, n1, n2 = self._adapter_names(idx)
        n, lnsv = self.ln_fwd(x, ln[0], ln[1], M, d, 1e-5)
        y, fsv = self._ffn_fwd(n, M, d, d // 2, n1, n2, ACT_RELU, None)
        return y, (lnsv, fsv)

    def adapter_bwd(self, dy, sv, idx, M, d):
        ln, n1, n2 = self._adapter_names(idx)
        lnsv, fsv = sv
        dn = self._ffn_bwd(dy, fsv, M, d, d // 2, n1, n2, ACT_RELU, None)
        return self.ln_bwd(dn, lnsv, ln[0], ln[1], M, d)

    # ------------------------------------------------------------------ name tables
    def _w2v2_layer_names(self, i):
        p = f"{self.ep}encoder.layers.{i}."
        a = {k: (p + f"attention.{n}.weight", p + f"attention.{n}.bias")
             for k, n in (("q", "q_proj"), ("k", "k_proj"), ("v", "v_proj"), ("o", "out_proj"))}
        return dict(attn=a, ln1=(p + "layer_norm.weight", p + "layer_norm.bias"),
                    fc1=(p + "feed_forward.intermediate_dense.weight", p + "feed_forward.intermediate_dense.bias"),
                    fc2=(p + "feed_forward.output_dense.weight", p + "feed_forward.output_dense.bias"),
                    ln2=(p + "final_layer_norm.weight", p + "final_layer_norm.bias"))

    def _bart_layer_names(self, side, i):
        p = f"{self.lp}model.{side}.layers.{i}."

        def att(a):
            return {k: (p + f"{a}.{n}.weight", p + f"{a}.{n}.bias")
                    for k, n in (("q", "q_proj"), ("k", "k_proj"), ("v", "v_proj"), ("o", "out_proj"))}
        nm = dict(attn=att("self_attn"), ln1=(p + "self_attn_layer_norm.weight", p + "self_attn_layer_norm.bias"),
                  fc1=(p + "fc1.weight", p + "fc1.bias"), fc2=(p + "fc2.weight", p + "fc2.bias"),
                  ln2=(p + "final_layer_norm.weight", p + "final_layer_norm.bias"))
        if side == "decoder":
            nm["xattn"] = att("encoder_attn")
            nm["lnx"] = (p + "encoder_attn_layer_norm.weight", p + "encoder_attn_layer_norm.bias")
        return nm

    def _t5_layer_names(self, side, i):
        p = f"{self.lp}{side}.block.{i}.layer."

        def att(a):
            return {k: (p + f"{a}.{n}.weight", None) for k, n in (("q", "q"), ("k", "k"), ("v", "v"), ("o", "o"))}
        ff = "2" if side == "decoder" else "1"
        nm = dict(attn=att("0.SelfAttention"), ln1=(p + "0.layer_norm.weight", None),
                  fc1=(p + f"{ff}.DenseReluDense.wi.weight", None), fc2=(p + f"{ff}.DenseReluDense.wo.weight", None),
                  ln2=(p + f"{ff}.layer_norm.weight", None))
        if side == "decoder":
            nm["xattn"] = att("1.EncDecAttention")
            nm["lnx"] = (p + "1.layer_norm.weight", None)
        return nm

    # ------------------------------------------------------------------ CNN feature extractor
    def _conv_geom(self, N):
        ec = self.ec
        Ts, n = [], N
        for k, s in zip(ec.conv_kernel, ec.conv_stride):
            n = (n - k) // s + 1
            Ts.append(n)
        return Ts

    def cnn_fwd(self, wave, B, N):
        ec, ep = self.ec, self.ep
        Ts = self._conv_geom(N)
        group = ec.feat_extract_norm == "group"
        nl = len(ec.conv_dim)
        # `wave` is referenced by raw pointer from the conv0 descriptor until the very end of backward
        sv = dict(Ts=Ts, B=B, N=N, y=[], pre=[], ln=[], wp=[None], wave=wave)
        p0 = f"{ep}feature_extractor.conv_layers.0."
        C0, T0 = ec.conv_dim[0], Ts[0]
        cb0 = self.P(p0 + "conv.bias") if ec.conv_bias else None
        y0 = self.new(B * T0, C0)
        c0ws = self.workspace("conv0_partials", ops.conv0_workspace_floats(B, C0, ec.conv_kernel[0]), torch.float32)
        if group:
            stats = self.new(B * C0 * 2, dt=torch.float64)
            c0 = ops.conv0_params(wave, self.P(p0 + "conv.weight"), cb0, self.P(p0 + "layer_norm.weight"),
                                  self.P(p0 + "layer_norm.bias"), stats, y0, B, N, C0, ec.conv_kernel[0],
                                  ec.conv_stride[0], T0, True, partials=c0ws)
            ops.conv0_fwd(c0, self.dt)
            sv["c0"], sv["stats"] = c0, stats
            sv["ln"].append(None)
        else:
            u0 = y0
            c0 = ops.conv0_params(wave, self.P(p0 + "conv.weight"), cb0, None, None, None, u0, B, N, C0,
                                  ec.conv_kernel[0], ec.conv_stride[0], T0, False, partials=c0ws)
            ops.conv0_fwd(c0, self.dt)
            sv["c0"] = c0
            y0, lnsv = self.ln_fwd(u0, p0 + "layer_norm.weight", p0 + "layer_norm.bias", B * T0, C0, 1e-5, act=ACT_GELU)
            sv["ln"].append(lnsv)
        sv["y"].append(y0)
        sv["pre"].append(None)
        x, Cin, Tin = y0, C0, T0
        for i in range(1, nl):
            p = f"{ep}feature_extractor.conv_layers.{i}."
            Co, k, s, To = ec.conv_dim[i], ec.conv_kernel[i], ec.conv_stride[i], Ts[i]
            wp = self.new(Co, k * Cin)
            ops.pack_conv_w(self.P(p + "conv.weight"), wp, Co, Cin, k, self.dt)
            sv["wp"].append(wp)
            cb = self.P(p + "conv.bias") if ec.conv_bias else None
            av = view(s * Cin, To, Tin * Cin)
            if group:
                pre = self.new(B * To, Co)
                y = self.lin(x, wp, cb, B * To, Co, k * Cin, act=ACT_GELU, aux_out=pre, av=av)
                sv["pre"].append(pre)
                sv["ln"].append(None)
            else:
                u = self.lin(x, wp, cb, B * To, Co, k * Cin, av=av)
                y, lnsv = self.ln_fwd(u, p + "layer_norm.weight", p + "layer_norm.bias", B * To, Co, 1e-5, act=ACT_GELU)
                sv["pre"].append(None)
                sv["ln"].append(lnsv)
            sv["y"].append(y)
            x, Cin, Tin = y, Co, To
        return x, sv

    def cnn_bwd(self, dfeat, sv):
        """dfeat: grad wrt the last conv layer's (activated) output [B*T_last, C]."""
        ec, ep = self.ec, self.ep
        Ts, B = sv["Ts"], sv["B"]
        group = ec.feat_extract_norm == "group"
        nl = len(ec.conv_dim)
        PAD = 2
        # dpre of layer i lives in a zero-padded per-clip buffer [B, T_i + 2*PAD, C_i]
        i = nl - 1
        Ci, Ti = ec.conv_dim[i], Ts[i]
        dpre = self.persist_zeros(f"cnn_dpre{i}", B * (Ti + 2 * PAD), Ci)
        padv = view(Ci, Ti, (Ti + 2 * PAD) * Ci, PAD * Ci)
        if group:
            ops.act_bwd(dfeat, sv["pre"][i], dpre, B * Ti, Ci, padv, ACT_GELU, self.dt)
        else:
            p = f"{ep}feature_extractor.conv_layers.{i}."
            tmp = self.ln_bwd(dfeat, sv["ln"][i], p + "layer_norm.weight", p + "layer_norm.bias", B * Ti, Ci, act=ACT_GELU)
            ops.act_bwd(tmp, None, dpre, B * Ti, Ci, padv, ACT_NONE, self.dt)        # rows into the padded per-clip buffer
        for i in range(nl - 1, 0, -1):
            p = f"{ep}feature_extractor.conv_layers.{i}."
            Co, Cin, k, s = ec.conv_dim[i], ec.conv_dim[i - 1], ec.conv_kernel[i], ec.conv_stride[i]
            To, Tin = Ts[i], Ts[i - 1]
            Tp = To + 2 * PAD
            padv = view(Co, To, Tp * Co, PAD * Co)
            x = sv["y"][i - 1]
            xv = view(s * Cin, To, Tin * Cin)
            if self.tr(p + "conv.weight"):
                dwp = self.zeros(Co, k * Cin, dt=torch.float32)
                self.wgrad(dpre, x, dwp, B * To, Co, k * Cin, dyv=padv, xv=xv)
                ops.unpack_conv_dw(dwp, self.G(p + "conv.weight"), Co, Cin, k)
                if ec.conv_bias:
                    ops.colsum(dpre, self.G(p + "conv.bias"), B * Tp, Co, Co, self.dt, folds=self.folds)
            # data gradient: one GEMM per residue r of the input position modulo the stride
            if i - 1 >= 1 and group:
                Tpp = Tin + 2 * PAD
                dprev = self.persist_zeros(f"cnn_dpre{i - 1}", B * Tpp, Cin)
                prev_off, prev_bs = PAD * Cin, Tpp * Cin
            else:
                dprev = self.new(B * Tin, Cin)
                prev_off, prev_bs = 0, Tin * Cin
            # round 4 (SMX_CONV_DGRAD_KC=0: the rows-contiguous read of the forward's tap-major weight, rounds 1-3): the taps of
            # each residue transposed once per step into a K-contiguous operand (ops.pack_conv_w_dgrad, 0.8 - 1.5 MB), so the
            # data gradient is a forward-layout GEMM and runs on the 256-wide kernels
            wd = None
            if self.dt == BF16 and k >= s and os.environ.get("SMX_CONV_DGRAD_KC", "1") != "0":
                wd = self.new(Co * Cin * k)
                ops.pack_conv_w_dgrad(self.P(p + "conv.weight"), wd, Co, Cin, k, s, self.dt)
            wd_off = 0
            for r in range(s):
                taps = list(range(r, k, s))
                nj = len(taps)
                U = (Tin - 1 - r) // s + 1 if Tin - 1 - r >= 0 else 0
                if nj == 0:          # stride > kernel: no feature extractor on the path has it (k >= s in every config)
                    raise NotImplementedError("conv layer with stride > kernel size")
                wd_r, wd_off = (wd[wd_off:wd_off + Cin * nj * Co] if wd is not None else None), wd_off + Cin * nj * Co
                if U <= 0:
                    continue
                av = view(Co, U, Tp * Co, (PAD - (nj - 1)) * Co)
                cv = view(s * Cin, U, prev_bs, prev_off + r * Cin)
                aux = sv["pre"][i - 1] if (i - 1 >= 1 and group) else None
                ev = view(s * Cin, U, Tin * Cin, r * Cin) if aux is not None else None
                if wd_r is not None:
                    ops.gemm(dpre, wd_r, dprev, B * U, Cin, nj * Co, self.dt, av=av, cv=cv, ev=ev, aux_in=aux,
                             act=ACT_GELU if aux is not None else ACT_NONE)
                    continue
                bv = view(k * Cin, Co, -s * Cin, (r + (nj - 1) * s) * Cin)
                ops.gemm(dpre, sv["wp"][i], dprev, B * U, Cin, nj * Co, self.dt, b_rc=True, av=av, bv=bv, cv=cv, ev=ev,
                         aux_in=aux, act=ACT_GELU if aux is not None else ACT_NONE)
            if i - 1 >= 1 and not group:
                pp = f"{ep}feature_extractor.conv_layers.{i - 1}."
                tmp = self.ln_bwd(dprev, sv["ln"][i - 1], pp + "layer_norm.weight", pp + "layer_norm.bias", B * Tin, Cin,
                                  act=ACT_GELU)
                Tpp = Tin + 2 * PAD
                dprev = self.persist_zeros(f"cnn_dpre{i - 1}", B * Tpp, Cin)
                ops.act_bwd(tmp, None, dprev, B * Tin, Cin, view(Cin, Tin, Tpp * Cin, PAD * Cin), ACT_NONE, self.dt)
            dpre = dprev
        # layer 0
        p0 = f"{ep}feature_extractor.conv_layers.0."
        C0, T0 = ec.conv_dim[0], Ts[0]
        if not self.tr(p0 + "conv.weight"):
            return
        c0 = sv["c0"]
        if group:
            bstats = self.new(B * C0 * 2, dt=torch.float64)
            ops.conv0_bwd(c0, dpre, bstats, self.G(p0 + "conv.weight"), None, self.G(p0 + "layer_norm.weight"),
                          self.G(p0 + "layer_norm.bias"), self.dt)
        else:
            du = self.ln_bwd(dpre, sv["ln"][0], p0 + "layer_norm.weight", p0 + "layer_norm.bias", B * T0, C0, act=ACT_GELU)
            ops.conv0_bwd(c0, du, None, self.G(p0 + "conv.weight"), self.G(p0 + "conv.bias") if ec.conv_bias else None,
                          None, None, self.dt)

    # ------------------------------------------------------------------ positional conv embedding
    def _posconv_J(self):
        """Frames per GEMM row of the time-blocked positional conv (csrc/posconv.hip; SMX_POSCONV_J=1: the plain N = Cg form)."""
        if self.dt != BF16 and os.environ.get("SMX_POSCONV_J_F32") != "1":      # (fp32 parity path: the plain form; =1: tests of the blocked one)
            return 1
        J = int(os.environ.get("SMX_POSCONV_J", "4"))
        return J if J > 1 and (J * (self.ec.hidden_size // self.ec.num_conv_pos_embedding_groups)) % 8 == 0 else 1

    def posconv_fwd(self, h, B, T):
        ec, ep = self.ec, self.ep
        d, K, G = ec.hidden_size, ec.num_conv_pos_embeddings, ec.num_conv_pos_embedding_groups
        Cg, P, Tp = d // G, ec.num_conv_pos_embeddings // 2, T + ec.num_conv_pos_embeddings - 1
        pre_n = f"{ep}encoder.pos_conv_embed.conv."
        wp = self.new(G * Cg * K * Cg)
        norm = self.new(ops.wn_scratch_floats(d, Cg, K), dt=torch.float32)        # [K] norms, then reduction scratch
        J = self._posconv_J()
        # (time-blocked form: the flipped data-gradient pack [G][ci][k' Cg + co] is emitted here too - its blocked operand is then the
        # same shifted row copy as the forward's)
        wf = self.new(G * Cg * K * Cg) if J > 1 else None
        ops.wn_fwd(self.P(pre_n + "parametrizations.weight.original1"), self.P(pre_n + "parametrizations.weight.original0"),
                   wp, wf, norm, d, Cg, K, self.dt)
        s = self.new(B * T, d)
        pre = self.new(B * T, d)
        if J > 1:
            # time-blocked: row (b, t') = frames J t' .. J t' + J - 1, N = J Cg outputs, K' = (K + J - 1) Cg inputs; the fp32 output is
            # the group-major layout [G][B][Tq][Cg], unpacked with the epilogue the fused GEMM had (same fp32 arithmetic)
            Tb = (T + J - 1) // J
            Tq, Kp = J * Tb, (K + J - 1) * Cg
            Tp2 = Tq + K - 1
            wJ = self.new(G * J * Cg * Kp)
            ops.posconv_pack_w(wp, wJ, G, Cg, K, J, False, self.dt)
            xg = self.new(G * B * Tp2 * Cg)
            ops.group_pack(h, xg, B, T, d, G, K + (Tq - T), P, self.dt)
            tmp = self.new(G * B * Tq * Cg, dt=torch.float32)
            ops.gemm(xg, wJ, tmp, B * Tb, J * Cg, Kp, self.dt, av=view(J * Cg, Tb, Tp2 * Cg), bv=view(Kp), cv=view(J * Cg), out_f32=True,
                     nbatch=G, batch_a=B * Tp2 * Cg, batch_b=J * Cg * Kp, batch_c=B * Tq * Cg)
            ops.posconv_unpack(tmp, self.P(pre_n + "bias"), h, pre, s, B, T, d, G, Tq, ACT_GELU, self.dt)
            return s, dict(xg=xg, wp=wp, wf=wf, norm=norm, pre=pre, B=B, T=T, J=J)
        xg = self.new(G * B * Tp * Cg)
        ops.group_pack(h, xg, B, T, d, G, K, P, self.dt)
        ops.gemm(xg, wp, s, B * T, Cg, K * Cg, self.dt, av=view(Cg, T, Tp * Cg), bv=view(K * Cg), cv=view(d),
                 bias=self.P(pre_n + "bias"), resid=h, aux_out=pre, act=ACT_GELU, nbatch=G, batch_a=B * Tp * Cg,
                 batch_b=Cg * K * Cg, batch_c=Cg, batch_bias=Cg)
        return s, dict(xg=xg, wp=wp, norm=norm, pre=pre, B=B, T=T, J=1)

    def posconv_bwd(self, ds, sv):
        ec, ep = self.ec, self.ep
        B, T = sv["B"], sv["T"]
        d, K, G = ec.hidden_size, ec.num_conv_pos_embeddings, ec.num_conv_pos_embedding_groups
        Cg, P, Tp = d // G, K // 2, T + K - 1
        pre_n = f"{ep}encoder.pos_conv_embed.conv."
        dpre = self.new(B * T, d)
        ops.act_bwd(ds, sv["pre"], dpre, B * T, d, view(d), ACT_GELU, self.dt)
        v_n, g_n = pre_n + "parametrizations.weight.original1", pre_n + "parametrizations.weight.original0"
        J = sv.get("J", 1)
        train = self.tr(v_n, g_n, pre_n + "bias")
        if train:
            ops.colsum(dpre, self.G(pre_n + "bias"), B * T, d, d, self.dt, folds=self.folds)
        n = G * Cg * K * Cg
        if J > 1:
            Tb = (T + J - 1) // J
            Tq, Kp = J * Tb, (K + J - 1) * Cg
            Tp2 = Tq + K - 1
            dyg = self.new(G * B * Tp2 * Cg)               # d pre, group-major, K - 1 - P zero frames in front: both gradients read it
            ops.group_pack(dpre, dyg, B, T, d, G, K + (Tq - T), K - 1 - P, self.dt)
            if train:
                # blocked weight gradient [G][J Cg][K'] = dY^T X over the B Tb rows (frame j of row t' sits K - 1 - P + J t' + j frames
                # into dyg), then its J shifted diagonals folded into the forward-pack layout
                nJ = J * Cg * Kp
                # (the one-workgroup-per-CU kernel only where the backward policy allows it - ops.pp_allowed - and the CU budget in
                # the key, like _wgrad_gemm / _choose_mode: a pick made on one GPU must not be reused beside RCCL)
                cands = [(1, 1), (1, 2), (1, 4)] + ([(8, 1), (8, 2)] if ops.pp_allowed() else [])
                key = ("posconv_wgrad", self.dt, B, T, G, Cg, K, J, ops.pp_cus())
                pick = ops._tuned_get(key)
                if isinstance(pick, list):
                    pick = tuple(pick)

                def run(mode, split, out):
                    ops.gemm(dyg, sv["xg"], out, J * Cg, Kp, B * Tb, self.dt, a_rc=True, b_rc=True,
                             av=view(J * Cg, Tb, Tp2 * Cg, (K - 1 - P) * Cg), bv=view(J * Cg, Tb, Tp2 * Cg), cv=view(Kp), out_f32=True,
                             atomic=0, split_k=split, split_stride=G * nJ if split > 1 else 0, nbatch=G, batch_a=B * Tp2 * Cg,
                             batch_b=B * Tp2 * Cg, batch_c=nJ, tr_mode=mode)
                slabs = self.workspace("posconv_slabs", 4 * G * nJ, torch.float32)
                if pick not in cands:
                    ts = ops.measure_candidates({c: (lambda c=c: run(c[0], c[1], slabs)) for c in cands})
                    pick = min(ts, key=ts.get)
                    ops._tuned_set(key, pick)
                    ops.TUNE_LIVE_KEYS.append(key)
                dwJ = self.new(G * nJ, dt=torch.float32)
                if pick[1] > 1:
                    run(pick[0], pick[1], slabs)
                    ops.reduce_slabs(slabs, pick[1], G * nJ, G * nJ, dwJ, accumulate=False)
                else:
                    run(pick[0], 1, dwJ)
                dwp = self.new(n, dt=torch.float32)
                ops.posconv_fold_dw(dwJ, dwp, G, Cg, K, J)
                scratch = self.new(ops.wn_scratch_floats(d, Cg, K), dt=torch.float32)
                ops.wn_bwd(dwp, self.P(v_n), self.P(g_n), sv["norm"], scratch, self.G(g_n), self.G(v_n), d, Cg, K)
            wJf = self.new(G * J * Cg * Kp)
            ops.posconv_pack_w(sv["wf"], wJf, G, Cg, K, J, False, self.dt)
            tmp = self.new(G * B * Tq * Cg, dt=torch.float32)
            ops.gemm(dyg, wJf, tmp, B * Tb, J * Cg, Kp, self.dt, av=view(J * Cg, Tb, Tp2 * Cg), bv=view(Kp), cv=view(J * Cg), out_f32=True,
                     nbatch=G, batch_a=B * Tp2 * Cg, batch_b=J * Cg * Kp, batch_c=B * Tq * Cg)
            dh = self.new(B * T, d)
            ops.posconv_unpack(tmp, None, ds, None, dh, B, T, d, G, Tq, ACT_NONE, self.dt)
            return dh
        if train:
            dwp = self.new(n, dt=torch.float32)
            split = self._split(Cg, K * Cg * G, B * T)
            slabs = self.workspace("wgrad_slabs", split * n, torch.float32)
            ops.gemm(dpre, sv["xg"], slabs, Cg, K * Cg, B * T, self.dt, a_rc=True, b_rc=True, av=view(d),
                     bv=view(Cg, T, Tp * Cg), cv=view(K * Cg), out_f32=True, atomic=0, split_k=split, split_stride=n,
                     nbatch=G, batch_a=Cg, batch_b=B * Tp * Cg, batch_c=Cg * K * Cg)
            ops.reduce_slabs(slabs, split, n, n, dwp, accumulate=False)
            scratch = self.new(ops.wn_scratch_floats(d, Cg, K), dt=torch.float32)
            ops.wn_bwd(dwp, self.P(v_n), self.P(g_n), sv["norm"], scratch, self.G(g_n), self.G(v_n), d, Cg, K)
        dyg = self.new(G * B * Tp * Cg)
        ops.group_pack(dpre, dyg, B, T, d, G, K, K - 1 - P, self.dt)
        dh = self.new(B * T, d)
        ops.gemm(dyg, sv["wp"], dh, B * T, Cg, K * Cg, self.dt, b_rc=True, av=view(Cg, T, Tp * Cg),
                 bv=view(K * Cg, Cg, -Cg, (K - 1) * Cg), cv=view(d), resid=ds, nbatch=G, batch_a=B * Tp * Cg,
                 batch_b=Cg * K * Cg, batch_c=Cg)
        return dh

    # ------------------------------------------------------------------ SpecAugment indices (host RNG)
    def _spec_augment_rows(self, B, T, lengths=None, host_only=False):
        """Flat row indices (b * T + t) of the frames HF's `_mask_hidden_states` replaces with `masked_spec_embed`
        (TF:models/wav2vec2/modeling_wav2vec2.py:1074-1119), or None."""
        ec = self.ec
        if getattr(ec, "mask_feature_prob", 0.0) > 0:
            raise NotImplementedError("mask_feature_prob > 0 (SpecAugment along the feature axis) is not built; the reference's "
                                      "checkpoints ship 0.0")
        if self._cap is not None:
            # fixed-capacity device list (negative = unused), refilled per replay
            return self._cap.spec_rows(self, B, T) if (ec.mask_time_prob > 0 and lengths is None) else None
        rec = getattr(self.host_rng, "mask", None)
        if rec is not None:
            mask = rec
        else:
            if ec.mask_time_prob <= 0:
                return np.zeros(0, dtype=np.int32) if host_only else None
            mask = compute_mask_indices((B, T), ec.mask_time_prob, ec.mask_time_length, self.host_rng, lengths,
                                        ec.mask_time_min_masks)
        self.last_spec_mask = mask
        rows = np.flatnonzero(mask.reshape(-1))
        if host_only:
            return rows.astype(np.int32)
        if rows.size == 0:
            return None
        return self.h2d(rows.astype(np.int32))

    # ------------------------------------------------------------------ attention dropout masks ahead of time (round 4)
    # The keep masks of the speech encoder's attention-probability dropout are bit matrices that depend on (seed, shape) only:
    # 41 us of VALU work per layer and step at config 2, on the critical path right before every attention forward.  The
    # optimizer step that ends a training step is HBM-bound (Adafactor: 1.4 ms at ~3.6 TB/s with the vector units mostly idle),
    # so `pregen_attention_masks` generates the NEXT step's masks for all layers on a second stream beside it; the next forward
    # takes a layer's pair if batch and frame count still match (`_take_premask`), otherwise - first step, another batch shape,
    # eval mode, gradient accumulation's inner micro-batches - the mask is generated in place as before.  A dropped layer's pair
    # is simply not used.  SMX_PREGEN_MASKS=0: off.
    def pregen_attention_masks(self, B, T):
        ec = self.ec
        p = float(ec.attention_dropout)
        if (self.dt != BF16 or p <= 0 or self.dev.type != "cuda" or os.environ.get("SMX_PREGEN_MASKS") == "0"
                or ec.hidden_size // ec.num_attention_heads != 64):
            self._premask = None
            return
        st = getattr(self, "_mask_stream", None)
        if st is None:
            st = self._mask_stream = torch.cuda.Stream()
        bufs = self._persist.setdefault("_premask_bufs", {})      # (B, T, layer) -> pair: the SAME buffers whenever the shape returns
        # keyed mode: the NEXT pass's step key is drawn and set now (on the main stream, behind this step's backward and ahead of
        # the event the mask stream waits for); the masks are hashed with the sites' fixed seeds + that key
        key = None
        if self.keyed():
            key = self._preset_key = int(self.drop_rng.integers(1, 2 ** 32 - 1))
            ops.set_step_key(key, self.dev.index)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())          # behind the backward that still reads the previous pairs
        st.wait_event(ev)
        new = dict(B=B, T=T, layers={}, key=key)
        with torch.cuda.stream(st):
            for i in range(self.L):
                seed = site_seed(f"enc{i}/attn") if key is not None else int(self.drop_rng.integers(1, 2 ** 32 - 1))
                masks = ops.attn_dropout_masks(B, ec.num_attention_heads, T, T, 64, p, seed, self.dt, self.dev,
                                               reuse=bufs.get((B, T, i)))
                if masks is None:
                    self._premask = None
                    return
                bufs[(B, T, i)] = masks
                if len(bufs) > 4 * self.L:                # (a few shapes at most: ragged training sets fall back to in-place masks)
                    for k in [k for k in bufs if k[:2] != (B, T)]:
                        del bufs[k]
                new["layers"][i] = dict(p=p, seed=seed, masks=masks, key=key)
            new["ev"] = torch.cuda.Event()
            new["ev"].record(st)
        self._premask = new

    def _take_premask(self, i, B, T):
        pm = getattr(self, "_premask", None)
        if pm is None or pm["B"] != B or pm["T"] != T:
            return None
        if self._cap is not None:
            # capture pass: the layer's graph reads the pair's (static, reused) buffers; the replay waits for the generation
            # itself and checks nothing - every replayed step is preceded by a pregen with the step's key
            e = pm["layers"].get(i)
            return dict(e, key=self._key) if e is not None else None
        if pm.get("ev") is not None:                     # first use in this forward: the main stream waits for the generation
            torch.cuda.current_stream().wait_event(pm["ev"])
            pm["ev"] = None
        e = pm["layers"].get(i)
        if e is None or e.get("used"):
            return None                                  # (a pair serves ONE forward; its buffers are overwritten by the next pregen)
        e["used"] = True
        return e

    # ------------------------------------------------------------------ speech encoder
    def speech_fwd(self, wave, B, N, training, sample_lengths=None):
        """sample_lengths: per-clip count of real (unpadded) samples = a right-padded `attention_mask` into the speech encoder.
        HF then (TF:models/wav2vec2/modeling_wav2vec2.py:1041-1060, 1349-1358, 688-697) reduces it to frame lengths with the conv
        arithmetic, draws the SpecAugment spans inside each clip's real frames, zeroes the padded frames ahead of the positional
        convolution and masks them as attention keys in every layer; the CNN itself still sees the padding."""
        ec, ep = self.ec, self.ep
        d, eps = ec.hidden_size, ec.layer_norm_eps
        if not training:
            self._premask = None                        # (pairs generated for a training step that did not come)
        feat, cnn_sv = self.cnn_fwd(wave, B, N)
        T = cnn_sv["Ts"][-1]
        C = ec.conv_dim[-1]
        M = B * T
        sv = dict(cnn=cnn_sv, T=T, B=B)
        frame_len = klen = None
        if sample_lengths is not None:
            frame_len = [max(1, min(T, ec.frames(int(n)))) for n in sample_lengths]
            if len(frame_len) != B:
                raise ValueError("one length per clip")
            klen = self.h2d(np.asarray(frame_len, dtype=np.int32))
        sv["frame_len"] = frame_len
        if ec.feat_proj_layer_norm:
            fn, sv["fp_ln"] = self.ln_fwd(feat, ep + "feature_projection.layer_norm.weight",
                                          ep + "feature_projection.layer_norm.bias", M, C, eps)
        else:
            fn = feat
        sv["fp_in"] = fn
        self._scope = "front"
        sv["d_fp"] = self._dp(ec.feat_proj_dropout, "fp") if training else None      # TF:...wav2vec2.py:429-434
        h = self.lin(fn, self.W(ep + "feature_projection.projection.weight"), self.P(ep + "feature_projection.projection.bias"),
                     M, d, C, drop=sv["d_fp"])
        sv["mask_rows"] = None
        if training and ec.apply_spec_augment and self.has(ep + "masked_spec_embed"):
            rows = self._spec_augment_rows(B, T, frame_len)
            if rows is not None:
                ops.mask_rows(h, rows, rows.numel(), self.P(ep + "masked_spec_embed"), d, self.dt)
                sv["mask_rows"] = rows
        sv["pad_rows"] = None
        if frame_len is not None and min(frame_len) < T:           # "make sure padded tokens output 0" (TF:...wav2vec2.py:688-692)
            pad = np.concatenate([np.arange(n, T) + b * T for b, n in enumerate(frame_len)]).astype(np.int32)
            pad_rows = self.h2d(pad)
            ops.mask_rows(h, pad_rows, pad_rows.numel(), self.zeros(d, dt=torch.float32), d, self.dt)
            sv["pad_rows"] = pad_rows
        s, sv["pc"] = self.posconv_fwd(h, B, T)
        stable = ec.do_stable_layer_norm
        d_in = self._dp(ec.hidden_dropout, "in") if training else None                # TF:...wav2vec2.py:700-703 / 786-788
        sv["d_in"] = d_in
        if not stable:
            x, sv["enc_ln"] = self.ln_fwd(s, ep + "encoder.layer_norm.weight", ep + "encoder.layer_norm.bias", M, d, eps,
                                          drop=d_in)
        else:
            x = self._dropped(s, d_in, M * d)
        drop = (ec.hidden_dropout, ec.attention_dropout, ec.activation_dropout) if training else None
        act = _act_id(ec.hidden_act)
        sv["layers"] = []
        hidden = [x]
        if self._cap is None:
            self.wait_params()          # (the front end above reads front-end tensors only; a replayed step waits between its graphs)
        for i in range(self.L):
            self._seg("front" if i == 0 else f"enc_fwd{i - 1}")
            self._note(**{f"fwd_in{i}": x})
            # TF:...wav2vec2.py:709-723: one draw per layer - HF draws in eval mode too (`torch.rand([])` precedes `self.training and`),
            # so the draw is unconditional here as well; only train-mode draws decide anything
            draw = self.host_rng.layerdrop() if self._cap is None else 2.0
            if training and draw < ec.layerdrop:
                sv["layers"].append(None)
                hidden.append(x)
                continue
            ops.GEMM_TAG = "enc_layer"
            self._scope = f"enc{i}"
            x, lsv = self.layer_fwd(x, B, T, d, ec.num_attention_heads, ec.intermediate_size, self._w2v2_layer_names(i),
                                    stable, act, eps, drop=drop, klen=klen, premask=self._take_premask(i, B, T))
            ops.GEMM_TAG = None
            sv["layers"].append(lsv)
            hidden.append(x)
            self._note(**{f"fwd_out{i}": x})
        self._seg(f"enc_fwd{self.L - 1}" if self.L else "front")
        if stable:
            x, sv["final_ln"] = self.ln_fwd(x, ep + "encoder.layer_norm.weight", ep + "encoder.layer_norm.bias", M, d, eps)
            hidden[-1] = x
        sv["hidden"] = hidden
        self.last_dropped = [i for i, l in enumerate(sv["layers"]) if l is None]       # LayerDrop: no gradient this step
        return x, sv

    def speech_bwd(self, dx, sv, ws=None):
        """dx: grad wrt the encoder's last hidden state.  ws = (dxw, sw): layer-weighted-sum mode - every hidden
        state l additionally receives sw[l] * dxw (ref:speechmix/hf_model.py:411-423)."""
        ec, ep = self.ec, self.ep
        d = ec.hidden_size
        B, T = sv["B"], sv["T"]
        M = B * T
        C = ec.conv_dim[-1]
        stable = ec.do_stable_layer_norm
        act = _act_id(ec.hidden_act)
        if ws is not None:
            dxw, sw = ws
            dx = self.new(M, d)
            ops.axpy_dev(dx, dxw, sw, self.L, M * d, True, self.dt)
        if stable:
            dx = self.ln_bwd(dx, sv["final_ln"], ep + "encoder.layer_norm.weight", ep + "encoder.layer_norm.bias", M, d)
        pending = None          # stage whose grouped weight gradients are still running on the second stream
        self._seg("pre_layers")
        for i in range(self.L - 1, -1, -1):
            self._note(**{f"bwd_in{i}": dx})
            if sv["layers"][i] is not None:
                ops.GEMM_TAG = "enc_layer"
                self._wg_begin()
                dx = self.layer_bwd(dx, sv["layers"][i], self._w2v2_layer_names(i), stable, act)
                # The layer's grouped weight-gradient launch can go to a second stream, joined one layer LATER
                # (SMX_WGRAD_SIDE=1; default: on this stream, joined at once): nothing downstream needs it before its stage is reported
                if pending is not None:
                    self._join_wg(pending)
                    pending = None
                on_side = self._wg_flush(side=self._wg_side_stream())
                ops.GEMM_TAG = None
                if on_side:
                    if ws is not None:
                        ops.axpy_dev(dx, ws[0], ws[1], i, M * d, False, self.dt)
                    pending = f"enc_layer{i}"
                    continue
            if pending is not None:              # (a dropped layer: stages are still reported in layer order on every rank)
                self._join_wg(pending)
                pending = None
            if ws is not None:
                ops.axpy_dev(dx, ws[0], ws[1], i, M * d, False, self.dt)
            self._note(**{f"bwd_out{i}": dx})
            self._stage(f"enc_layer{i}")
        if pending is not None:
            self._join_wg(pending)
        if not stable:
            dx = self.ln_bwd(dx, sv["enc_ln"], ep + "encoder.layer_norm.weight", ep + "encoder.layer_norm.bias", M, d)
        else:
            dx = self._dropped(dx, sv["d_in"], M * d)
        dh = self.posconv_bwd(dx, sv["pc"])
        if sv.get("pad_rows") is not None:                         # zeroed frames pass no gradient back
            ops.mask_rows_bwd(dh, sv["pad_rows"], sv["pad_rows"].numel(), None, d, self.dt)
        if sv["mask_rows"] is not None:
            rows = sv["mask_rows"]
            me = ep + "masked_spec_embed"
            ops.mask_rows_bwd(dh, rows, rows.numel(), self.G(me) if self.tr(me) else None, d, self.dt)
        wn, bn = ep + "feature_projection.projection.weight", ep + "feature_projection.projection.bias"
        dh = self._dropped(dh, sv["d_fp"], M * d)
        if self.tr(wn):
            self.wgrad(dh, sv["fp_in"], self.G(wn), M, d, C, gb=self.G(bn))
        dfn = self.new(M, C)
        self.dgrad(dh, self.W(wn), dfn, M, d, C)
        if ec.feat_proj_layer_norm:
            dfeat = self.ln_bwd(dfn, sv["fp_ln"], ep + "feature_projection.layer_norm.weight",
                                ep + "feature_projection.layer_norm.bias", M, C)
        else:
            dfeat = dfn
        if any(self.st.requires_grad(n) for n in self.st.offsets if n.startswith(ep + "feature_extractor.")):
            self.cnn_bwd(dfeat, sv["cnn"])

    # ------------------------------------------------------------------ length adapters + projection
    def bridge_fwd(self, x, B, T):
        d = self.ec.hidden_size
        sv = dict(B=B, Ts=[T], xs=[], wps=[])
        for i in range(self.downloop):
            To = (T - 2) // 2 + 1
            wp = self.new(d, 2 * d)
            ops.pack_conv_w(self.P(f"length_adapters.{i}.weight"), wp, d, d, 2, self.dt)
            y = self.lin(x, wp, self.P(f"length_adapters.{i}.bias"), B * To, d, 2 * d, av=view(2 * d, To, T * d))
            sv["xs"].append(x)
            sv["wps"].append(wp)
            x, T = y, To
            sv["Ts"].append(T)
        dd = self.lc.d_model
        sv["post_adapter"] = x
        e = self.lin(x, self.W("enc_to_dec_proj.weight"), self.P("enc_to_dec_proj.bias"), B * T, dd, d)
        return e, T, sv

    def bridge_bwd(self, de, sv):
        d, dd, B = self.ec.hidden_size, self.lc.d_model, sv["B"]
        T = sv["Ts"][-1]
        x = sv["post_adapter"]
        if self.tr("enc_to_dec_proj.weight"):
            self.wgrad(de, x, self.G("enc_to_dec_proj.weight"), B * T, dd, d, gb=self.G("enc_to_dec_proj.bias"))
        dx = self.new(B * T, d)
        self.dgrad(de, self.W("enc_to_dec_proj.weight"), dx, B * T, dd, d)
        for i in range(self.downloop - 1, -1, -1):
            Tin, To = sv["Ts"][i], sv["Ts"][i + 1]
            xin, wp = sv["xs"][i], sv["wps"][i]
            wn, bn = f"length_adapters.{i}.weight", f"length_adapters.{i}.bias"
            if self.tr(wn):
                dwp = self.zeros(d, 2 * d, dt=torch.float32)
                self.wgrad(dx, xin, dwp, B * To, d, 2 * d, xv=view(2 * d, To, Tin * d), gb=self.G(bn))
                ops.unpack_conv_dw(dwp, self.G(wn), d, d, 2)
            dprev = self.persist_zeros(f"adapter_dx{i}", B * Tin, d)
            # k == stride: the column matrix IS the input -> plain dgrad through the tap-major weight
            self.dgrad(dx, wp, dprev, B * To, d, 2 * d, cv=view(2 * d, To, Tin * d))
            dx = dprev
        return dx

    # ------------------------------------------------------------------ seq2seq LM
    def _check_positions(self, S=0, Ld=0):
        """BART / mBART learned positions (TF:models/bart/modeling_bart.py:74-98: rows 2 + arange(T) of a
        [max_position_embeddings + 2, d] table).  HF raises IndexError past the table; the kernels index it unchecked
        (and in backward add into it), so the bound is enforced here, on the host, before any launch."""
        lp = self.lp
        for side, T in (("encoder", S), ("decoder", Ld)):
            name = f"{lp}model.{side}.embed_positions.weight"
            if T and self.has(name):
                rows = self.st.offsets[name][2][0]
                if T + 2 > rows:
                    raise IndexError(f"{side} sequence of {T} positions needs {T + 2} rows of {name}, which has {rows} "
                                     f"(max_position_embeddings = {rows - 2}): index out of range in self")

    def _check_ids(self, ids, V, what, allow_ignore=False):
        """Token ids must lie in [0, V) (labels: or equal -100).  Free for host tensors; device tensors are checked only
        with SMX_CHECK_IDS=1 (the reduction forces a device sync)."""
        if ids is None or (ids.is_cuda and os.environ.get("SMX_CHECK_IDS") != "1"):
            return
        if ids.numel() == 0:
            return
        x = ids
        if allow_ignore:
            x = x[x != -100]
            if x.numel() == 0:
                return
        lo, hi = int(x.min()), int(x.max())
        if lo < 0 or hi >= V:
            raise IndexError(f"{what}: token id out of range [0, {V}) (min {lo}, max {hi})")

    def _t5_bias(self, side, Tq, Tk):
        """Relative-position bias [H, Tq, Tk] fp32 (TF:models/t5/modeling_t5.py:216-279).  Bucket indices are
        integer host work; the gather from the [buckets, H] table is a pure index op."""
        lc = self.lc
        name = f"{self.lp}{side}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"
        nb, md = lc.relative_attention_num_buckets, lc.relative_attention_max_distance
        ck = ("_t5_buckets", side, Tq, Tk)
        if ck in self._persist:                     # (cached on the device: no host-to-device copy inside a step)
            buckets = self._persist[ck]
            return self.P(name)[buckets].permute(2, 0, 1).contiguous(), buckets
        ctx = torch.arange(Tq)[:, None]
        mem = torch.arange(Tk)[None, :]
        rel = mem - ctx
        bidir = side == "encoder"
        ret = torch.zeros_like(rel)
        n = nb
        if bidir:
            n //= 2
            ret += (rel > 0).long() * n
            rel = rel.abs()
        else:
            rel = -torch.min(rel, torch.zeros_like(rel))
        max_exact = n // 2
        small = rel < max_exact
        large = max_exact + (torch.log(rel.float() / max_exact) / math.log(md / max_exact) * (n - max_exact)).long()
        large = torch.min(large, torch.full_like(large, n - 1))
        buckets = self._persist[ck] = (ret + torch.where(small, rel, large)).to(self.dev)
        table = self.P(name)                                   # [buckets, H] fp32
        return table[buckets].permute(2, 0, 1).contiguous(), buckets

    def _t5_bias_bwd(self, entry, buckets):
        tname, dbias, T = entry
        H = dbias.shape[0]
        ops.attn_bias_scatter(dbias, buckets.reshape(-1).to(torch.int32).contiguous(), self.G(tname), H, T, T,
                              self.lc.relative_attention_num_buckets)

    def head_chunk_rows(self, Md, Vp):
        """Rows per chunk of the streamed LM head, or 0 when the head runs as one GEMM.  The [B L, V] fp32 logits (+ their
        gradient in the compute dtype) are what the reference's author trims with `preprocess_logits_for_metrics`
        (ref:train.py:312-313): 0.3 GB at config 2, 1.5 GB at config 4 (V = 250 054).  Streaming computes them chunk by chunk
        inside the loss (forward) and again inside backward: they are never materialised, at the price of one more head
        GEMM per step and one read-modify-write pass over the tied embedding's fp32 gradient per chunk (config 4, B = 32:
        80.3 -> 87.5 ms/step with 128-row chunks for 2 GB less peak memory of 38.6 GB - measured round 3 - so `auto` only streams
        when logits + gradient exceed 2 GB, in chunks of up to 1 GB).  SMX_HEAD_STREAM = auto | 0 | 1 | <rows>."""
        mode = os.environ.get("SMX_HEAD_STREAM", "auto")
        es = 2 if self.dt == BF16 else 4
        total = Md * Vp * (4 + es)
        if mode == "0" or (mode == "auto" and total <= (2048 << 20)):
            return 0
        if mode.isdigit() and int(mode) > 1:
            return min(Md, max(64, int(mode) // 64 * 64))
        rows = (1024 << 20) // (Vp * (4 + es)) // 64 * 64
        return min(Md, max(64, rows))

    def _head_logits(self, sv, y_rows, out, rows):
        """logits of `rows` decoder rows: tied-embedding head GEMM (+ final_logits_bias), fp32 out."""
        ops.gemm(y_rows, self.W(sv["head"]), out, rows, sv["V"], self.lc.d_model, self.dt, cv=view(sv["Vp"]), bias=sv["flb"],
                 out_f32=True, alpha=sv["head_alpha"])

    def lm_fwd(self, inputs_embeds, input_ids, dec_ids, B, S, Ld, training, enc_klen=None, want_logits=True):
        """enc_klen: int32 [B] on the device = valid text-encoder positions per row (a right-padded `attention_mask`, the
        reference's hook ref:speechmix/model.py:132-136): masked as keys in the text encoder's self-attention and in the
        decoder's cross-attention (TF:models/bart/modeling_bart.py:741-760, 1010-1016)."""
        lc, lp = self.lc, self.lp
        d = lc.d_model
        t5 = lc.model_type == "t5"
        pre_ln = lc.model_type in ("mbart", "t5")
        act = _act_id(lc.activation_function)
        sv = dict(B=B, S=S, Ld=Ld, t5=t5, key=getattr(self, "_key", None) if training else None)
        self._scope = "lm"
        emb_name = lp + ("shared.weight" if t5 else "model.shared.weight")
        escale = math.sqrt(d) if (lc.scale_embedding and not t5) else 1.0
        sv["emb_name"], sv["escale"] = emb_name, escale
        H, F = lc.encoder_attention_heads, lc.encoder_ffn_dim
        # dropout sites: TF:models/bart/modeling_bart.py:826-829,1062-1064 (after layernorm_embedding) + the per-layer
        # ones; T5: on the stack input and after the final norm (TF:models/t5/modeling_t5.py:1020,1101-1102)
        drop = (lc.dropout, lc.attention_dropout, lc.activation_dropout) if training else None
        pdrop = lc.dropout if training else 0.0
        # ---- text encoder
        if inputs_embeds is None:
            x = self.new(B * S, d)
            ops.embed_fwd(input_ids, self.W(emb_name), x, B * S, d, escale, self.dt)
            sv["enc_ids"] = input_ids
        else:
            x = inputs_embeds
            sv["enc_ids"] = None
        ebias = dbias = None
        if t5:
            if lc.is_gated_act:
                raise NotImplementedError("gated T5 feed-forward is not on the SpeechMix path")
            ebias, eb = self._t5_bias("encoder", S, S)
            dbias, db = self._t5_bias("decoder", Ld, Ld)
            sv["t5_buckets"] = (eb, db)
            sv["d_enc_in"] = self._dp(pdrop, "enc_in")
            h = self._dropped(x, sv["d_enc_in"], B * S * d)
            eps = lc.layer_norm_epsilon
        else:
            eps = 1e-5
            pe = lp + "model.encoder."
            self._check_positions(S, Ld)
            h, sv["enc_emb_ln"] = self.ln_fwd(x, pe + "layernorm_embedding.weight", pe + "layernorm_embedding.bias", B * S, d,
                                              eps, pos=self.W(pe + "embed_positions.weight"), pos_period=S, pos_offset=2,
                                              want_sum=True, drop=self._dp(pdrop, "enc_emb"))
        sv["enc_layers"] = []
        for i in range(lc.encoder_layers):
            nm = self._t5_layer_names("encoder", i) if t5 else self._bart_layer_names("encoder", i)
            self._scope = f"lme{i}"
            h, lsv = self.layer_fwd(h, B, S, d, H, F, nm, pre_ln, act, eps, scale=1.0 if t5 else None, rms=t5, bias=ebias,
                                    drop=drop, klen=enc_klen)
            if self.lm_adapters:
                h, lsv["adapter"] = self.adapter_fwd(h, i, B * S, d)
            sv["enc_layers"].append(lsv)
        self._scope = "lm"
        if t5:
            h, sv["enc_final_ln"] = self.ln_fwd(h, lp + "encoder.final_layer_norm.weight", None, B * S, d, eps, rms=True,
                                                drop=self._dp(pdrop, "enc_final"))
        elif lc.model_type == "mbart":
            h, sv["enc_final_ln"] = self.ln_fwd(h, lp + "model.encoder.layer_norm.weight", lp + "model.encoder.layer_norm.bias",
                                                B * S, d, eps)
        enc = h
        sv["enc_out"] = enc
        # ---- decoder
        y = self.new(B * Ld, d)
        ops.embed_fwd(dec_ids, self.W(emb_name), y, B * Ld, d, escale, self.dt)
        sv["dec_ids"] = dec_ids
        Hd, Fd = lc.decoder_attention_heads, lc.decoder_ffn_dim
        if not t5:
            pd = lp + "model.decoder."
            y, sv["dec_emb_ln"] = self.ln_fwd(y, pd + "layernorm_embedding.weight", pd + "layernorm_embedding.bias", B * Ld, d,
                                              eps, pos=self.W(pd + "embed_positions.weight"), pos_period=Ld, pos_offset=2,
                                              want_sum=True, drop=self._dp(pdrop, "dec_emb"))
        else:
            sv["d_dec_in"] = self._dp(pdrop, "dec_in")
            y = self._dropped(y, sv["d_dec_in"], B * Ld * d)
        sv["dec_layers"] = []
        for i in range(lc.decoder_layers):
            nm = self._t5_layer_names("decoder", i) if t5 else self._bart_layer_names("decoder", i)
            self._scope = f"lmd{i}"
            y, lsv = self.layer_fwd(y, B, Ld, d, Hd, Fd, nm, pre_ln, act, eps, causal=True, scale=1.0 if t5 else None,
                                    enc=enc, Tk=S, rms=t5, bias=dbias, drop=drop, enc_klen=enc_klen)
            if self.lm_adapters:
                y, lsv["adapter"] = self.adapter_fwd(y, lc.encoder_layers + i, B * Ld, d)
            sv["dec_layers"].append(lsv)
        self._scope = "lm"
        if t5:
            y, sv["dec_final_ln"] = self.ln_fwd(y, lp + "decoder.final_layer_norm.weight", None, B * Ld, d, eps, rms=True,
                                                drop=self._dp(pdrop, "dec_final"))
        elif lc.model_type == "mbart":
            y, sv["dec_final_ln"] = self.ln_fwd(y, lp + "model.decoder.layer_norm.weight", lp + "model.decoder.layer_norm.bias",
                                                B * Ld, d, eps)
        sv["dec_out"] = y
        # ---- LM head (tied embedding) + final_logits_bias
        V = lc.vocab_size
        Vp = (V + 7) // 8 * 8
        head = lp + "lm_head.weight" if self.has(lp + "lm_head.weight") else emb_name
        sv["head"] = head
        alpha = d ** -0.5 if (t5 and lc.tie_word_embeddings) else 1.0
        sv["head_alpha"] = alpha
        flb = self.st.module.get_buffer(lp + "final_logits_bias").view(-1) if not t5 else None
        sv["flb"], sv["V"], sv["Vp"] = flb, V, Vp
        if not want_logits and self.head_chunk_rows(B * Ld, Vp):
            sv["logits"] = None                    # streamed head: lm_losses / lm_bwd compute them chunk by chunk
            return None, enc, sv
        logits = self.new(B * Ld, Vp, dt=torch.float32)
        self._head_logits(sv, y, logits, B * Ld)
        sv["logits"] = logits
        return logits, enc, sv

    # ------------------------------------------------------------------ greedy decoding with a KV cache (inference)
    def lm_encode(self, inputs_embeds, input_ids, B, S):
        """Text-encoder half of lm_fwd in eval mode, without saved state.  -> encoder output [B*S, d]."""
        lc, lp = self.lc, self.lp
        d = lc.d_model
        t5 = lc.model_type == "t5"
        pre_ln = lc.model_type in ("mbart", "t5")
        act = _act_id(lc.activation_function)
        emb_name = lp + ("shared.weight" if t5 else "model.shared.weight")
        escale = math.sqrt(d) if (lc.scale_embedding and not t5) else 1.0
        if inputs_embeds is None:
            x = self.new(B * S, d)
            ops.embed_fwd(input_ids, self.W(emb_name), x, B * S, d, escale, self.dt)
        else:
            x = inputs_embeds
        if t5:
            if lc.is_gated_act:
                raise NotImplementedError("gated T5 feed-forward is not on the SpeechMix path")
            eps = lc.layer_norm_epsilon
            ebias, _ = self._t5_bias("encoder", S, S)
            h = x
            for i in range(lc.encoder_layers):
                h, _ = self.layer_fwd(h, B, S, d, lc.encoder_attention_heads, lc.encoder_ffn_dim,
                                      self._t5_layer_names("encoder", i), True, act, eps, scale=1.0, rms=True, bias=ebias)
                if self.lm_adapters:
                    h, _ = self.adapter_fwd(h, i, B * S, d)
            h, _ = self.ln_fwd(h, lp + "encoder.final_layer_norm.weight", None, B * S, d, eps, rms=True)
            return h
        eps = 1e-5
        pe = lp + "model.encoder."
        self._check_positions(S=S)
        h, _ = self.ln_fwd(x, pe + "layernorm_embedding.weight", pe + "layernorm_embedding.bias", B * S, d, eps,
                           pos=self.W(pe + "embed_positions.weight"), pos_period=S, pos_offset=2, want_sum=True)
        for i in range(lc.encoder_layers):
            h, _ = self.layer_fwd(h, B, S, d, lc.encoder_attention_heads, lc.encoder_ffn_dim, self._bart_layer_names("encoder", i),
                                  pre_ln, act, eps)
            if self.lm_adapters:
                h, _ = self.adapter_fwd(h, i, B * S, d)
        if pre_ln:
            h, _ = self.ln_fwd(h, pe + "layer_norm.weight", pe + "layer_norm.bias", B * S, d, eps)
        return h

    def greedy_decode(self, enc, B, S, max_new_tokens, start_id, eos_id, pad_id, forced=None, keep_logits=None):
        """Greedy decoding against a fixed encoder output with per-layer K/V caches: the cross-attention K/V are
        projected once, every step projects ONE new token per clip (its K/V land in the cache through the GEMM's output
        view), attends over the cached prefix and takes the arg-max of the LM head.  Replaces the reference's loops that
        re-run the whole decoder - and in the notebook the whole speech encoder - per token (ref:train.py:18-34,
        ref:eval.ipynb cell 6).  BART / mBART (learned positions, LayerNorm, biases) and T5 (RMSNorm, no biases, unscaled
        scores + the decoder's relative-position bias row of the current step, TF:models/t5/modeling_t5.py:216-279,
        logits x d^-0.5 when the head is tied).  Returns int64 [B, n] generated ids (without the start token; rows that
        hit eos are padded with pad_id after it) and the number of steps run.  forced [B, n] int64: feed these tokens
        instead of the arg-max (scoring a given continuation through the cached path); keep_logits: list that receives
        each step's [B, V] logits."""
        lc, lp = self.lc, self.lp
        t5 = lc.model_type == "t5"
        d, H, F = lc.d_model, lc.decoder_attention_heads, lc.decoder_ffn_dim
        pre_ln = lc.model_type in ("mbart", "t5")
        act = _act_id(lc.activation_function)
        hd = (lc.d_kv if t5 else d // H)
        inner = H * hd                                   # == d for every model on the path (T5: heads x d_kv)
        if inner != d:
            raise NotImplementedError("cached decoding assumes num_heads * d_kv == d_model")
        eps = lc.layer_norm_epsilon if t5 else 1e-5
        scale = 1.0 if t5 else hd ** -0.5
        emb_name = lp + ("shared.weight" if t5 else "model.shared.weight")
        escale = math.sqrt(d) if (lc.scale_embedding and not t5) else 1.0
        pd = lp + "model.decoder."
        Lmax = max_new_tokens
        if not t5:
            self._check_positions(Ld=Lmax)
        names = [(self._t5_layer_names if t5 else self._bart_layer_names)("decoder", i) for i in range(lc.decoder_layers)]

        def cat_b(pair_a, pair_b):
            return self.st.cat([pair_a[1], pair_b[1]], "p32") if pair_a[1] else None
        V = lc.vocab_size
        Vp = (V + 7) // 8 * 8
        head = lp + "lm_head.weight" if self.has(lp + "lm_head.weight") else emb_name
        flb = self.st.module.get_buffer(lp + "final_logits_bias").view(-1) if not t5 else None
        head_alpha = d ** -0.5 if (t5 and lc.tie_word_embeddings) else 1.0
        # Decoding state.  Plain calls own theirs; graph-replayed calls (below) share one static set per configuration, because
        # a captured step holds the addresses of everything it touches.
        use_graphs = (forced is None and keep_logits is None and self.dev.type == "cuda" and self.dt == BF16
                      and os.environ.get("SMX_DECODE_GRAPH", "1") != "0")
        key = (B, S, Lmax, start_id, eos_id, pad_id, self.st.master.data_ptr())
        dc = getattr(self, "_decode_cache", None)
        if dc is None:
            dc = self._decode_cache = {}
        stt = dc.get(key) if use_graphs else None
        if stt is None:
            stt = dict(xkv=[self.new(B * S, 2 * d) for _ in names], cache=[self.zeros(B, Lmax, 2 * d) for _ in names],
                       logits=self.new(B, Vp, dt=torch.float32), tok=torch.empty(B, dtype=torch.int64, device=self.dev),
                       nxt=torch.empty(B, dtype=torch.int64, device=self.dev), done=torch.zeros(B, dtype=torch.bool, device=self.dev),
                       out=torch.empty(B, Lmax, dtype=torch.int64, device=self.dev), lse=self.new(B * H, dt=torch.float32),
                       pbias=torch.empty(H, Lmax, Lmax, dtype=torch.float32, device=self.dev) if t5 else None,
                       fin=torch.empty((), dtype=torch.int64, device=self.dev), graphs={}, calls=0, pool=None, eager_done=set(),
                       failures=0)
            stt["bytes"] = sum(t.numel() * t.element_size() for t in stt["xkv"] + stt["cache"] + [stt["logits"]])
            if use_graphs:
                # a handful of configurations at most, and no more than 4 GB of static state (each holds its K/V caches, the
                # cross-attention K/V and its captured steps); S - the LM-encoder length - is part of the key, so variable-length
                # evaluation mostly runs eagerly and the oldest entries make room
                while dc and (len(dc) >= 4 or sum(v["bytes"] for v in dc.values()) + stt["bytes"] > (4 << 30)):
                    dc.pop(next(iter(dc)))
                dc[key] = stt
        xkv, cache, logits, tok, nxt, done, out, lse, pbias, fin = (stt[k] for k in ("xkv", "cache", "logits", "tok", "nxt", "done",
                                                                                      "out", "lse", "pbias", "fin"))
        # cross-attention K/V of every layer, once per call
        for li, nm in enumerate(names):
            kn, vn = nm["xattn"]["k"], nm["xattn"]["v"]
            self.lin(enc, self.st.cat([kn[0], vn[0]]), cat_b(kn, vn), B * S, 2 * d, d, y=xkv[li])
        if t5:
            pbias.copy_(self._t5_bias("decoder", Lmax, Lmax)[0])               # [H, Lmax, Lmax]: row t = step t's bias
        tok.fill_(start_id)
        done.zero_()
        out.fill_(pad_id)
        fin.fill_(-1)                                  # first step after which every clip had emitted eos

        def step(t):
            """Everything step t does on the device: embed the current tokens, the decoder layers against the caches, the head's
            arg-max, and the bookkeeping of `out` / `done` / `tok` (a handful of ints; no host round trip)."""
            y = self.new(B, d)
            ops.embed_fwd(tok, self.W(emb_name), y, B, d, escale, self.dt)
            if not t5:
                y, _ = self.ln_fwd(y, pd + "layernorm_embedding.weight", pd + "layernorm_embedding.bias", B, d, eps,
                                   pos=self.W(pd + "embed_positions.weight"), pos_period=1, pos_offset=2 + t, want_sum=True)
            sbias = pbias[:, t, :t + 1].contiguous() if t5 else None          # [H, 1, t+1]
            for li, nm in enumerate(names):
                def attend(xq, wq, bq, kbuf, k_bs, k_ld, Tk, bias=None):
                    q = self.lin(xq, self.W(wq), self._b(bq), B, d, d)
                    desc = ops.AttnDesc(B, H, 1, Tk, hd, False, scale, bias)
                    desc.set("Q", q, 0, d, d)
                    desc.set("K", kbuf, 0, k_bs, k_ld)
                    desc.set("V", kbuf, d, k_bs, k_ld)
                    o = self.new(B, d)
                    desc.set("O", o, 0, d, d)
                    ops.attention_fwd(desc, lse, self.dt)
                    return o
                a = nm["attn"]
                x0 = y
                xin = self.ln_fwd(x0, nm["ln1"][0], nm["ln1"][1], B, d, eps, rms=t5)[0] if pre_ln else x0
                # this token's K | V straight into the cache row t of every clip
                self.lin(xin, self.st.cat([a["k"][0], a["v"][0]]), cat_b(a["k"], a["v"]), B, 2 * d, d,
                         y=cache[li], cv=view(Lmax * 2 * d, 0, 0, t * 2 * d))
                o = attend(xin, a["q"][0], a["q"][1], cache[li], Lmax * 2 * d, 2 * d, t + 1, sbias)
                s1 = self.lin(o, self.W(a["o"][0]), self._b(a["o"][1]), B, d, d, resid=x0)
                if pre_ln:
                    h = s1
                    xin2 = self.ln_fwd(h, nm["lnx"][0], nm["lnx"][1], B, d, eps, rms=t5)[0]
                else:
                    h = self.ln_fwd(s1, nm["ln1"][0], nm["ln1"][1], B, d, eps)[0]
                    xin2 = h
                xa = nm["xattn"]
                o2 = attend(xin2, xa["q"][0], xa["q"][1], xkv[li], S * 2 * d, 2 * d, S)
                s2 = self.lin(o2, self.W(xa["o"][0]), self._b(xa["o"][1]), B, d, d, resid=h)
                if pre_ln:
                    n2 = self.ln_fwd(s2, nm["ln2"][0], nm["ln2"][1], B, d, eps, rms=t5)[0]
                    y, _ = self._ffn_fwd(n2, B, d, F, nm["fc1"], nm["fc2"], act, s2)
                else:
                    h2 = self.ln_fwd(s2, nm["lnx"][0], nm["lnx"][1], B, d, eps)[0]
                    s3, _ = self._ffn_fwd(h2, B, d, F, nm["fc1"], nm["fc2"], act, h2)
                    y = self.ln_fwd(s3, nm["ln2"][0], nm["ln2"][1], B, d, eps)[0]
                if self.lm_adapters:
                    y, _ = self.adapter_fwd(y, lc.encoder_layers + li, B, d)
            if t5:
                y = self.ln_fwd(y, lp + "decoder.final_layer_norm.weight", None, B, d, eps, rms=True)[0]
            elif pre_ln:
                y = self.ln_fwd(y, pd + "layer_norm.weight", pd + "layer_norm.bias", B, d, eps)[0]
            ops.gemm(y, self.W(head), logits, B, V, d, self.dt, cv=view(Vp), bias=flb, out_f32=True, alpha=head_alpha)
            ops.cross_entropy(logits, None, None, nxt, None, B, V, Vp, Vp, self.dt)
            if forced is not None:
                out[:, t] = nxt
                tok.copy_(forced[:, t])
                return
            # finished rows keep emitting pad
            out[:, t] = torch.where(done, torch.full_like(nxt, pad_id), nxt)
            done.logical_or_(nxt == eos_id)
            tok.copy_(torch.where(done, torch.full_like(nxt, eos_id), nxt))
            fin.copy_(torch.where((fin < 0) & done.all(), torch.full_like(fin, t), fin))

        # Graph replay (SMX_DECODE_GRAPH=0: off): a step is ~100 launches of 5 - 20 us kernels and the host needs ~2 ms to issue
        # them, so the decoder is bound by the launch rate.  The first call of a configuration runs eagerly (kernel picks,
        # one-time launcher set-up); from the second call on, step t is captured once into a HIP graph - positions are launch
        # parameters, so every t has its own graph - and replayed.  The end-of-sequence test costs a host round trip and is made
        # every eighth step in that mode (tokens past eos are pad either way).
        replay = use_graphs and stt["calls"] >= 1
        stt["calls"] += 1
        steps = 0
        for t in range(Lmax):
            if replay and t not in stt["eager_done"] and t not in stt["graphs"]:
                # a step is captured only after it has run eagerly once (a first call that stopped early at eos leaves its later
                # steps' GEMM shapes untuned, and the tuner's synchronize is illegal inside a capture)
                step(t)
                stt["eager_done"].add(t)
            elif replay:
                g = stt["graphs"].get(t)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    gc_was = gc.isenabled()
                    gc.disable()              # (no finaliser may run on a capturing thread: graphs.StepGraphs.capture)
                    try:
                        with torch.cuda.graph(g, pool=stt["pool"], capture_error_mode=os.environ.get("SMX_CAPTURE_MODE", "thread_local")):          # (see graphs.StepGraphs._begin)
                            step(t)
                    except RuntimeError as e:          # capture refused (a runtime without graph support for some call): stay eager
                        import warnings
                        warnings.warn(f"greedy_decode: HIP graph capture failed ({e}); decoding eagerly")
                        stt["graphs"].clear()
                        stt["failures"] += 1
                        # one retry (from the next call on, after this call's eager pass); a second failure disables replay
                        stt["calls"] = 1 if stt["failures"] < 2 else -(1 << 30)
                        replay = False
                        step(t)
                        steps += 1
                        if bool(done.all()):
                            break
                        continue
                    finally:
                        if gc_was:
                            gc.enable()
                    if stt["pool"] is None:
                        stt["pool"] = g.pool()
                    stt["graphs"][t] = g
                g.replay()
            else:
                step(t)
                if use_graphs:
                    stt["eager_done"].add(t)
            steps += 1
            if keep_logits is not None:
                keep_logits.append(logits[:, :V].clone())
            if forced is not None:
                continue
            if (not replay or (t & 7) == 7 or t == Lmax - 1) and bool(done.all()):
                break
        if forced is None:
            f_ = int(fin.item())
            if f_ >= 0:
                steps = f_ + 1                         # (replay mode may have run up to seven steps past it: those emitted pad)
        ids = out[:, :steps].clone()
        return ids, steps

    # ------------------------------------------------------------------ SpeechMixSelf hidden-state matching
    def self_mse(self, enc_s, enc_t, B, S, Lt, want_grad=True):
        """ref:speechmix/model.py:247-255.  enc_s [B*S,d] speech-side LM-encoder output, enc_t [B*Lt,d] text side.
        attn = softmax(bmm(H_t, H_s.view(B,d,-1)) / sqrt(d)); mse = MSE(bmm(attn, H_s), H_t).  The `.view(B,d,-1)`
        is a reinterpretation of H_s's memory (not a transpose) and is reproduced as such.  Tiny matrices: done in
        fp32 with the simple GEMM kernel.  Returns (mse loss [1] fp32, dH_s [B*S,d] fp32 or None)."""
        d = self.lc.d_model
        hs = self.new(B * S, d, dt=torch.float32)
        ht = self.new(B * Lt, d, dt=torch.float32)
        ops.cast_to_f32(enc_s, hs, B * S * d, self.dt)
        ops.cast_to_f32(enc_t, ht, B * Lt * d, self.dt)
        attn = self.new(B * Lt, S, dt=torch.float32)
        scale = 1.0 / math.sqrt(d)
        # A[b] = H_t[b] [Lt,d] @ R[b] [d,S]   with R = H_s[b] memory read as [d,S]
        ops.gemm(ht, hs, attn, Lt, S, d, F32, b_rc=True, bv=view(S), cv=view(S), alpha=scale, nbatch=B,
                 batch_a=Lt * d, batch_b=S * d, batch_c=Lt * S)
        ops.softmax_rows(attn, B * Lt, S)
        proj = self.new(B * Lt, d, dt=torch.float32)
        ops.gemm(attn, hs, proj, Lt, d, S, F32, b_rc=True, av=view(S), bv=view(d), cv=view(d), nbatch=B,
                 batch_a=Lt * S, batch_b=S * d, batch_c=Lt * d)
        loss = self.zeros(1, dt=torch.float32)
        dproj = self.new(B * Lt, d, dt=torch.float32) if want_grad else None
        ops.mse(proj, ht, loss, dproj, B * Lt * d)
        if not want_grad:
            return loss, None
        # d attn = dproj @ H_s^T ; dH_s = attn^T @ dproj ; dA = softmax_bwd(attn, dattn)/sqrt(d) ; dR = H_t^T @ dA
        dattn = self.new(B * Lt, S, dt=torch.float32)
        ops.gemm(dproj, hs, dattn, Lt, S, d, F32, av=view(d), bv=view(d), cv=view(S), nbatch=B, batch_a=Lt * d,
                 batch_b=S * d, batch_c=Lt * S)
        dhs = self.new(B * S, d, dt=torch.float32)
        ops.gemm(attn, dproj, dhs, S, d, Lt, F32, a_rc=True, b_rc=True, av=view(S), bv=view(d), cv=view(d), nbatch=B,
                 batch_a=Lt * S, batch_b=Lt * d, batch_c=S * d)
        dA = self.new(B * Lt, S, dt=torch.float32)
        ops.softmax_rows_bwd(attn, dattn, dA, B * Lt, S, scale)
        ops.gemm(ht, dA, dhs, d, S, Lt, F32, a_rc=True, b_rc=True, av=view(d), bv=view(S), cv=view(S), atomic=2, nbatch=B,
                 batch_a=Lt * d, batch_b=Lt * S, batch_c=S * d)
        return loss, dhs

    def _wait_head_wgrad(self):
        """The tied embedding's gradient is about to receive scatter-added token-embedding gradients on this stream: the head's
        weight gradient (second stream, lm_bwd) must have written it first."""
        ev = getattr(self, "_head_ev", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._head_ev = None

    def lm_bwd(self, dlogits, sv, gscale, extra_denc=None):
        """dlogits [B*Ld, Vp] compute dtype.  Returns grad wrt inputs_embeds [B*S, d] (or None for token input)."""
        lc, lp = self.lc, self.lp
        d, B, S, Ld, t5 = lc.d_model, sv["B"], sv["S"], sv["Ld"], sv["t5"]
        pre_ln = lc.model_type in ("mbart", "t5")
        act = _act_id(lc.activation_function)
        V, Vp, head = sv["V"], sv["Vp"], sv["head"]
        a = sv["head_alpha"] * gscale
        Md, Ms = B * Ld, B * S
        emb_name, escale = sv["emb_name"], sv["escale"]
        lm_trainable = self.tr(head)
        self._ensure_key(sv.get("key"))
        dy = self.new(Md, d)
        if isinstance(dlogits, str):             # streamed head (lm_losses): recompute each chunk's logits, then its three products
            R = self.head_chunk_rows(Md, Vp)
            ws = self.workspace("head_logits", R * Vp, torch.float32)[:R * Vp].view(R, Vp)
            dws = self.workspace("head_dlogits", R * Vp, self.tdt)[:R * Vp].view(R, Vp)
            lab, y = sv["labels_flat"], sv["dec_out"]
            scratch = self.zeros(1, dt=torch.float32)
            am = self.new(R, dt=torch.int64)
            for r0 in range(0, Md, R):
                n = min(R, Md - r0)
                self._head_logits(sv, y[r0:r0 + n], ws, n)
                ops.cross_entropy(ws, lab[r0:r0 + n], scratch, am, dws, n, V, Vp, Vp, self.dt, count_labels=lab)
                if lm_trainable:
                    self.wgrad(dws, y[r0:r0 + n], self.G(head), n, V, d, dyv=view(Vp), alpha=a, side_ok=False)
                self.dgrad(dws, self.W(head), dy[r0:r0 + n], n, V, d, av=view(Vp), alpha=a)
        else:
            # the data gradient first: the decoder's backward chain hangs on it.  The head's weight gradient (dE = dlogits^T y:
            # 79 GF, ~0.4 ms) goes to the second stream like every other weight gradient of the stage (round 5; rounds 1-4 kept it
            # on the main stream, in front of the chain, because the tied embedding also receives the scatter-added gradients of
            # the token embeddings: those launches now wait for `_head_ev`, which has long passed when they come up)
            self.dgrad(dlogits, self.W(head), dy, Md, V, d, av=view(Vp), alpha=a)
            if lm_trainable:
                on_side = getattr(self, "_side_active", False) and os.environ.get("SMX_HEAD_WGRAD_SIDE", "1") != "0"
                self.wgrad(dlogits, sv["dec_out"], self.G(head), Md, V, d, dyv=view(Vp), alpha=a, side_ok=on_side)   # (tied embedding)
                if on_side:
                    self._head_ev = torch.cuda.Event()
                    self._head_ev.record(self._side)
        if t5:
            dy = self.ln_bwd(dy, sv["dec_final_ln"], lp + "decoder.final_layer_norm.weight", None, Md, d, rms=True)
        elif lc.model_type == "mbart":
            dy = self.ln_bwd(dy, sv["dec_final_ln"], lp + "model.decoder.layer_norm.weight", lp + "model.decoder.layer_norm.bias",
                             Md, d)
        denc = [self.new(Ms, d), True]
        # cross-attention K/V gradients of all decoder layers side by side (see attn_bwd) when the layers' k|v weights sit at a
        # uniform stride in the flat store (they do: identical layers), so that ONE rows-contiguous batched view reads them all
        nL = lc.decoder_layers
        kv_names = [(self._t5_layer_names("decoder", i) if t5 else self._bart_layer_names("decoder", i))["xattn"] for i in range(nL)]
        koff = [self.st.offsets[n["k"][0]][0] for n in kv_names]
        uniform = nL > 1 and all(koff[i + 1] - koff[i] == koff[1] - koff[0] for i in range(nL - 1)) and \
            os.environ.get("SMX_XATTN_CAT") != "0"
        if uniform:
            denc.append(dict(buf=self.new(Ms, nL * 2 * d), layer=0, ld=nL * 2 * d))
        # T5: every layer of a stack adds the stack's relative-position bias (owned by block 0) to its self-attention
        # scores, so the table's gradient is the scatter of the score gradients summed over batch AND layers
        tb = {}
        if t5:
            for side, T in (("encoder", S), ("decoder", Ld)):
                tname = f"{lp}{side}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"
                if self.tr(tname):
                    tb[side] = (tname, self.zeros(lc.encoder_attention_heads, T, T, dt=torch.float32), T)
        for i in range(lc.decoder_layers - 1, -1, -1):
            nm = self._t5_layer_names("decoder", i) if t5 else self._bart_layer_names("decoder", i)
            if "adapter" in sv["dec_layers"][i]:
                dy = self.adapter_bwd(dy, sv["dec_layers"][i]["adapter"], lc.encoder_layers + i, Md, d)
            if uniform:
                denc[2]["layer"] = i
            side = self._side if getattr(self, "_side_active", False) else None
            if side is not None:
                self._wg_begin(rows=Md)
            self._wg_defer_begin()
            dy = self.layer_bwd(dy, sv["dec_layers"][i], nm, pre_ln, act, rms=t5, denc=denc,
                                dbias=tb["decoder"][1] if "decoder" in tb else None)
            if side is not None:
                self._wg_flush(side=side)
            self._wg_defer_flush()
            denc[1] = False
        if "decoder" in tb:
            self._t5_bias_bwd(tb["decoder"], sv["t5_buckets"][1])
        if uniform:          # d enc = [dK|dV of layer 0 | ... | layer L-1] @ [Wk; Wv of layer 0; ...]: one GEMM, K = L * 2d
            wk0 = self.st.cat([kv_names[0]["k"][0], kv_names[0]["v"][0]])
            self.dgrad(denc[2]["buf"], wk0, denc[0], Ms, nL * 2 * d, d, bv=view(d, 2 * d, koff[1] - koff[0], 0))
        if not t5:
            pd = lp + "model.decoder."
            pos_n = pd + "embed_positions.weight"
            dy = self.ln_bwd(dy, sv["dec_emb_ln"], pd + "layernorm_embedding.weight", pd + "layernorm_embedding.bias", Md, d,
                             dpos=self.G(pos_n) if self.tr(pos_n) else None, pos_period=Ld, pos_offset=2)
        if t5:
            dy = self._dropped(dy, sv["d_dec_in"], Md * d)
        if self.tr(emb_name):
            self._wait_head_wgrad()
            ops.embed_bwd(sv["dec_ids"], dy, self.G(emb_name), Md, d, escale, self.dt)
        # ---- text encoder
        dh = denc[0]
        if extra_denc is not None:      # SpeechMixSelf: gradient of the hidden-state matching loss (fp32)
            ops.add_f32_into(extra_denc, dh, Ms * d, self.dt)
        if t5:
            dh = self.ln_bwd(dh, sv["enc_final_ln"], lp + "encoder.final_layer_norm.weight", None, Ms, d, rms=True)
        elif lc.model_type == "mbart":
            dh = self.ln_bwd(dh, sv["enc_final_ln"], lp + "model.encoder.layer_norm.weight", lp + "model.encoder.layer_norm.bias",
                             Ms, d)
        for i in range(lc.encoder_layers - 1, -1, -1):
            nm = self._t5_layer_names("encoder", i) if t5 else self._bart_layer_names("encoder", i)
            if "adapter" in sv["enc_layers"][i]:
                dh = self.adapter_bwd(dh, sv["enc_layers"][i]["adapter"], i, Ms, d)
            self._wg_defer_begin()
            dh = self.layer_bwd(dh, sv["enc_layers"][i], nm, pre_ln, act, rms=t5,
                                dbias=tb["encoder"][1] if "encoder" in tb else None)
            self._wg_defer_flush()
        if "encoder" in tb:
            self._t5_bias_bwd(tb["encoder"], sv["t5_buckets"][0])
        if not t5:
            pe = lp + "model.encoder."
            pos_n = pe + "embed_positions.weight"
            dh = self.ln_bwd(dh, sv["enc_emb_ln"], pe + "layernorm_embedding.weight", pe + "layernorm_embedding.bias", Ms, d,
                             dpos=self.G(pos_n) if self.tr(pos_n) else None, pos_period=S, pos_offset=2)
        if t5:
            dh = self._dropped(dh, sv["d_enc_in"], Ms * d)
        if sv["enc_ids"] is not None:
            if self.tr(emb_name):
                self._wait_head_wgrad()
                ops.embed_bwd(sv["enc_ids"], dh, self.G(emb_name), Ms, d, escale, self.dt)
            return None
        return dh

    # ------------------------------------------------------------------ whole step
    def lm_losses(self, e, dec_ids, labels, B, S, Ld, text_ids=None, training=False, want_grad=True, enc_klen=None,
                  want_logits=True):
        # `training` here is the LM's own mode (dropout): SpeechMixSelf keeps the LM in eval (ref:speechmix/model.py:239)
        """LM on `inputs_embeds` e [B*S,d] (+ optional SpeechMixSelf teacher pass on text_ids) -> losses and dlogits.
        Plain: CE (ref:speechmix/model.py:132-137).  Self: CE + KLD(batchmean) + MSE (ref:speechmix/model.py:235-266)."""
        stream_ok = text_ids is None and not want_logits            # (SpeechMixSelf's KLD needs both logit sets at once)
        logits, enc, lsv = self.lm_fwd(e, None, dec_ids.reshape(-1).contiguous(), B, S, Ld, training, enc_klen=enc_klen,
                                       want_logits=not stream_ok)
        V, Vp = lsv["V"], lsv["Vp"]
        M = B * Ld
        argmax = self.new(M, dt=torch.int64)
        if logits is None:
            # streamed head: chunk -> logits -> CE / arg-max -> dropped; backward recomputes the chunk (lm_bwd)
            R = self.head_chunk_rows(M, Vp)
            ws = self.workspace("head_logits", R * Vp, torch.float32)[:R * Vp].view(R, Vp)
            lab = labels.reshape(-1).contiguous() if labels is not None else None
            ce = self.zeros(1, dt=torch.float32) if lab is not None else None
            y = lsv["dec_out"]
            for r0 in range(0, M, R):
                n = min(R, M - r0)
                self._head_logits(lsv, y[r0:r0 + n], ws, n)
                ops.cross_entropy(ws, lab[r0:r0 + n] if lab is not None else None, ce, argmax[r0:r0 + n], None, n, V, Vp, Vp,
                                  self.dt, count_labels=lab)
            out = dict(logits=None, lm_enc_last=enc, argmax=argmax.view(B, Ld), lsv=lsv, loss=ce, dlogits=None, extra_denc=None)
            if lab is not None:
                out.update(ce=ce, dlogits="streamed" if want_grad else None)
                lsv["labels_flat"] = lab
            return out
        out = dict(logits=logits, lm_enc_last=enc, argmax=argmax.view(B, Ld), lsv=lsv, loss=None, dlogits=None, extra_denc=None)
        if labels is None:
            ops.cross_entropy(logits, None, None, argmax, None, M, V, Vp, Vp, self.dt)
            return out
        lab = labels.reshape(-1).contiguous()
        ce = self.zeros(1, dt=torch.float32)
        dlogits = self.new(M, Vp) if want_grad else None
        if text_ids is None:
            ops.cross_entropy(logits, lab, ce, argmax, dlogits, M, V, Vp, Vp, self.dt)
            out.update(loss=ce, ce=ce, dlogits=dlogits)
            return out
        Lt = text_ids.shape[1]
        logits_t, enc_t, _ = self.lm_fwd(None, text_ids.reshape(-1).contiguous(), dec_ids.reshape(-1).contiguous(), B, Lt, Ld,
                                         False)
        kld = self.zeros(1, dt=torch.float32)
        ops.cross_entropy(logits, lab, ce, argmax, dlogits, M, V, Vp, Vp, self.dt, logits_t=logits_t, kld=kld,
                          kld_scale=1.0 / B)
        mse, dhs = self.self_mse(enc, enc_t, B, S, Lt, want_grad=want_grad)
        out.update(loss=kld + ce + mse, ce=ce, kld=kld, mse=mse, dlogits=dlogits, extra_denc=dhs)
        return out

    def speech_side_fwd(self, wave, training=False, prompt_ids=None, weighted_sum=False, sample_lengths=None):
        """Everything ahead of the LM: speech encoder -> (layer-weighted sum) -> length adapters -> enc_to_dec_proj ->
        (text-prompt embeddings prepended).  -> (inputs_embeds [B*S, d_lm], S, state for speech_side_bwd, extras).
        sample_lengths (see speech_fwd): extras["lm_lengths"] then holds each clip's valid LM-encoder positions - the frame
        length pushed through every Conv1d(k=2, s=2) length adapter (the conv arithmetic of TF:...wav2vec2.py:997-1036), plus
        the prompt."""
        B, N = wave.shape
        self.mark("fwd:start")
        x, ssv = self.speech_fwd(wave, B, N, training, sample_lengths)
        self.mark("fwd:speech")
        T, d = ssv["T"], self.ec.hidden_size
        ws = None
        xin = x
        if weighted_sum:
            hidden = ssv["hidden"]
            sw = self.new(len(hidden), dt=torch.float32)
            xin = self.new(B * T, d)
            ops.weighted_sum_fwd(hidden, self.P("weights_sum"), xin, sw, B * T * d, self.dt)
            ws = sw
        e, S, bsv = self.bridge_fwd(xin, B, T)
        P = 0
        if prompt_ids is not None:
            lc = self.lc
            P = prompt_ids.numel()
            dd = lc.d_model
            t5 = lc.model_type == "t5"
            emb_name = self.lp + ("shared.weight" if t5 else "model.shared.weight")
            escale = math.sqrt(dd) if (lc.scale_embedding and not t5) else 1.0
            pe = self.new(P, dd)
            ops.embed_fwd(prompt_ids, self.W(emb_name), pe, P, dd, escale, self.dt)
            # concatenation along time is pure data movement
            e = torch.cat((pe.view(1, P, dd).expand(B, P, dd), e.view(B, S, dd)), 1).contiguous().view(B * (P + S), dd)
            S = S + P
        state = dict(speech=ssv, bridge=bsv, B=B, ws=ws, P=P, prompt_ids=prompt_ids, key=getattr(self, "_key", None) if training else None)
        lm_lengths = None
        if ssv.get("frame_len") is not None:
            lm_lengths = []
            for n in ssv["frame_len"]:
                for _ in range(self.downloop):
                    n = max((n - 2) // 2 + 1, 1)
                lm_lengths.append(n + P)
        extras = dict(enc_last=x, T=T, post_adapter=bsv["post_adapter"], hidden=ssv["hidden"], sw=ws, lm_lengths=lm_lengths)
        return e, S, state, extras

    def speech_side_bwd(self, de, sv):
        """de: gradient wrt inputs_embeds [B*S, d_lm] (prompt rows included) -> all gradients ahead of the LM."""
        self._ensure_key(sv.get("key"))
        P = sv.get("P", 0)
        if P:
            lc, B = self.lc, sv["B"]
            dd = lc.d_model
            S2 = de.shape[0] // B
            de3 = de.view(B, S2, dd)
            t5 = lc.model_type == "t5"
            emb_name = self.lp + ("shared.weight" if t5 else "model.shared.weight")
            if self.tr(emb_name):
                escale = math.sqrt(dd) if (lc.scale_embedding and not t5) else 1.0
                dp = de3[:, :P].contiguous().view(B * P, dd)
                self._wait_head_wgrad()
                ops.embed_bwd(sv["prompt_ids"].repeat(B).contiguous(), dp, self.G(emb_name), B * P, dd, escale, self.dt)
            de = de3[:, P:].contiguous().view(B * (S2 - P), dd)
        self._stage("lm")
        dx = self.bridge_bwd(de, sv["bridge"])
        ws = None
        if sv.get("ws") is not None:
            ssv = sv["speech"]
            hidden = ssv["hidden"]
            n = ssv["B"] * ssv["T"] * self.ec.hidden_size
            dots = self.new(len(hidden), dt=torch.float32)
            if self.tr("weights_sum"):
                ops.weighted_sum_bwd(hidden, self.P("weights_sum"), dx, dots, self.G("weights_sum"), sv["ws"], n, self.dt)
            ws = (dx, sv["ws"])
        self._stage("bridge")
        self.speech_bwd(dx, sv["speech"], ws=ws)
        self._stage("frontend")

    def forward(self, wave, dec_ids, labels, training=False, prompt_ids=None, text_ids=None, weighted_sum=False,
                lm_training=None, sample_lengths=None, lm_mask=True, want_logits=True):
        """wave [B,N] fp32 cuda; dec_ids [B,Ld] int64; labels [B,Ld] int64 or None; text_ids [B,Lt] (SpeechMixSelf);
        prompt_ids [P] int64: token ids of a text prompt whose embeddings are prepended to every clip
        (ref:speechmix/model.py:168-171, batch-expanded like ref:speechmix/hf_model.py:433-436)."""
        self.st.refresh_shadow()
        B = wave.shape[0]
        Ld = dec_ids.shape[1]
        self.begin_pass(training or bool(lm_training))
        e, S, state, ex = self.speech_side_fwd(wave, training, prompt_ids, weighted_sum, sample_lengths)
        self.mark("fwd:bridge")
        # sample_lengths: the speech encoder's padding mask; lm_mask: also mask the padded positions as LM-encoder keys (what HF's
        # SpeechEncoderDecoderModel does with the reduced mask; the reference's own forward passes no mask to the LM)
        enc_klen = None
        if ex.get("lm_lengths") is not None and lm_mask:
            enc_klen = self.h2d(np.asarray(ex["lm_lengths"], dtype=np.int32))
        lo = self.lm_losses(e, dec_ids, labels, B, S, Ld, text_ids=text_ids,
                            training=training if lm_training is None else lm_training, enc_klen=enc_klen, want_logits=want_logits)
        self.mark("fwd:lm")
        self.saved = dict(state, lm=lo["lsv"], dlogits=lo["dlogits"], extra_denc=lo["extra_denc"], Ld=Ld)
        return dict(loss=lo["loss"], argmax=lo["argmax"], logits=lo["logits"], enc_last=ex["enc_last"],
                    lm_enc_last=lo["lm_enc_last"], inputs_embeds=e, S=S, T=ex["T"], post_adapter=ex["post_adapter"],
                    hidden=ex["hidden"], sw=ex["sw"], parts={k: lo[k] for k in ("ce", "kld", "mse") if k in lo})

    def backward(self, gscale=1.0, zero_grads=True):
        ops.IN_BACKWARD = True           # kernel choice: see ops.PP_CONCURRENT_BACKWARD_OK
        try:
            self._backward(gscale, zero_grads)
            if self.folds is not None:
                self.folds.flush()       # anything queued after the last reported stage
        finally:
            ops.IN_BACKWARD = False
            if self.folds is not None:
                self.folds.items.clear()

    def wait_params(self):
        """The parameters beyond the front end may still be under the previous optimizer step's tail on a second stream (trainer.py,
        SMX_OPT_OVERLAP): the current stream waits for it here - before the first encoder layer of a forward, and wherever else parameters
        are read."""
        ev = getattr(self, "param_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.param_event = None

    def note_dropped(self, zeroed):
        """Called after every backward (eager or replayed) with `last_dropped` of its forward."""
        cur = set(self.last_dropped)
        self.dropped_since_zero = cur if zeroed else (self.dropped_since_zero & cur)

    def reset_side_state(self):
        """Drop whatever a backward that did not finish (an aborted capture pass) left queued: deferred weight gradients / column sums,
        the second-stream flags, pending events and the folds - nothing of it may reach the next eager step."""
        self._side_active = False
        ops.GEMM_CONCURRENT = False
        for name in ("_wg_defer", "_wg_group", "_wg_rows", "_cs_defer"):
            v = getattr(self, name, None)
            if isinstance(v, (list, dict)):
                v.clear()
        for name in ("_cs_pending", "_head_ev"):
            if hasattr(self, name):
                setattr(self, name, None)
        if self.folds is not None:
            self.folds.items.clear()
        self.saved = None

    def lm_side_bwd(self, dlogits, lsv, gscale, extra_denc=None):
        """LM backward with the LM stage's weight gradients on the second compute stream (joined by the next _stage)."""
        if os.environ.get("SMX_LM_WGRAD_STREAM") != "0" and self.st.device.type == "cuda":
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream()
            self._side_active = True
            ops.GEMM_CONCURRENT = True
        return self.lm_bwd(dlogits, lsv, gscale, extra_denc=extra_denc)

    def _backward(self, gscale, zero_grads):
        sv = self.saved
        if sv is None or sv["dlogits"] is None:
            raise RuntimeError("backward() needs a forward() with labels")
        self.begin_grads(zero=zero_grads)
        extra = sv.get("extra_denc")
        if extra is not None and gscale != 1.0:
            extra = extra * gscale
        de = self.lm_side_bwd(sv["dlogits"], sv["lm"], gscale, extra_denc=extra)
        self.speech_side_bwd(de, sv)
        self.end_grads()
        self.note_dropped(zero_grads)
        self.saved = None
