"""Thin, validating Python wrappers over the C-ABI kernels in libspeechmix_hip.so.

Every function takes torch tensors only as *device memory handles* (data_ptr + shape); all arithmetic
happens in the hand-written HIP kernels.  There is no PyTorch/CPU fallback: a missing library, a
non-GPU tensor or a non-zero return code raises.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, BF16, F32  # noqa: F401

_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def torch_dtype(dt):
    return _TORCH_DT[dt]


def _stream():
    """Raw handle of PyTorch's CURRENT stream on the CURRENT device.  `torch.cuda.current_stream().cuda_stream` costs ~8 us of
    Python per call (device-index normalisation, a Stream object) - 7 ms of host time per training step at ~830 launches; the two
    C accessors return the same value in ~0.5 us.  The device is re-read on every call: a launch that precedes the rank's
    `torch.cuda.set_device(local_rank)` (the reference's train.py builds the model before TrainingArguments / Trainer pick the
    device) must not pin later launches to device 0's stream."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("speechmix_amd kernels need GPU tensors (no CPU fallback)")
    return t.data_ptr()


def view(ld, rows_per_batch=0, batch_stride=0, off=0):
    return L.RowView(batch_stride, ld, off, rows_per_batch, 0)


def _gemm_params(a, b, c, M, N, K, a_rc=False, b_rc=False, av=None, bv=None, cv=None, bias=None, resid=None,
                 aux_out=None, aux_in=None, act=ACT_NONE, out_f32=False, atomic=False, split_k=1, alpha=1.0, nbatch=1,
                 batch_a=0, batch_b=0, batch_c=0, batch_bias=0, tr_mode=1, ev=None, batch_e=None, split_stride=0, drop=None):
    p = L.GemmParams()
    p.A, p.B, p.C = _ptr(a), _ptr(b), _ptr(c)
    p.bias, p.resid, p.aux_out, p.aux_in = _ptr(bias), _ptr(resid), _ptr(aux_out), _ptr(aux_in)
    p.a = av if av is not None else view(K if not a_rc else M)
    p.b = bv if bv is not None else view(K if not b_rc else N)
    p.c = cv if cv is not None else view(N)
    p.e = ev if ev is not None else p.c
    p.batch_a, p.batch_b, p.batch_c, p.batch_bias = batch_a, batch_b, batch_c, batch_bias
    p.batch_e = batch_c if batch_e is None else batch_e
    p.M, p.N, p.K, p.a_rc, p.b_rc = M, N, K, int(a_rc), int(b_rc)
    p.act, p.out_f32, p.atomic = act, int(out_f32), int(atomic)
    p.nbatch, p.split_k, p.tr_mode, p.alpha = nbatch, split_k, tr_mode, alpha
    p.split_stride = split_stride
    if drop is not None and drop[0] > 0:
        p.drop_p, p.drop_seed = drop
    return p


# ---- kernel choice for bf16 GEMMs --------------------------------------------------------------------------------
# Two hand-written kernels serve every bf16 GEMM: the 128x128 LDS-DMA kernel at 4 workgroups/CU (tr_mode 1) and the
# persistent 256x256 ping-pong kernel (tr_mode 8, csrc/gemm_pp.hip).  Which one is faster depends on the shape (tile
# quantisation over 256 CUs, K tiles per output tile, epilogue weight), so the first launch of every distinct
# (layout, shape, epilogue) key times both on the live operands and the winner is cached for the rest of the process.
# SMX_GEMM_PP = auto (default) | 0 (128x128 only) | 1 (ping-pong whenever it is applicable).
PP_MODE = os.environ.get("SMX_GEMM_PP", "auto")
_TUNED = {}
# Kernel picks decide the K split and with it the fp32 summation order: bf16 results are bit-reproducible across processes only
# with the same picks, and a bench line is reproducible across boxes only with the same kernels.  Round 4: the picks of the
# benchmarked configurations ship IN-TREE (speechmix_amd/tune/mi355x_picks.json, measured on an MI355X with tools/make_picks.sh) and
# are loaded by default; keys the file does not hold are tuned live (medians of interleaved launches, below).  SMX_TUNE=live
# ignores the shipped file; SMX_TUNE_FILE=path loads picks from `path` (on top of the shipped ones unless SMX_TUNE=live) and
# writes every pick made by this process back at exit (JSON, keys as repr strings); tuner_state() / load_tuner_state() carry
# them between ranks (StepRunner broadcasts rank 0's).
_TUNE_FILE = os.environ.get("SMX_TUNE_FILE", "")
_TUNE_SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune", "mi355x_picks.json")
_TUNE_DIRTY = False
TUNE_LIVE_KEYS = []          # keys that had to be tuned live in this process (bench.py reports their count)


def _tuned_get(key):
    return _TUNED.get(repr(key))


def _tuned_set(key, val):
    global _TUNE_DIRTY
    _TUNED[repr(key)] = val
    _TUNE_DIRTY = True


def tuner_state():
    """Every pick made so far (GEMM kernel per launch key, weight-gradient candidates): {repr(key): pick}."""
    return dict(_TUNED)


def load_tuner_state(state):
    _TUNED.update(state)


def _tune_file_load():
    import json
    paths = [] if os.environ.get("SMX_TUNE", "") == "live" else [_TUNE_SHIPPED]
    if _TUNE_FILE:
        paths.append(_TUNE_FILE)
    for path in paths:
        if os.path.exists(path):
            with open(path) as f:
                _TUNED.update({k: (tuple(v) if isinstance(v, list) else v) for k, v in json.load(f).items()
                               if not k.startswith("_")})


def _tune_file_save():
    if _TUNE_FILE and _TUNE_DIRTY:
        import json
        tmp = _TUNE_FILE + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(_TUNED, f, indent=0, sort_keys=True)
        os.replace(tmp, _TUNE_FILE)


_tune_file_load()
import atexit as _atexit
_atexit.register(_tune_file_save)
_TUNE_MARGIN = float(os.environ.get("SMX_TUNE_MARGIN", "1.03"))     # a challenger must beat the 128x128 kernel by this factor
# The ping-pong kernel needs a whole CU per workgroup (160 KB of LDS, the full register file).  A kernel running beside it
# - RCCL's all-reduce of the gradient buckets on the side stream during backward - takes CUs away, and the displaced
# workgroups then run as a second round (up to 2x the launch time), where the 4-workgroup/CU kernel only loses a quarter of
# those CUs.  When gradients are all-reduced concurrently (world size > 1, StepRunner sets PP_BACKWARD_CUS) the ping-pong
# launches of backward therefore leave CUs free for RCCL: their persistent grid and their K-slice counts are sized for
# PP_BACKWARD_CUS workgroups (216: what an encoder layer's grouped weight-gradient launch uses anyway), so they finish in
# one round as long as RCCL holds no more than 256 - 216 = 40 CUs; forward launches (nothing runs beside them) use every CU.
# SMX_PP_BACKWARD=0 takes the ping-pong kernel out of backward altogether (the round-1 policy; cost on one GPU: DESIGN.md
# section 5), SMX_PP_BACKWARD_CUS=n forces the reserve on a single GPU, SMX_GEMM_PP=1 overrides both.
PP_CONCURRENT_BACKWARD_OK = os.environ.get("SMX_PP_BACKWARD", "1") != "0"
PP_BACKWARD_CUS = int(os.environ.get("SMX_PP_BACKWARD_CUS", "0"))      # 0: all CUs
PP_RESERVED_DEFAULT = 216
IN_BACKWARD = False


def pp_cus():
    """Workgroups a ping-pong launch may occupy right now (0: every CU)."""
    return PP_BACKWARD_CUS if (IN_BACKWARD and PP_BACKWARD_CUS > 0) else 0


def pp_allowed():
    return PP_MODE != "0" and (PP_MODE == "1" or PP_CONCURRENT_BACKWARD_OK or not IN_BACKWARD)


ACT_SAVE_GRAD = 0x100     # SMX_ACT_SAVE_GRAD: the GEMM's side tensor holds the epilogue's local derivative (smx_common.h)


def _pp_applicable(p, dtype):
    if p.act & ACT_SAVE_GRAD:
        # saved-derivative side tensors: the forward ACT class of a (KC, KC) launch and the bias-free ACTGRAD class of a (KC, RC)
        # data gradient, aligned views only (csrc/gemm_common.h pp_saved_ok)
        if p.a_rc or p.atomic or p.out_f32 or p.resid or (p.N & 7) or ((p.c.ld | p.c.off | p.e.ld | p.e.off) & 7):
            return False
        if not ((not p.b_rc and not p.aux_in) or (p.b_rc and p.aux_in and not p.aux_out and not p.bias)):
            return False
    if dtype != BF16 or p.atomic == 1 or p.M < 256 or p.N < 64 or max(p.M, p.N, p.K) >= (1 << 22):
        return False
    if (p.K & 7) and not (p.a_rc and p.b_rc):
        return False
    # rows-contiguous operands through a batched view: (RC, RC) (conv weight gradients) and the B operand of (KC, RC)
    # (conv data gradients)
    if p.a_rc and not p.b_rc and p.a.rows_per_batch > 0:
        return False
    kst = (p.K + 63) // 64
    per = (kst + p.split_k - 1) // p.split_k
    if (p.split_k - 1) * per >= kst:
        return False
    items = ((p.M + 255) // 256) * ((p.N + 255) // 256) * p.nbatch * p.split_k
    return items >= 96                     # fewer work items than that cannot occupy the chip with one workgroup per CU


WS_MODE = os.environ.get("SMX_GEMM_WS", "auto")       # wave-specialised 192x256 kernel (tr_mode 14, csrc/gemm_ws.hip): auto | 0
FR_MODE = os.environ.get("SMX_GEMM_FR", "auto")       # free-running 256x256 schedule (tr_mode 12, csrc/gemm_fr.hip): auto | 0 | 1


def _fr_applicable(p, dtype):
    """tr_mode 12 runs the ping-pong kernel's tile under the free-running schedule; instantiated for the (KC, KC), (KC, RC)
    and (RC, RC) layouts without batched views of rows-contiguous operands (those fall back to the ping-pong kernel)."""
    if FR_MODE == "0" or not _pp_applicable(p, dtype):
        return False
    if (p.a_rc and not p.b_rc) or (p.a_rc and p.a.rows_per_batch > 0) or (p.b_rc and p.b.rows_per_batch > 0):
        return False
    return True


# SMX_GEMM_BYTES_LOG=path: the ALGORITHMIC bytes (_gemm_bytes) of every bf16 GEMM launch of the process, in launch order, written
# as a JSON list at exit.  Every smx_gemm / smx_gemm_group call dispatches exactly one `gemm_bf16_*` kernel, so the list lines up
# with the dispatch order of a rocprofv3 counter pass of the same command: tools/pmc_traffic.py puts the two side by side per
# kernel instantiation (measured HBM traffic / algorithmic bytes = the waste ratio).
_BYTES_LOG_PATH = os.environ.get("SMX_GEMM_BYTES_LOG", "")
_BYTES_LOG = [] if _BYTES_LOG_PATH else None


def _bytes_log_save():
    if _BYTES_LOG is not None:
        import json
        with open(_BYTES_LOG_PATH, "w") as f:
            json.dump(_BYTES_LOG, f)


_atexit.register(_bytes_log_save)


def _launch(p, dtype):
    if (p.tr_mode & 255) in (8, 12, 13, 14) and pp_cus():
        p.tr_mode = (p.tr_mode & 0xffff) | (pp_cus() << 16)          # persistent grid cap (gemm_pp.hip)
    if _BYTES_LOG is not None and dtype == BF16:
        _BYTES_LOG.append(_gemm_bytes(p, dtype))
    L.check(L.lib().smx_gemm(C.byref(p), dtype, _stream()), "smx_gemm")


TUNE_ROUNDS = int(os.environ.get("SMX_TUNE_ROUNDS", "5"))


CAPTURING = False        # a graphs.StepGraphs capture pass is running: nothing may synchronise or time


class CaptureAbort(RuntimeError):
    """Raised when something that cannot be captured (live tuning, a host-to-device copy) is met during a capture pass."""


def measure_candidates(runs, rounds=None, reps=2):
    """Live timing of alternative launches of ONE piece of work: runs = {candidate: callable that enqueues it}.  Every candidate
    is warmed once, then `rounds` interleaved rounds time `reps` back-to-back launches of each; a candidate's figure is the
    MEDIAN of its samples after dropping those above twice its fastest (the ~1.8-ms single-launch outliers of this pool,
    DESIGN.md 6b).  A candidate whose launch raises RuntimeError (a class the variant is not instantiated for) is left out.
    -> {candidate: milliseconds per launch}."""
    if CAPTURING:
        raise CaptureAbort("a kernel pick is missing (live tuning cannot run inside a stream capture)")
    rounds = rounds or TUNE_ROUNDS
    ok = []
    for c, run in runs.items():
        try:
            run()
            ok.append(c)
        except RuntimeError:
            continue
    torch.cuda.synchronize()
    samples = {c: [] for c in ok}
    evs = {c: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rounds)] for c in ok}
    for r in range(rounds):
        for c in ok:
            e0, e1 = evs[c][r]
            e0.record()
            for _ in range(reps):
                runs[c]()
            e1.record()
    torch.cuda.synchronize()
    out = {}
    for c in ok:
        ts = sorted(e0.elapsed_time(e1) / reps for e0, e1 in evs[c])
        ts = [t for t in ts if t <= 2.0 * ts[0]]
        out[c] = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    return out


def _run_mode(p, dtype, mode):
    def run():
        p.tr_mode = mode
        _launch(p, dtype)
    return run


HALF_MODE = os.environ.get("SMX_GEMM_HALF", "auto")      # 64 x 128 tiles of the 128x128 kernel (tr_mode 9): auto | 0 | 1


def _half_applicable(p, dtype):
    """tr_mode 9 covers K-contiguous A operands, bf16 outputs of the aligned epilogue classes, no split-K; it is only worth
    timing when the 128-row tiling leaves a good part of the 1024 resident workgroup slots empty."""
    if HALF_MODE == "0" or dtype != L.BF16 or p.a_rc or p.split_k != 1 or p.out_f32 or p.atomic:
        return False
    if (p.N & 7) or ((p.c.ld | p.c.off | p.e.ld | p.e.off) & 7) or p.M < 128:
        return False
    tiles = ((p.M + 127) // 128) * ((p.N + 127) // 128) * max(p.nbatch, 1)
    return tiles < 640


W8_MODE = os.environ.get("SMX_GEMM_W8", "auto")       # 256 x 128 tiles, eight waves, two workgroups / CU (tr_mode 11): auto | 0 | 1


def _w8_applicable(p, dtype):
    """tr_mode 11 covers K-contiguous A operands, bf16 outputs of the aligned epilogue classes, no split-K; it needs enough
    256 x 128 tiles to occupy the 512 resident workgroup slots."""
    if W8_MODE == "0" or dtype != L.BF16 or p.a_rc or p.split_k != 1 or p.out_f32 or p.atomic:
        return False
    if (p.N & 7) or ((p.c.ld | p.c.off | p.e.ld | p.e.off) & 7):
        return False
    tiles = ((p.M + 255) // 256) * ((p.N + 127) // 128) * max(p.nbatch, 1)
    return tiles >= 360


def _choose_mode(p, dtype):
    """-> tr_mode for this launch: 1 (128x128), 8 (ping-pong), 9 (64x128) or 11 (256x128, eight waves)."""
    cands = [1]
    if pp_allowed() and _pp_applicable(p, dtype):
        if PP_MODE == "1":
            return 8
        cands.append(8)
        if _fr_applicable(p, dtype):
            if FR_MODE == "1":
                return 12
            cands.append(12)
            if p.M >= 1024 and p.split_k == 1:          # 192 x 256 tiles: better quantisation of N = 768 / 2304 at 16 k rows
                cands.append(13)
                if WS_MODE != "0":                     # the same tile with twelve compute + four loader waves (csrc/gemm_ws.hip)
                    cands.append(14)
    if _half_applicable(p, dtype):
        if HALF_MODE == "1":
            return 9
        cands.append(9)
    if _w8_applicable(p, dtype):
        if W8_MODE == "1":
            return 11
        cands.append(11)
    if len(cands) == 1:
        return 1
    key = (tuple(cands), pp_cus(), p.a_rc, p.b_rc, p.M, p.N, p.K, p.nbatch, p.split_k, bool(p.bias), bool(p.resid), bool(p.aux_out), bool(p.aux_in),
           p.act, p.out_f32, p.atomic, p.drop_p > 0, p.a.rows_per_batch > 0, p.b.rows_per_batch > 0, p.c.rows_per_batch > 0)
    mode = _tuned_get(key)
    if mode is None:
        # re-running the launch must not change the result: no accumulation into C, no side input aliasing the output
        safe = p.atomic == 0 and p.resid != p.C and p.aux_in != p.C and p.A != p.C and p.B != p.C
        mode = 1
        if safe:
            times = measure_candidates({m: _run_mode(p, dtype, m) for m in cands})
            mode = min(times, key=lambda m: times[m] * (1.0 if m == 1 else _TUNE_MARGIN))      # ties go to the 128x128 kernel
            TUNE_LIVE_KEYS.append(key)
            if TUNE_LOG is not None:
                TUNE_LOG.append((key, times.get(1), times.get(8), mode, times.get(9), times.get(11), times.get(12), times.get(13), times.get(14)))
        _tuned_set(key, mode)
    return mode


TUNE_LOG = None      # set to a list to record (key, ms_128, ms_pp, choice) of every tuning decision


def gemm(a, b, c, M, N, K, dtype, **kw):
    """C[M,N] (+)= epi(alpha * A B^T).  a/b/c are tensors (base pointers); av/bv/cv are RowViews in elements.
    drop = (p, seed): dropout after the activation (before the residual add); mask index = m * N + n.
    tr_mode: kernel variant; omit it to let the bf16 path choose between the 128x128 and the ping-pong kernel."""
    p = _gemm_params(a, b, c, M, N, K, **kw)
    if "tr_mode" not in kw:
        p.tr_mode = _choose_mode(p, dtype)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = prof.events()
        e0.record()
        _launch(p, dtype)
        e1.record()
        prof.add((p.a_rc, p.b_rc, p.tr_mode & 255), e0, e1, 2.0 * M * N * K * p.nbatch, (M, N, K, p.nbatch, p.split_k),
                 _gemm_bytes(p, dtype))
        return
    _launch(p, dtype)


def _gemm_bytes(p, dtype):
    """ALGORITHMIC bytes of one GEMM launch: every distinct operand element read once, every output element written once
    (fp32 slabs: one per K slice), side tensors of the epilogue once.  An overlapping row view (a Conv1d window: rows
    `ld` apart, K > ld elements long) counts its distinct elements, not rows x K."""
    es = 2 if dtype == BF16 else 4

    def operand(v, rows, k, rc):
        if rc:                      # rows-contiguous: [k, rows] with leading dimension ld over k
            return rows * k
        if v.rows_per_batch > 0 and 0 < abs(v.ld) < k:
            nb = (rows + v.rows_per_batch - 1) // v.rows_per_batch
            return nb * ((v.rows_per_batch - 1) * abs(v.ld) + k)
        return rows * k
    a = operand(p.a, p.M, p.K, p.a_rc)
    b = operand(p.b, p.N, p.K, p.b_rc)
    mn = p.M * p.N
    c = mn * (4 * max(p.split_k, 1) if p.out_f32 else es)
    if p.atomic == 2:
        c += mn * 4                 # accumulate: the destination is read as well
    side = mn * es * (bool(p.resid) + bool(p.aux_out) + bool(p.aux_in))
    return float(p.nbatch) * ((a + b) * es + c + side)


def pp_split(Mo, No, Kred):
    """K-slice count for a weight-gradient GEMM on the ping-pong kernel: one round of 256x256 work items over the CUs
    (over the CUs backward may use when RCCL runs beside it: pp_cus)."""
    tiles = ((Mo + 255) // 256) * ((No + 255) // 256)
    ksteps = (Kred + 63) // 64
    want = max(1, min((pp_cus() or 256) // max(tiles, 1), ksteps // 4))
    per = (ksteps + want - 1) // want
    return (ksteps + per - 1) // per


def gemm_splitk(a, b, c, M, N, K, dtype, split, slabs, **kw):
    """Split-K form of `gemm` for outputs with too few tiles to fill the chip: `split` K-slices write fp32 partial slabs
    (slabs: >= split * M * ceil8(N) floats), one streaming pass sums them and applies the whole epilogue."""
    ldn = (N + 7) // 8 * 8
    stride = M * ldn
    plain = {k: kw[k] for k in ("a_rc", "b_rc", "av", "bv", "tr_mode") if k in kw}
    gemm(a, b, slabs, M, N, K, dtype, cv=view(ldn), out_f32=True, split_k=split, split_stride=stride, **plain)
    epi = {k: v for k, v in kw.items() if k not in ("av", "bv", "tr_mode")}
    p = _gemm_params(a, b, c, M, N, K, **epi)
    L.check(L.lib().smx_gemm_splitk_epilogue(C.byref(p), C.c_void_p(_ptr(slabs)), split, C.c_longlong(stride), ldn, _stream()),
            "smx_gemm_splitk_epilogue")


class GemmProfile:
    """Live per-launch timing of the GEMM kernel variants with HIP events on the launch stream (bench.py).
    Variants: (a_rc, b_rc) = (0,0) forward, (0,1) data gradient, (1,1) weight gradient.  Every record carries the launch's
    flops, its ALGORITHMIC bytes (_gemm_bytes), the engine's tag and whether a second compute stream was active beside it
    (the LM stage's weight-gradient stream: durations then include the contention)."""
    _ROLE = {(0, 0): "fwd", (0, 1): "dgrad", (1, 1): "wgrad", (1, 0): "rc_kc"}
    FAMILY = {1: "gemm_bf16_dma_kernel", 9: "gemm_bf16_dma_kernel", 11: "gemm_bf16_dma8_kernel", 8: "gemm_bf16_pp_kernel",
              12: "gemm_bf16_fr_kernel", 13: "gemm_bf16_fr_kernel"}

    @staticmethod
    def name(key):
        a, b, mode = key
        kern = GemmProfile.FAMILY.get(mode, "gemm_bf16_dma_kernel")
        tile = ", 64x128 tiles" if mode == 9 else ", 192x256 tiles" if mode == 13 else ""
        return f"{kern}<{str(bool(a)).lower()},{str(bool(b)).lower()}> ({GemmProfile._ROLE[(a, b)]}{tile})"

    def __init__(self):
        self.pool, self.used, self.recs, self.tags = [], 0, [], []

    def events(self):
        if self.used + 2 > len(self.pool):
            self.pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(256))
        e = self.pool[self.used], self.pool[self.used + 1]
        self.used += 2
        return e

    def add(self, key, e0, e1, flops, shape=None, nbytes=0.0):
        self.recs.append((key, e0, e1, flops, shape, nbytes, GEMM_CONCURRENT))
        self.tags.append(GEMM_TAG)

    def _groups(self, pick=lambda rec, tag: True):
        """{(variant, shape): [ms, ...]} plus per-group flops / bytes per launch, over the records `pick` accepts."""
        out = {}
        for rec, tag in zip(self.recs, self.tags):
            if not pick(rec, tag):
                continue
            key, e0, e1, fl, shape, nb, conc = rec
            g = out.setdefault((key, shape), dict(ms=[], flops=fl, bytes=nb))
            g["ms"].append(e0.elapsed_time(e1))
        return out

    @staticmethod
    def _median(v):
        v = sorted(v)
        n = len(v)
        return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])

    def robust(self, pick=lambda rec, tag: True):
        """Outlier-proof totals: every (variant, shape) group contributes launches x its MEDIAN duration (single launches of
        ~1.8 ms appear once per few hundred steps on this pool).  -> dict(launches, total_ms, flops, bytes, tflops)."""
        d = dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0)
        for g in self._groups(pick).values():
            n = len(g["ms"])
            d["launches"] += n
            d["total_ms"] += n * self._median(g["ms"])
            d["flops"] += n * g["flops"]
            d["bytes"] += n * g["bytes"]
        d["tflops"] = d["flops"] / (d["total_ms"] * 1e-3) / 1e12 if d["total_ms"] > 0 else 0.0
        return d

    def by_tag(self, tag):
        """Launches recorded while ops.GEMM_TAG == tag (the engine tags the speech encoder's transformer layers "enc_layer":
        their Linear forward / data-gradient / weight-gradient GEMMs are what north_star's 0.40 is defined on):
        -> dict(launches, total_ms, flops, tflops), medians per (variant, shape)."""
        return self.robust(lambda rec, t: t == tag)

    def families(self):
        """{kernel family: robust totals + launches that ran beside the second stream} - the fixed rule bench.py nominates its
        roofline kernel by: the family with the largest share of the GEMM flops."""
        out = {}
        for fam in sorted(set(self.FAMILY.values())):
            d = self.robust(lambda rec, t: self.FAMILY.get(rec[0][2]) == fam)
            if d["launches"]:
                d["concurrent_launches"] = sum(1 for rec in self.recs if self.FAMILY.get(rec[0][2]) == fam and rec[6])
                out[fam] = d
        return out

    def by_shape(self):
        """-> {(variant, (M, N, K, nbatch, split_k)): dict(launches, total_ms, flops)} (tools/gpu_gemm_shapes.py)."""
        out = {}
        for key, e0, e1, fl, shape, nb, conc in self.recs:
            d = out.setdefault((key, shape), dict(launches=0, total_ms=0.0, flops=0.0))
            d["launches"] += 1
            d["total_ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
        return out

    def summary(self):
        """-> {variant: dict(launches, total_ms, avg_us, median_us, flops, bytes, tflops)}; totals are launches x median per
        (variant, shape) group.  Call after a device synchronize."""
        out = {}
        for key in sorted({rec[0] for rec in self.recs}):
            d = self.robust(lambda rec, t: rec[0] == key)
            d["avg_us"] = 1e3 * d["total_ms"] / d["launches"]
            out[key] = d
        return out


GEMM_PROFILE = None
GEMM_TAG = None          # set by the engine around stages whose GEMMs a report singles out (bench.py: encoder_gemms)
GEMM_CONCURRENT = False  # set by the engine while a second compute stream runs beside the launches (LM stage of backward)


class OpProfile:
    """Live timing of the non-GEMM kernel families (HIP events on the launch stream) with their ALGORITHMIC bytes / flops,
    for the HBM-side lines of bench.py: name -> launches, ms, bytes, flops."""

    def __init__(self):
        self.pool, self.used, self.recs = [], 0, []

    def span(self, name, nbytes=0.0, flops=0.0):
        if self.used + 2 > len(self.pool):
            self.pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(256))
        e0, e1 = self.pool[self.used], self.pool[self.used + 1]
        self.used += 2
        self.recs.append((name, e0, e1, float(nbytes), float(flops)))
        return e0, e1

    def summary(self):
        out = {}
        for name, e0, e1, nb, fl in self.recs:
            d = out.setdefault(name, dict(launches=0, total_ms=0.0, bytes=0.0, flops=0.0))
            d["launches"] += 1
            d["total_ms"] += e0.elapsed_time(e1)
            d["bytes"] += nb
            d["flops"] += fl
        for d in out.values():
            sec = d["total_ms"] * 1e-3
            d["GBps"] = d["bytes"] / sec / 1e9 if sec > 0 else 0.0
            d["tflops"] = d["flops"] / sec / 1e12 if sec > 0 else 0.0
        return out


OP_PROFILE = None


class _Span:
    """with _Span(name, bytes, flops): ...   (no-op unless OP_PROFILE is set)"""

    def __init__(self, name, nbytes=0.0, flops=0.0):
        self.ev = OP_PROFILE.span(name, nbytes, flops) if OP_PROFILE is not None else None

    def __enter__(self):
        if self.ev:
            self.ev[0].record()

    def __exit__(self, *a):
        if self.ev:
            self.ev[1].record()
        return False


def _es(dtype):
    return 2 if dtype == BF16 else 4


def norm_fwd(x, y, gamma, beta, mean, rstd, M, D, dtype, eps=1e-5, rms=False, act=ACT_NONE, pos=None, pos_period=0,
             pos_offset=0, xsum_out=None, drop=None):
    p = L.NormParams(_ptr(x), _ptr(pos), _ptr(xsum_out), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd),
                     M, D, pos_period, pos_offset, int(rms), act, eps)
    if drop is not None and drop[0] > 0:
        p.drop_p, p.drop_seed = drop
    with _Span("norm_fwd", M * D * _es(dtype) * (3 if xsum_out is not None else 2)):
        L.check(L.lib().smx_norm_fwd(C.byref(p), dtype, _stream()), "smx_norm_fwd")


class FoldQueue:
    """Deferred second stages of the two-stage column reductions of backward (bias gradients, LayerNorm gamma / beta
    gradients): every site leaves its partial rows in a scratch tensor and queues (scratch, rows, columns, destination);
    `flush` reduces the whole queue in one launch per 48 entries (smx_fold_many) instead of one 8-us launch per site.
    The engine flushes at the end of every backward stage, before that stage's gradients are handed to the reducer."""

    def __init__(self):
        self.items = []

    def add(self, ws, ws_off, dst, nrows, ncols, ld, alpha=1.0):
        self.items.append((ws, ws_off, dst, nrows, ncols, ld, alpha))      # holds `ws` alive until the flush is enqueued

    def flush(self):
        items, self.items = self.items, []
        for i in range(0, len(items), L.FOLD_MAX):
            chunk = items[i:i + L.FOLD_MAX]
            t = L.FoldTable()
            t.n = len(chunk)
            for j, (ws, off, dst, nrows, ncols, ld, alpha) in enumerate(chunk):
                e = t.e[j]
                e.ws, e.dst = _ptr(ws) + 4 * off, _ptr(dst)
                e.nrows, e.ncols, e.ld, e.alpha = nrows, ncols, ld, alpha
            L.check(L.lib().smx_fold_many(C.byref(t), _stream()), "smx_fold_many")


def transpose_many(jobs):
    """jobs: [(src [rows, cols], dst [cols, rows])] of 16-bit tensors (rows, cols multiples of 8) - one launch per 64 matrices."""
    for i in range(0, len(jobs), L.TR_MAX):
        chunk = jobs[i:i + L.TR_MAX]
        t = L.TrTable()
        t.n = len(chunk)
        for j, (src, dst) in enumerate(chunk):
            e = t.e[j]
            e.src, e.dst, e.rows, e.cols = _ptr(src), _ptr(dst), src.shape[0], src.shape[1]
        L.check(L.lib().smx_transpose_many(C.byref(t), _stream()), "smx_transpose_many")


def norm_bwd(dy, x, dx, gamma, beta, mean, rstd, dgamma, dbeta, M, D, dtype, rms=False, act=ACT_NONE, dres=None,
             dpos=None, pos_period=0, pos_offset=0, drop=None, folds=None, drop2=None, dx_drop=None, gb2=None):
    """folds: a FoldQueue - the gamma / beta partial rows are then reduced at its next flush instead of by a launch here.
    drop2 = (p, seed), dx_drop, gb2 (needs folds and dgamma / dbeta): also writes dx * mask(drop2) and queues its column
    sums into gb2 (the bias gradient of the Linear whose dropped output fed this norm) - one pass instead of a separate
    smx_dropout_colsum over dx."""
    ws = None
    rows = L.lib().smx_norm_bwd_partial_rows(M)      # one partial-row set per block of the fused kernel (4 / 8 / 16 rows by M)
    third = dx_drop is not None
    nrow = 3 if third else 2
    if third:
        if folds is None or (dgamma is None and dbeta is None) or drop2 is None or gb2 is None:
            raise RuntimeError("norm_bwd: dx_drop needs a FoldQueue, gamma / beta gradients, drop2 and gb2")
    if dgamma is not None or dbeta is not None:
        need = rows * nrow * D
        if folds is not None:
            ws = torch.empty(need, dtype=torch.float32, device=dy.device)
        else:
            ws = _NORM_WS.get(dy.device)
            if ws is None or ws.numel() < need:
                ws = _NORM_WS[dy.device] = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=dy.device)
    p = L.NormBwdParams(_ptr(dy), _ptr(x), _ptr(dres), _ptr(dx), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd),
                        _ptr(dgamma), _ptr(dbeta), _ptr(dpos), _ptr(ws), M, D, pos_period, pos_offset, int(rms), act)
    if drop is not None and drop[0] > 0:
        p.drop_p, p.drop_seed = drop
    if folds is not None and ws is not None:
        p.defer_fold = 1
    if third:
        p.dx_drop, p.drop2_p, p.drop2_seed = _ptr(dx_drop), drop2[0], drop2[1]
    with _Span("norm_bwd", M * D * _es(dtype) * ((4 if dres is not None else 3) + (1 if third else 0))):
        L.check(L.lib().smx_norm_bwd(C.byref(p), dtype, _stream()), "smx_norm_bwd")
    if folds is not None and ws is not None:
        assert rows == L.lib().smx_norm_bwd_partial_rows(M)
        if dgamma is not None:
            folds.add(ws, 0, dgamma, rows, D, nrow * D)
        if dbeta is not None:
            folds.add(ws, D, dbeta, rows, D, nrow * D)
        if third:
            folds.add(ws, 2 * D, gb2, rows, D, nrow * D)


_NORM_WS = {}


class AttnDesc:
    """Strided description of Q/K/V/O living inside fused projection buffers (element units)."""

    def __init__(self, B, H, Tq, Tk, D, causal, scale, bias=None, drop=None, klen=None, masks=None):
        """klen: optional int32 [B] device tensor - keys at positions >= klen[b] are padding (right-padded attention mask).
        masks = (mask_q, mask_k): the dropout bit matrices of `drop`, already generated (attn_dropout_masks)."""
        self.p = L.AttnParams()
        p = self.p
        p.B, p.H, p.Tq, p.Tk, p.D, p.causal, p.scale = B, H, Tq, Tk, D, int(causal), scale
        if drop is not None and drop[0] > 0:
            p.drop_p, p.drop_seed = drop
        p.bias = _ptr(bias)
        if klen is not None:
            assert klen.dtype == torch.int32 and klen.numel() == B and klen.is_cuda
        p.klen = _ptr(klen)
        self._keep = [bias, klen]
        if masks is not None:
            p.mask_q, p.mask_k = _ptr(masks[0]), _ptr(masks[1])
            self._keep += list(masks)

    def set(self, name, tensor, elem_off, batch_stride, ld):
        """name in Q K V O dO dQ dK dV"""
        setattr(self.p, name, _ptr(tensor) + elem_off * tensor.element_size())
        pre = {"Q": "q", "K": "k", "V": "v", "O": "o", "dO": "do", "dQ": "dq", "dK": "dk", "dV": "dv"}[name]
        setattr(self.p, pre + "_bs", batch_stride)
        setattr(self.p, pre + "_ld", ld)
        self._keep.append(tensor)


def attn_dropout_masks(B, H, Tq, Tk, D, p, seed, dtype, device, reuse=None):
    """The keep mask of one attention call's probability dropout as bit matrices (both orientations), generated on the CURRENT
    stream.  -> (mask_q, mask_k) int32 tensors, or None when this (dtype, head_dim) path hashes in-kernel.  reuse: a previous
    pair of the same sizes to overwrite."""
    a = L.AttnParams()
    a.B, a.H, a.Tq, a.Tk, a.D = B, H, Tq, Tk, D
    a.drop_p, a.drop_seed = p, seed
    nq, nk = C.c_longlong(0), C.c_longlong(0)
    L.check(L.lib().smx_attn_mask_words(C.byref(a), dtype, C.byref(nq), C.byref(nk)), "smx_attn_mask_words")
    if not nq.value:
        return None
    if reuse is not None and reuse[0].numel() == nq.value and reuse[1].numel() == nk.value:
        mq, mk = reuse
    else:
        mq = torch.empty(nq.value, dtype=torch.int32, device=device)
        mk = torch.empty(nk.value, dtype=torch.int32, device=device)
    a.mask_q, a.mask_k = _ptr(mq), _ptr(mk)
    L.check(L.lib().smx_attn_dropout_mask(C.byref(a), _stream()), "smx_attn_dropout_mask")
    return mq, mk


def attention_fwd(desc, lse, dtype):
    desc.p.lse = _ptr(lse)
    if desc.p.drop_p > 0 and not desc.p.mask_q:
        # MFMA path: the keep mask of this call as bit matrices (both orientations), generated once and reused by backward
        nq, nk = C.c_longlong(0), C.c_longlong(0)
        L.check(L.lib().smx_attn_mask_words(C.byref(desc.p), dtype, C.byref(nq), C.byref(nk)), "smx_attn_mask_words")
        if nq.value:
            dev = lse.device
            mq = torch.empty(nq.value, dtype=torch.int32, device=dev)
            mk = torch.empty(nk.value, dtype=torch.int32, device=dev)
            desc.p.mask_q, desc.p.mask_k = _ptr(mq), _ptr(mk)
            desc._keep += [mq, mk]
            L.check(L.lib().smx_attn_dropout_mask(C.byref(desc.p), _stream()), "smx_attn_dropout_mask")
    pp = desc.p
    with _Span("attention_fwd", 0, 4.0 * pp.B * pp.H * pp.Tq * pp.Tk * pp.D * (0.5 if pp.causal else 1.0)):
        L.check(L.lib().smx_attention_fwd(C.byref(desc.p), dtype, _stream()), "smx_attention_fwd")


def attention_bwd(desc, lse, delta, dtype, dbias=None):
    desc.p.lse, desc.p.delta, desc.p.dbias = _ptr(lse), _ptr(delta), _ptr(dbias)
    pp = desc.p
    with _Span("attention_bwd", 0, 10.0 * pp.B * pp.H * pp.Tq * pp.Tk * pp.D * (0.5 if pp.causal else 1.0)):
        L.check(L.lib().smx_attention_bwd(C.byref(desc.p), dtype, _stream()), "smx_attention_bwd")


def attn_bias_scatter(dbias, bucket, dtable, H, Tq, Tk, nbuckets):
    """dtable[bucket[q, k], h] += dbias[h, q, k] (T5 relative-position bias table gradient)."""
    L.check(L.lib().smx_attn_bias_scatter(C.c_void_p(_ptr(dbias)), C.c_void_p(_ptr(bucket)), C.c_void_p(_ptr(dtable)), H, Tq, Tk,
                                          nbuckets, _stream()), "smx_attn_bias_scatter")


def conv0_workspace_floats(B, Cc, k):
    fn = L.lib().smx_conv0_workspace_floats
    fn.restype = C.c_longlong
    return int(fn(B, Cc, k))


def conv0_params(wave, w, cbias, gamma, beta, stats, y, B, N, Cc, k, stride, T0, group, eps=1e-5, partials=None):
    p = L.Conv0Params()
    p.partials = _ptr(partials)
    p.wave, p.w, p.cbias, p.gamma, p.beta, p.stats, p.y = _ptr(wave), _ptr(w), _ptr(cbias), _ptr(gamma), _ptr(beta), \
        _ptr(stats), _ptr(y)
    p.B, p.N, p.C, p.k, p.stride, p.T0, p.group, p.eps = B, N, Cc, k, stride, T0, int(group), eps
    return p


def conv0_fwd(p, dtype):
    with _Span("conv0_fwd", p.B * p.T0 * p.C * _es(dtype) + 2 * p.B * p.N * 4):      # activation written once, waveform read twice
        L.check(L.lib().smx_conv0_fwd(C.byref(p), dtype, _stream()), "smx_conv0_fwd")


def conv0_bwd(p, dy, bstats, dw, dcbias, dgamma, dbeta, dtype):
    p.dy, p.bstats, p.dw, p.dcbias, p.dgamma, p.dbeta = _ptr(dy), _ptr(bstats), _ptr(dw), _ptr(dcbias), _ptr(dgamma), \
        _ptr(dbeta)
    with _Span("conv0_bwd", p.B * p.T0 * p.C * _es(dtype) + 2 * p.B * p.N * 4):      # dy read once
        L.check(L.lib().smx_conv0_bwd(C.byref(p), dtype, _stream()), "smx_conv0_bwd")


def cast_from_f32(src, dst, n, dtype):
    L.check(L.lib().smx_cast_from_f32(C.c_void_p(_ptr(src)), C.c_void_p(_ptr(dst)), C.c_longlong(n), dtype, _stream()),
            "smx_cast_from_f32")


def cast_to_f32(src, dst, n, dtype):
    L.check(L.lib().smx_cast_to_f32(C.c_void_p(_ptr(src)), C.c_void_p(_ptr(dst)), C.c_longlong(n), dtype, _stream()),
            "smx_cast_to_f32")


def pack_conv_w(w, out, Co, Ci, k, dtype):
    L.check(L.lib().smx_pack_conv_w(C.c_void_p(_ptr(w)), C.c_void_p(_ptr(out)), Co, Ci, k, dtype, _stream()),
            "smx_pack_conv_w")


def pack_conv_w_dgrad(w, out, Co, Ci, k, stride, dtype):
    L.check(L.lib().smx_pack_conv_w_dgrad(C.c_void_p(_ptr(w)), C.c_void_p(_ptr(out)), Co, Ci, k, stride, dtype, _stream()),
            "smx_pack_conv_w_dgrad")


def unpack_conv_dw(dwp, dw, Co, Ci, k):
    L.check(L.lib().smx_unpack_conv_dw(C.c_void_p(_ptr(dwp)), C.c_void_p(_ptr(dw)), Co, Ci, k, _stream()),
            "smx_unpack_conv_dw")


def embed_fwd(ids, table, out, M, D, scale, dtype):
    L.check(L.lib().smx_embed_fwd(C.c_void_p(_ptr(ids)), C.c_void_p(_ptr(table)), C.c_void_p(_ptr(out)), M, D,
                                  C.c_float(scale), dtype, _stream()), "smx_embed_fwd")


def embed_bwd(ids, dy, dtable, M, D, scale, dtype):
    L.check(L.lib().smx_embed_bwd(C.c_void_p(_ptr(ids)), C.c_void_p(_ptr(dy)), C.c_void_p(_ptr(dtable)), M, D,
                                  C.c_float(scale), dtype, _stream()), "smx_embed_bwd")


def colsum(x, out, M, N, ld, dtype, alpha=1.0, folds=None):
    """out[n] += alpha * sum_m x[m, n]; tall inputs go through the two-stage kernel with a cached scratch buffer, or -
    with a FoldQueue - leave their partial rows for its next flush."""
    fn = L.lib().smx_colsum_ws_floats
    fn.restype = C.c_longlong
    need = int(fn(M, N))
    if (folds is not None and M >= L.lib().smx_colsum_min_rows() and N % 8 == 0 and ld % 8 == 0
            and os.environ.get("SMX_COLSUM") != "atomic"):
        ws = torch.empty(need, dtype=torch.float32, device=x.device)
        L.check(L.lib().smx_colsum_ws(C.c_void_p(_ptr(x)), C.c_void_p(0), M, N, C.c_longlong(ld), C.c_float(alpha),
                                      dtype, C.c_void_p(_ptr(ws)), _stream()), "smx_colsum_ws")
        folds.add(ws, 0, out, L.lib().smx_colsum_slices(M, N), N, (N + 7) // 8 * 8, alpha)
        return
    ws = _COLSUM_WS.get(x.device)
    if os.environ.get("SMX_COLSUM") == "atomic":          # A/B switch: single-stage atomic kernel
        L.check(L.lib().smx_colsum(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), M, N, C.c_longlong(ld), C.c_float(alpha),
                                   dtype, _stream()), "smx_colsum")
        return
    if ws is None or ws.numel() < need:
        ws = _COLSUM_WS[x.device] = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=x.device)
    L.check(L.lib().smx_colsum_ws(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), M, N, C.c_longlong(ld), C.c_float(alpha),
                                  dtype, C.c_void_p(_ptr(ws)), _stream()), "smx_colsum_ws")


_COLSUM_WS = {}


def dropout_colsum(x, out, M, N, p, seed, colsum_out, dtype, alpha=1.0, folds=None):
    """out = dropout(x; p, seed) and colsum_out[n] += alpha * sum_m out[m, n] in one pass (N % 8 == 0); with a FoldQueue
    the column sums land at its next flush."""
    fn = L.lib().smx_colsum_ws_floats
    fn.restype = C.c_longlong
    need = max(int(fn(M, N)), 64 * N)
    if folds is not None:
        slices = L.lib().smx_dropout_colsum_slices(M, N)
        ws = torch.empty(max(need, slices * N), dtype=torch.float32, device=x.device)
        L.check(L.lib().smx_dropout_colsum(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), M, N, C.c_float(p), C.c_uint(seed),
                                           C.c_void_p(0), C.c_float(alpha), C.c_void_p(_ptr(ws)), dtype, _stream()),
                "smx_dropout_colsum")
        folds.add(ws, 0, colsum_out, slices, N, N, alpha)
        return
    ws = _COLSUM_WS.get(x.device)
    if ws is None or ws.numel() < need:
        ws = _COLSUM_WS[x.device] = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=x.device)
    L.check(L.lib().smx_dropout_colsum(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), M, N, C.c_float(p), C.c_uint(seed),
                                       C.c_void_p(_ptr(colsum_out)), C.c_float(alpha), C.c_void_p(_ptr(ws)), dtype, _stream()),
            "smx_dropout_colsum")


def cross_entropy(logits, labels, loss, argmax, dlogits, M, V, ldl, ldd, dtype, gscale=1.0, lse=None, logits_t=None,
                  kld=None, kld_scale=0.0, count_labels=None):
    """count_labels: all labels of the batch when this call covers a row chunk of it (the mean's denominator stays global)."""
    p = L.CEParams(_ptr(logits), _ptr(labels), _ptr(loss), _ptr(argmax), _ptr(dlogits), _ptr(lse), M, V, ldl, ldd, gscale,
                   _ptr(logits_t), _ptr(kld), kld_scale, _ptr(count_labels), count_labels.numel() if count_labels is not None else 0)
    nb = M * V * (8 + (_es(dtype) if dlogits is not None else 0))          # two passes over the fp32 logits + the gradient written
    with _Span("cross_entropy", nb):
        L.check(L.lib().smx_cross_entropy(C.byref(p), dtype, _stream()), "smx_cross_entropy")


def add(a, b, out, n, dtype):
    L.check(L.lib().smx_add(C.c_void_p(_ptr(a)), C.c_void_p(_ptr(b)), C.c_void_p(_ptr(out)), C.c_longlong(n), dtype,
                            _stream()), "smx_add")


def zero_ranges(base, table, n):
    """base[off : off + cnt] = 0 for the n (off, cnt) rows of the int64 device table (cnt <= 65536)."""
    L.check(L.lib().smx_zero_ranges(C.c_void_p(_ptr(base)), C.c_void_p(_ptr(table)), n, _stream()), "smx_zero_ranges")


def mask_rows(x, rows, nrows, emb, D, dtype):
    L.check(L.lib().smx_mask_rows(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(rows)), nrows, C.c_void_p(_ptr(emb)), D, dtype,
                                  _stream()), "smx_mask_rows")


def mask_rows_bwd(dx, rows, nrows, demb, D, dtype):
    L.check(L.lib().smx_mask_rows_bwd(C.c_void_p(_ptr(dx)), C.c_void_p(_ptr(rows)), nrows, C.c_void_p(_ptr(demb)), D,
                                      dtype, _stream()), "smx_mask_rows_bwd")


def group_pack(x, xg, B, T, Cc, G, K, pad_front, dtype):
    L.check(L.lib().smx_group_pack(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(xg)), B, T, Cc, G, K, pad_front, dtype,
                                   _stream()), "smx_group_pack")


def posconv_pack_w(wp, out, G, Cg, K, J, flip, dtype):
    L.check(L.lib().smx_posconv_pack_w(C.c_void_p(_ptr(wp)), C.c_void_p(_ptr(out)), G, Cg, K, J, int(flip), dtype, _stream()),
            "smx_posconv_pack_w")


def posconv_unpack(tmp, bias, resid, pre, y, B, T, Cc, G, Tq, act, dtype):
    with _Span("posconv_unpack", B * T * Cc * (4 + _es(dtype) * (1 + (resid is not None) + (pre is not None)))):
        L.check(L.lib().smx_posconv_unpack(C.c_void_p(_ptr(tmp)), C.c_void_p(_ptr(bias)), C.c_void_p(_ptr(resid)), C.c_void_p(_ptr(pre)),
                                           C.c_void_p(_ptr(y)), B, T, Cc, G, Tq, act, dtype, _stream()), "smx_posconv_unpack")


def posconv_fold_dw(dwJ, dwp, G, Cg, K, J):
    L.check(L.lib().smx_posconv_fold_dw(C.c_void_p(_ptr(dwJ)), C.c_void_p(_ptr(dwp)), G, Cg, K, J, _stream()), "smx_posconv_fold_dw")


def wn_scratch_floats(Cc, Cg, K):
    """Size of the `norm` / `scratch` buffers of wn_fwd / wn_bwd: K results followed by the partial rows of their reductions."""
    return K * (1 + L.lib().smx_wn_partial_blocks(Cc, Cg))


def wn_fwd(v, g, wp, wf, norm, Cc, Cg, K, dtype):
    L.check(L.lib().smx_wn_fwd(C.c_void_p(_ptr(v)), C.c_void_p(_ptr(g)), C.c_void_p(_ptr(wp)), C.c_void_p(_ptr(wf)),
                               C.c_void_p(_ptr(norm)), Cc, Cg, K, dtype, _stream()), "smx_wn_fwd")


def wn_bwd(dwp, v, g, norm, scratch, dg, dv, Cc, Cg, K):
    L.check(L.lib().smx_wn_bwd(C.c_void_p(_ptr(dwp)), C.c_void_p(_ptr(v)), C.c_void_p(_ptr(g)), C.c_void_p(_ptr(norm)),
                               C.c_void_p(_ptr(scratch)), C.c_void_p(_ptr(dg)), C.c_void_p(_ptr(dv)), Cc, Cg, K,
                               _stream()), "smx_wn_bwd")


def sumsq(g, n, out):
    L.check(L.lib().smx_sumsq(C.c_void_p(_ptr(g)), C.c_longlong(n), C.c_void_p(_ptr(out)), _stream()), "smx_sumsq")


def optimizer_step(p, g, m, v, shadow, gnorm_sq, n, lr, kind="adamw", beta1=0.9, beta2=0.999, eps=1e-8,
                   weight_decay=0.0, step=1, grad_scale=1.0, max_grad_norm=0.0):
    o = L.OptParams()
    o.p, o.g, o.m, o.v, o.shadow, o.gnorm_sq = _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(shadow), _ptr(gnorm_sq)
    o.n, o.lr, o.beta1, o.beta2, o.eps, o.weight_decay = n, lr, beta1, beta2, eps, weight_decay
    o.bias_c1, o.bias_c2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    o.grad_scale, o.max_grad_norm, o.kind = grad_scale, max_grad_norm, 1 if kind == "adamw" else 0
    L.check(L.lib().smx_optimizer_step(C.byref(o), _stream()), "smx_optimizer_step")


class AdafactorPlan:
    """Host-built work lists + device state for smx_adafactor_step over a set of tensors living in one flat buffer.
    tensors: [(element offset, shape)] in the flat p / g buffers.  (TF:optimization.py Adafactor: factored second
    moments for >= 2-D tensors over their last two dims, unfactored for 1-D.)"""
    ROWS_MIN, TILE_ELEMS, MAXC, VEC_TILE, MAX_RT = 64, 8192, 2048, 8192, 128

    def __init__(self, tensors, device):
        import numpy as np
        self.n = len(tensors)
        tt = np.zeros(self.n, dtype=np.dtype([("off", "<i8"), ("nb", "<i4"), ("R", "<i4"), ("C", "<i4"), ("row_off", "<i4"),
                                               ("col_off", "<i4"), ("rm_off", "<i4"), ("factored", "<i4"), ("tile0", "<i4"),
                                               ("ntile", "<i4"), ("_pad", "<i4")]))
        assert tt.dtype.itemsize == C.sizeof(L.AfTensor)
        self._tt_dtype = tt.dtype
        # tiles: (tensor, b, r0, nr, c0, nc, full_rows, full_cols, cp_off, rp_off); segs: (tensor, b, cp_off, n_rt, rp_off, n_ct).
        # Every reduction of the step has a fixed order (csrc/adafactor.hip): a row tile owns row `rt` of its segment's [n_rt][C]
        # block of column partials (cp_off), a column tile row `ct` of its [n_ct][R] block of row sums (rp_off), the fold adds them
        # in tile order; a tensor's tiles are contiguous (tile0, ntile).
        tiles, segs = [], []
        row_n = col_n = rm_n = cp_n = 0
        for t, (off, shape) in enumerate(tensors):
            numel = 1
            for d in shape:
                numel *= d
            tile0 = len(tiles)
            if len(shape) >= 2:
                R, Cn = shape[-2], shape[-1]
                nb = numel // (R * Cn)
                if Cn < 64:
                    tr = R                                   # narrow (conv kernels): the kernel's thread-per-row path
                else:                                        # tall matrices: taller tiles, at most MAX_RT partial rows per column to fold
                    tr = min(R, max(self.ROWS_MIN, self.TILE_ELEMS // Cn, (R + self.MAX_RT - 1) // self.MAX_RT))
                n_rt = (R + tr - 1) // tr
                n_ct = (Cn + self.MAXC - 1) // self.MAXC
                for b in range(nb):
                    seg_cp = seg_rp = 0
                    if n_rt > 1:
                        seg_cp, cp_n = cp_n, cp_n + n_rt * Cn
                    if n_ct > 1:
                        seg_rp, cp_n = cp_n, cp_n + n_ct * R
                    segs.append((t, b, seg_cp, n_rt, seg_rp, n_ct))
                    for rt, r0 in enumerate(range(0, R, tr)):
                        for ct, c0 in enumerate(range(0, Cn, self.MAXC)):
                            tiles.append((t, b, r0, min(tr, R - r0), c0, min(self.MAXC, Cn - c0), int(n_ct == 1), int(n_rt == 1),
                                          seg_cp + rt * Cn + c0 if n_rt > 1 else 0, seg_rp + ct * R if n_ct > 1 else 0))
                tt[t] = (off, nb, R, Cn, row_n, col_n, rm_n, 1, tile0, len(tiles) - tile0, 0)
                row_n += nb * R
                col_n += nb * Cn
                rm_n += nb
            else:
                for c0 in range(0, numel, self.VEC_TILE):
                    tiles.append((t, 0, 0, 1, c0, min(self.VEC_TILE, numel - c0), 1, 1, 0, 0))
                tt[t] = (off, 1, 1, numel, 0, col_n, 0, 0, tile0, len(tiles) - tile0, 0)
                col_n += numel
        assert cp_n < 2 ** 31
        self._numel = []
        for off, shape in tensors:
            n = 1
            for dd in shape:
                n *= dd
            self._numel.append((off, n))
        self.ntiles, self.nsegs = len(tiles), len(segs)
        self.row_n, self.col_n = max(row_n, 1), max(col_n, 1)
        dev = device
        self.tensors = torch.from_numpy(tt.view(np.uint8).copy()).to(dev)
        self.tiles = torch.tensor(tiles if tiles else [[0] * 10], dtype=torch.int32, device=dev)
        self.segs = torch.tensor(segs if segs else [[0] * 6], dtype=torch.int32, device=dev)
        self.usq_part = torch.zeros(max(len(tiles), 1), dtype=torch.float32, device=dev)
        self.gsq_part = torch.zeros(max(len(tiles), 1), dtype=torch.float32, device=dev)
        self.gn2 = torch.zeros(1, dtype=torch.float32, device=dev)          # sum (grad_scale g)^2 of the last step (clipping on)
        self.cpart = torch.empty(max(cp_n, 1), dtype=torch.float32, device=dev)
        self.row = torch.zeros(self.row_n, dtype=torch.float32, device=dev)
        self.col = torch.zeros(self.col_n, dtype=torch.float32, device=dev)
        self.racc = torch.empty(self.row_n, dtype=torch.float32, device=dev)
        self.cacc = torch.empty(self.col_n, dtype=torch.float32, device=dev)
        self.rmean = torch.zeros(max(rm_n, 1), dtype=torch.float32, device=dev)
        self.usq = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.beta2t = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.steps = np.zeros(self.n, dtype=np.int64)          # per-tensor step counts (HF keeps state["step"] per tensor)
        self._np = np
        # beta2t travels through a small ring of PINNED host buffers: an asynchronous copy from pageable memory is either a
        # hidden host-synchronous staged copy or a read of a freed buffer; a ring slot is reused only after its copy's event
        self._b2_host = [torch.empty(self.n, dtype=torch.float32).pin_memory() if dev.type == "cuda" else
                         torch.empty(self.n, dtype=torch.float32) for _ in range(4)]
        self._b2_ev = [None] * 4
        self._b2_i = 0

    def prepare(self, p, g, shadow, lr, active=None, decay_rate=-0.8, eps1=1e-30, clip_threshold=1.0, grad_scale=1.0, max_grad_norm=0.0):
        """Everything of a step ahead of its launches: per-tensor step counts, the decay factors' upload, the parameter block.
        -> (parameter block, active element count).  Once per optimizer step."""
        np = self._np
        act = np.ones(self.n, dtype=bool) if active is None else np.asarray(active, dtype=bool)
        self.steps[act] += 1
        b2 = np.where(act, 1.0 - np.power(np.maximum(self.steps, 1).astype(np.float64), decay_rate), -1.0).astype(np.float32)
        i = self._b2_i
        self._b2_i = (i + 1) % len(self._b2_host)
        if self._b2_ev[i] is not None:
            self._b2_ev[i].synchronize()                 # (4 steps old: long complete)
        self._b2_host[i].numpy()[:] = b2
        self.beta2t.copy_(self._b2_host[i], non_blocking=True)
        if self.beta2t.is_cuda:
            self._b2_ev[i] = torch.cuda.Event()
            self._b2_ev[i].record()
        o = L.AfParams()
        o.p, o.g, o.shadow = _ptr(p), _ptr(g), _ptr(shadow)
        o.tensors, o.tiles, o.segs = _ptr(self.tensors), _ptr(self.tiles), _ptr(self.segs)
        o.row, o.col, o.racc, o.cacc, o.rmean, o.usq = _ptr(self.row), _ptr(self.col), _ptr(self.racc), _ptr(self.cacc), \
            _ptr(self.rmean), _ptr(self.usq)
        o.usq_part, o.cpart = _ptr(self.usq_part), _ptr(self.cpart)
        o.beta2t, o.gn2, o.gsq_part = _ptr(self.beta2t), _ptr(self.gn2), _ptr(self.gsq_part)
        o.racc_n, o.cacc_n, o.ntensors, o.ntiles, o.nsegs = self.row_n, self.col_n, self.n, self.ntiles, self.nsegs
        o.lr, o.eps1, o.clip_threshold, o.grad_scale, o.max_grad_norm = lr, eps1, clip_threshold, grad_scale, max_grad_norm
        nact = float(sum(t[1] for t, a in zip(self._numel, act) if a)) if hasattr(self, "_numel") else 0.0
        return o, nact

    def early_stats(self, o, split):
        """The statistics pass over every tensor OUTSIDE split = (first, last) on the current stream (gradients that are final before the
        step's backward has ended); `finish(..., early=True)` does the rest."""
        t_a, t_b = int(self.tile0_of(split[0])), int(self.tile0_of(split[1]))
        lib = L.lib()
        if t_a > 0:
            L.check(lib.smx_adafactor_phase(C.byref(o), 2, 0, t_a, _stream()), "smx_adafactor_phase")
        if t_b < self.ntiles:
            L.check(lib.smx_adafactor_phase(C.byref(o), 2, t_b, self.ntiles - t_b, _stream()), "smx_adafactor_phase")

    def finish(self, o, nact, split, tail_stream, early=False):
        """The phased step: statistics (everything, or with early = True only tensors split = (first, last) - the others' were taken by
        `early_stats` on `tail_stream`, which is joined here), global norm + folds and the update of the split range on the current stream;
        the update of all the others on `tail_stream` behind them.  -> the event after which every parameter is final."""
        t_a, t_b = int(self.tile0_of(split[0])), int(self.tile0_of(split[1]))
        lib = L.lib()
        # (op profile: the compute stream's share only - the statistics pass reads the gradients once; the tail runs on `tail_stream`)
        with _Span("adafactor_stats_and_front", 4.0 * nact):
            if early:
                L.check(lib.smx_adafactor_phase(C.byref(o), 2, t_a, t_b - t_a, _stream()), "smx_adafactor_phase")
                torch.cuda.current_stream().wait_stream(tail_stream)
                L.check(lib.smx_adafactor_phase(C.byref(o), 3, 0, 0, _stream()), "smx_adafactor_phase")
            else:
                L.check(lib.smx_adafactor_phase(C.byref(o), 0, 0, 0, _stream()), "smx_adafactor_phase")
            L.check(lib.smx_adafactor_phase(C.byref(o), 1, t_a, t_b - t_a, _stream()), "smx_adafactor_phase")
        ev0 = torch.cuda.Event()
        ev0.record()
        tail_stream.wait_event(ev0)
        with torch.cuda.stream(tail_stream):
            if t_a > 0:
                L.check(lib.smx_adafactor_phase(C.byref(o), 1, 0, t_a, _stream()), "smx_adafactor_phase")
            if t_b < self.ntiles:
                L.check(lib.smx_adafactor_phase(C.byref(o), 1, t_b, self.ntiles - t_b, _stream()), "smx_adafactor_phase")
            done = torch.cuda.Event()
            done.record()
        return done

    def step(self, p, g, shadow, lr, active=None, decay_rate=-0.8, eps1=1e-30, clip_threshold=1.0, grad_scale=1.0,
             max_grad_norm=0.0, split=None, tail_stream=None):
        """active: optional bool sequence per tensor; tensors without a gradient this step are skipped (state untouched).
        max_grad_norm > 0: global-norm clipping (HF Trainer's clip_grad_norm_ before optimizer.step); the norm comes out of the
        step's own statistics pass over the gradient.  split / tail_stream: the phased form (`finish`)."""
        o, nact = self.prepare(p, g, shadow, lr, active, decay_rate, eps1, clip_threshold, grad_scale, max_grad_norm)
        if split is None:
            with _Span("adafactor_step", 22.0 * nact):       # g read 3x, p read + written, bf16 copy written (csrc/adafactor.hip)
                L.check(L.lib().smx_adafactor_step(C.byref(o), _stream()), "smx_adafactor_step")
            return None
        return self.finish(o, nact, split, tail_stream)

    def tile0_of(self, t):
        """First tile of tensor t (tensors in the order given to the constructor); t == n: the tile count."""
        if not hasattr(self, "_tile0"):
            tt = self.tensors.cpu().numpy().view(self._tt_dtype)
            self._tile0 = [int(x) for x in tt["tile0"]] + [self.ntiles]
        return self._tile0[t]


def act_bwd(dy, pre, dx, M, N, out_view, act, dtype):
    L.check(L.lib().smx_act_bwd(C.c_void_p(_ptr(dy)), C.c_void_p(_ptr(pre)), C.c_void_p(_ptr(dx)), M, N,
                                C.byref(out_view), act, dtype, _stream()), "smx_act_bwd")


def reduce_slabs(slabs, nsplit, n, stride, dst, accumulate=True):
    with _Span("reduce_slabs", 4.0 * n * (nsplit + (2 if accumulate else 1))):
        _reduce_slabs(slabs, nsplit, n, stride, dst, accumulate)


def _reduce_slabs(slabs, nsplit, n, stride, dst, accumulate):
    L.check(L.lib().smx_reduce_slabs(C.c_void_p(_ptr(slabs)), nsplit, C.c_longlong(n), C.c_longlong(stride),
                                     C.c_void_p(_ptr(dst)), int(accumulate), _stream()), "smx_reduce_slabs")


def gemm_group(problems, dtype, mode=8):
    """ONE persistent launch of the 256x256 kernel over up to 4 weight-gradient problems (smx_gemm_group).  problems: list of
    (a, b, c, M, N, K, kw) with the arguments of `gemm` (a_rc = b_rc = True, out_f32, plain views)."""
    arr = (L.GemmParams * len(problems))()
    flops = nbytes = 0.0
    for i, (a, b, c, M, N, K, kw) in enumerate(problems):
        arr[i] = _gemm_params(a, b, c, M, N, K, **kw)
        flops += 2.0 * M * N * K
        nbytes += _gemm_bytes(arr[i], dtype)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = prof.events()
        e0.record()
    arr[0].tr_mode = mode | (pp_cus() << 16)          # mode 12: free-running schedule; bits 16..: persistent grid cap
    if _BYTES_LOG is not None and dtype == BF16:
        _BYTES_LOG.append(nbytes)
    L.check(L.lib().smx_gemm_group(arr, len(problems), dtype, _stream()), "smx_gemm_group")
    if prof is not None:
        e1.record()
        a, b, c, M, N, K, kw = problems[0]
        prof.add((1, 1, mode), e0, e1, flops, ("group", len(problems), K, 1, kw.get("split_k", 1)), nbytes)


def reduce_slabs_many(items, accumulate=True):
    """items: [(slabs, nsplit, n, dst)] with slab stride n - one launch for all of them (smx_reduce_slabs_many)."""
    k = len(items)
    sl = (C.c_void_p * k)(*[_ptr(it[0]) for it in items])
    ds = (C.c_void_p * k)(*[_ptr(it[3]) for it in items])
    ns = (C.c_longlong * k)(*[it[2] for it in items])
    sp = (C.c_int * k)(*[it[1] for it in items])
    with _Span("reduce_slabs", 4.0 * sum(it[2] * (it[1] + (2 if accumulate else 1)) for it in items)):
        L.check(L.lib().smx_reduce_slabs_many(sl, ds, ns, sp, k, int(accumulate), _stream()), "smx_reduce_slabs_many")


def softmax_rows(x, R, Cn):
    L.check(L.lib().smx_softmax_rows(C.c_void_p(_ptr(x)), R, Cn, _stream()), "smx_softmax_rows")


def softmax_rows_bwd(p, dp, dx, R, Cn, scale):
    L.check(L.lib().smx_softmax_rows_bwd(C.c_void_p(_ptr(p)), C.c_void_p(_ptr(dp)), C.c_void_p(_ptr(dx)), R, Cn,
                                         C.c_float(scale), _stream()), "smx_softmax_rows_bwd")


def mse(a, b, loss, da, n, gscale=1.0):
    L.check(L.lib().smx_mse(C.c_void_p(_ptr(a)), C.c_void_p(_ptr(b)), C.c_void_p(_ptr(loss)), C.c_void_p(_ptr(da)),
                            C.c_longlong(n), C.c_float(gscale), _stream()), "smx_mse")


def add_f32_into(src, dst, n, dtype):
    L.check(L.lib().smx_add_f32_into(C.c_void_p(_ptr(src)), C.c_void_p(_ptr(dst)), C.c_longlong(n), dtype, _stream()),
            "smx_add_f32_into")


def _wsum_params(hidden, w, out, dy, dots, dw, sw, n):
    p = L.WsumParams()
    for i, h in enumerate(hidden):
        p.h[i] = _ptr(h)
    p.w, p.out, p.dy, p.dots, p.dw, p.sw, p.n, p.L1 = _ptr(w), _ptr(out), _ptr(dy), _ptr(dots), _ptr(dw), _ptr(sw), n, \
        len(hidden)
    return p


def weighted_sum_fwd(hidden, w, out, sw, n, dtype):
    p = _wsum_params(hidden, w, out, None, None, None, sw, n)
    L.check(L.lib().smx_weighted_sum_fwd(C.byref(p), dtype, _stream()), "smx_weighted_sum_fwd")


def weighted_sum_bwd(hidden, w, dy, dots, dw, sw, n, dtype):
    p = _wsum_params(hidden, w, None, dy, dots, dw, sw, n)
    L.check(L.lib().smx_weighted_sum_bwd(C.byref(p), dtype, _stream()), "smx_weighted_sum_bwd")


def axpy_dev(y, x, a, idx, n, init, dtype):
    L.check(L.lib().smx_axpy_dev(C.c_void_p(_ptr(y)), C.c_void_p(_ptr(x)), C.c_void_p(_ptr(a)), idx, C.c_longlong(n),
                                 int(init), dtype, _stream()), "smx_axpy_dev")


CURRENT_KEY = {}         # device index -> the step key last set on it by this process (None: never set = 0 in the library)


def set_step_key(key, device_index=None):
    """The library's step key (csrc/smx_common.h): every kernel that hashes a dropout mask uses (its seed argument + key).
    Stream-ordered on the CURRENT stream; 0 restores the plain-seed behaviour."""
    if device_index is None:
        device_index = torch._C._cuda_getDevice()
    L.check(L.lib().smx_set_step_key(C.c_uint(int(key) & 0xffffffff), _stream()), "smx_set_step_key")
    CURRENT_KEY[device_index] = int(key) & 0xffffffff


def copy_bytes(src, dst, nbytes=None):
    """dst <- src (same byte count, contiguous, 16-byte aligned) as one kernel on the current stream; src may be a PINNED host
    tensor.  The replayed step's copies (graphs.py): `Tensor.copy_` goes through hipMemcpyAsync, which costs ~150 us of idle GPU."""
    n = src.numel() * src.element_size() if nbytes is None else nbytes
    if not dst.is_cuda or not (src.is_cuda or src.is_pinned()):
        raise RuntimeError("copy_bytes: device destination, device or pinned source")
    L.check(L.lib().smx_copy_bytes(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_longlong(n), _stream()), "smx_copy_bytes")


def scale_dev(x, scale):
    """x *= scale in place, `scale` a 0-dim fp32 DEVICE tensor, the product formed in fp32 (one rounding per element, no host read)."""
    dt = BF16 if x.dtype == torch.bfloat16 else F32
    assert x.is_contiguous() and scale.dtype == torch.float32 and scale.is_cuda and x.dtype in (torch.bfloat16, torch.float32)
    L.check(L.lib().smx_scale_dev(C.c_void_p(x.data_ptr()), C.c_longlong(x.numel()), C.c_void_p(scale.data_ptr()), dt, _stream()), "smx_scale_dev")


def dropout(x, out, n, p, seed, dtype):
    """out = x * mask(seed) / (1 - p), mask index = flat element index (same function the fused epilogues use)."""
    L.check(L.lib().smx_dropout(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), C.c_longlong(n), C.c_float(p), C.c_uint(seed), dtype, _stream()), "smx_dropout")
