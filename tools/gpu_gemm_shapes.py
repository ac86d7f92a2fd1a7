"""Per-shape GEMM timing inside the real training step (HIP events around every smx_gemm launch)."""
import contextlib, io, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
from bench import synth_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", share_layer_ratio=0, down_scale=2, compute_dtype="bf16")
model.eval()
runner = StepRunner(model)
ops.TUNE_LOG = []
wave, labels = synth_batch(32, model.decoder_model.config.vocab_size, 0, torch.device("cuda:0"))
for _ in range(2):
    runner.step(wave, labels)
prof = ops.GemmProfile()
ops.GEMM_PROFILE = prof
for _ in range(steps):
    runner.step(wave, labels)
torch.cuda.synchronize()
ops.GEMM_PROFILE = None
rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1]["total_ms"])
tot = sum(d["total_ms"] for _, d in rows) / steps
print(f"GEMM total {tot:.2f} ms/step")
names = {(0, 0): "fwd  ", (0, 1): "dgrad", (1, 1): "wgrad", (1, 0): "rc_kc"}
names = {(a, b, m): n + ("/pp " if m == 8 else "/64 " if m == 9 else "/w8 " if m == 11 else "/fr " if m == 12 else "/f19" if m == 13 else "/128") for (a, b), n in names.items() for m in (0, 1, 4, 8, 9, 11, 12, 13)}
for (var, shape), d in rows[:60]:
    M, N, K, nb, sk = shape
    print(f"{d['total_ms']/steps:7.3f} ms/step {d['launches']/steps:5.1f}x avg {1e3*d['total_ms']/d['launches']:8.1f} us "
          f"{d['flops']/d['total_ms']/1e9:7.1f} TF/s  {names[var]} M={M!s:>7} N={N:6d} K={K:7d} nb={nb} split={sk}")

print("\n# kernel choices (us 128x128, us ping-pong, us 64x128, us 256x128, us free-running 256x256, us free-running 192x256, pick)")
us = lambda t: f"{t*1e3:8.1f}" if t is not None else "       -"
for key, t1, t8, mode, *rest in sorted(ops.TUNE_LOG, key=lambda r: -max(r[1] or 0, r[2] or 0)):
    print(f"{us(t1)} {us(t8)} {us(rest[0] if rest else None)} {us(rest[1] if len(rest) > 1 else None)} {us(rest[2] if len(rest) > 2 else None)} {us(rest[3] if len(rest) > 3 else None)} -> {mode}   {key[:9]}")
