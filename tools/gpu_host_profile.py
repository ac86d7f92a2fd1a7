"""Where the HOST's time per training step goes (cProfile over a few steps of config 2; the step is GPU-bound only while the host
enqueues faster than the GPU executes: ~20 ms of Python per ~32-ms step).   python tools/gpu_host_profile.py [steps]"""
import contextlib, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
from bench import synth_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
from speechmix_amd import graphs as _g
# the EAGER host is what this tool profiles by default; SMX_STEP_GRAPHS=1: the step replayed from captured graphs
_g.MODE = os.environ.get("SMX_STEP_GRAPHS", "0")
_g.ENABLED = _g.MODE != "0"
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", share_layer_ratio=0, down_scale=2, compute_dtype="bf16")
model.train()
runner = StepRunner(model, lr=5e-4, optimizer="adafactor")
wave, labels = synth_batch(32, model.decoder_model.config.vocab_size, 0, torch.device("cuda:0"))
for _ in range(8):
    runner.step(wave, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    runner.step(wave, labels)
host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
print(f"host enqueue time per step (unprofiled): {host * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    runner.step(wave, labels)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
