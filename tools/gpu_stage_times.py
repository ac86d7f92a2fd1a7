"""Wall time of the training step's stages on the GPU (HIP events at stage boundaries, no profiler attached) and the host's
enqueue time per step - tells launch-bound regions (stage wall >> its kernels' time in the rocprof table) from busy ones."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
from bench import synth_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", share_layer_ratio=0, down_scale=2, compute_dtype="bf16")
model.train("eval" not in sys.argv)
runner = StepRunner(model, lr=5e-4, optimizer="adafactor")
wave, labels = synth_batch(32, model.decoder_model.config.vocab_size, 0, torch.device("cuda:0"))
for _ in range(3):
    runner.step(wave, labels)
torch.cuda.synchronize()
eng = model.engine
acc, host = {}, []
for _ in range(steps):
    eng.marks = []
    t0 = time.perf_counter()
    runner.step(wave, labels)
    host.append(time.perf_counter() - t0)
    end = torch.cuda.Event(enable_timing=True); end.record()
    torch.cuda.synchronize()
    marks = eng.marks + [("opt:end", end)]
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc.setdefault(n1, []).append(e0.elapsed_time(e1))
eng.marks = None
tot = 0.0
for k, v in acc.items():
    ms = sum(v) / len(v)
    tot += ms
    print(f"{ms:8.3f} ms  -> {k}")
print(f"{tot:8.3f} ms  sum of stages;  host enqueue time per step {1e3 * sum(host) / len(host):.2f} ms")
