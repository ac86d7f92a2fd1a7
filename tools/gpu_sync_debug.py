"""Implicit host<->device synchronisations inside a training step (torch.cuda.set_sync_debug_mode("warn")): every blocking copy or
`.item()` inside StepRunner.step costs the host its lead over the GPU (config 5 lost 4.5 % to one such copy until round 5).
    python tools/gpu_sync_debug.py [CFG=2] [eager|replay]"""
import contextlib, io, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import graphs
from tools.gpu_bench_cfg import build
from speechmix_amd.trainer import StepRunner

cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
mode = sys.argv[2] if len(sys.argv) > 2 else "eager"
graphs.MODE = "1" if mode == "replay" else "0"
graphs.ENABLED = graphs.MODE != "0"
model = build(cfg).train()
V = model.decoder_model.config.vocab_size
g = torch.Generator().manual_seed(0)
B = 32
wave = (torch.randn(B, 160000, generator=g) * 0.1).clamp_(-1, 1).cuda()
labels = torch.randint(4, V, (B, 32), generator=g).cuda()
text = torch.randint(4, V, (B, 33), generator=g).cuda() if cfg == "5" else None
runner = StepRunner(model, lr=1e-5, optimizer="adafactor")
for _ in range(6):
    runner.step(wave, labels, text_input_ids=text)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for _ in range(3):
        runner.step(wave, labels, text_input_ids=text)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
seen = {}
for x in w:
    key = (str(x.message)[:90], x.filename.split("/")[-1], x.lineno)
    seen[key] = seen.get(key, 0) + 1
print(f"config {cfg} {mode}: {len(w)} synchronising calls in 3 steps")
for (msg, fn, ln), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"  {n:3d} x {fn}:{ln}  {msg}")
