"""Which parameters break the data-parallel identity grad(batch) = mean(grad(halves))?  Per-tensor relative deviations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gpu_bench_cfg import build
from speechmix_amd.trainer import StepRunner
cfg = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = build(cfg).eval()
V = model.decoder_model.config.vocab_size
g = torch.Generator().manual_seed(1234)
wave = (torch.randn(B, 160000, generator=g) * 0.1).clamp_(-1, 1).cuda()
labels = torch.randint(4, V, (B, 32), generator=g).cuda()
text = torch.randint(4, V, (B, 33), generator=g).cuda() if cfg == "5" else None
sl = lambda t, a, b: t[a:b] if t is not None else None
runner = StepRunner(model, lr=0.0, optimizer="sgd", max_grad_norm=0.0)
runner.step(wave, labels, text_input_ids=text); g_all = model.store.grad.clone()
runner.step(wave, labels, text_input_ids=text); g_all2 = model.store.grad.clone()
h = B // 2
runner.step(wave[:h], labels[:h], text_input_ids=sl(text, 0, h)); g_half = model.store.grad.clone()
runner.step(wave[h:], labels[h:], text_input_ids=sl(text, h, B)); g_half += model.store.grad; g_half *= 0.5
print("rerun rel", ((g_all - g_all2).norm() / g_all.norm()).item())
print("total rel", ((g_all - g_half).norm() / g_all.norm()).item())
rows = []
for n, (o, k, shp) in model.store.offsets.items():
    a, b = g_all[o:o + k], g_half[o:o + k]
    na = a.norm().item()
    if na > 0:
        rows.append(((a - b).norm().item() / na, na, n))
rows.sort(reverse=True)
for r, na, n in rows[:25]:
    print(f"{r:.3e}  |g| {na:.3e}  {n}")
print("...")
tot = g_all.norm().item()
rows.sort(key=lambda t: -t[1])
for r, na, n in rows[:12]:
    print(f"largest |g|: {na / tot:.3f} of total, rel {r:.3e}  {n}")
