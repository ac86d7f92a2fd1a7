"""Adafactor step timing on synthetic tensor sets (whole step, events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from tools.gpu_check_pp import bench
dev = torch.device("cuda:0")
sets = {"emb 50265x768": [(50265, 768)], "64 x 3072x768": [(3072, 768)] * 64, "64 x 768x3072": [(768, 3072)] * 64,
        "6 x 512x512x3": [(512, 512, 3)] * 6, "200 x vec 3072": [(3072,)] * 200, "768x48x128": [(768, 48, 128)]}
for name, shapes in sets.items():
    offs, total = [], 0
    for s in shapes:
        offs.append(total)
        total += (torch.Size(s).numel() + 63) // 64 * 64
    p = torch.randn(total, device=dev); g = torch.randn(total, device=dev) * 0.1
    sh = torch.zeros(total, dtype=torch.bfloat16, device=dev)
    plan = ops.AdafactorPlan(list(zip(offs, shapes)), dev)
    t = bench(lambda: plan.step(p, g, sh, None, 5e-4), n=10)
    print(f"{name:18s}: {total/1e6:7.1f} M params, {plan.ntiles:6d} tiles, {t:8.1f} us, {22.0*total/t/1e6:6.2f} TB/s (22 B/param)", flush=True)
