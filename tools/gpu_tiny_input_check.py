"""Smallest inputs the reference's filter lets through and below: 1 clip, 0.5 - 1 s, 1 - 3 label tokens, batch of 1; fp32 and bf16
paths agree, losses and gradients finite; training step runs."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
g = torch.Generator().manual_seed(1)
models = {}
for dt in ("fp32", "bf16"):
    with contextlib.redirect_stdout(io.StringIO()):
        models[dt] = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, compute_dtype=dt, init_seed=0).eval()
for (B, N, L) in ((1, 16000, 3), (1, 8000, 1), (2, 4000, 2), (5, 1200, 2), (1, 160000, 1)):
    wave = (torch.randn(B, N, generator=g) * 0.1).clamp_(-1, 1)
    labels = torch.randint(4, 50000, (B, L), generator=g)
    res = {}
    for dt, m in models.items():
        try:
            o = m(wave.cuda(), labels=labels.cuda(), return_model_detail=True)
            o["loss"].backward()
            torch.cuda.synchronize()
            gn = sum(float(p.grad.float().pow(2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
            for p in m.parameters():
                p.grad = None
            res[dt] = (o["raw_logits"].detach().float().cpu(), float(o["loss"].detach()), gn, tuple(o["encoder_last_hidden_state"].shape))
        except Exception as e:
            res[dt] = repr(e)[:160]
    if all(not isinstance(v, str) for v in res.values()):
        a, b = res["fp32"], res["bf16"]
        print(f"B={B} N={N} L={L}: frames {a[3]} loss {a[1]:.4f}/{b[1]:.4f} grad norm {a[2]:.3f}/{b[2]:.3f} logits diff {(a[0] - b[0]).abs().max().item():.3e}", flush=True)
    else:
        print(f"B={B} N={N} L={L}: {res}", flush=True)
m = models["bf16"].train()
r = StepRunner(m, lr=5e-4, optimizer="adafactor", max_grad_norm=1.0)
wave = (torch.randn(1, 8000, generator=g) * 0.1).clamp_(-1, 1).cuda(); labels = torch.randint(4, 50000, (1, 2), generator=g).cuda()
losses = [float(r.step(wave, labels)) for _ in range(5)]
print("train mode, 1 clip x 0.5 s, 5 Adafactor steps:", [round(x, 3) for x in losses])
