"""20-second clips (T = 999 frames, the upper end of the reference length filter) with and without padding masks: fp32 and bf16 paths of the whole model agree, losses and gradient norms finite."""
import contextlib, io, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechmix_amd.model import SpeechMixEED
outs = {}
g = torch.Generator().manual_seed(3)
B = 3
wave = (torch.randn(B, 320000, generator=g) * 0.1).clamp_(-1, 1)
wave[1, 200000:] = 0; wave[2, 50000:] = 0
lens = torch.tensor([320000, 200000, 50000])
labels = torch.randint(4, 50000, (B, 40), generator=g); labels[:, -1] = 2
for dt in ("fp32", "bf16"):
    with contextlib.redirect_stdout(io.StringIO()):
        m = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, compute_dtype=dt, init_seed=0).eval()
    for am in (None, lens):
        o = m(wave.cuda(), labels=labels.cuda(), return_model_detail=True, attention_mask=am)
        o["loss"].backward()
        torch.cuda.synchronize()
        gn = sum(float(p.grad.float().pow(2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
        outs[(dt, am is not None)] = (o["raw_logits"].float().cpu(), float(o["loss"]), gn)
        for p in m.parameters():
            p.grad = None
        print(dt, "mask" if am is not None else "nomask", "T frames", o["encoder_last_hidden_state"].shape, "loss", float(o["loss"]), "grad norm", gn, flush=True)
for k in (False, True):
    a, b = outs[("fp32", k)], outs[("bf16", k)]
    print("mask" if k else "nomask", "bf16 vs fp32 logits max diff", (a[0] - b[0]).abs().max().item(), "range", a[0].abs().max().item(), "loss diff", abs(a[1] - b[1]), "grad-norm rel diff", abs(a[2] - b[2]) / a[2])
