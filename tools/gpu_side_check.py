"""Gradient of one config-2 step with the encoder weight gradients on the second stream vs on the main stream (same seeds)."""
import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
from bench import synth_batch
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, compute_dtype="bf16", init_seed=0)
model.train()
runner = StepRunner(model, lr=0.0, optimizer="sgd", max_grad_norm=0.0)
wave, labels = synth_batch(32, model.decoder_model.config.vocab_size, 0, torch.device("cuda:0"))
res = {}
for rep in range(2):
    for side in ("1", "0"):
        os.environ["SMX_WGRAD_SIDE"] = side
        np.random.seed(3); torch.manual_seed(3)
        model.engine.drop_rng = np.random.default_rng(11)
        loss = runner.step(wave, labels).item()
        torch.cuda.synchronize()
        g = model.store.grad.clone()
        res.setdefault(side, []).append((loss, g))
        print(f"rep {rep} side={side}: loss {loss:.6f} |g| {g.norm().item():.6f} dropped {model.engine.last_dropped}", flush=True)
a, b = res["1"][1][1], res["0"][1][1]
d = (a - b).abs()
print("side1 vs side0: max diff", d.max().item(), "rel", (d.max() / b.abs().max()).item())
print("side0 rep0 vs rep1:", (res["0"][0][1] - res["0"][1][1]).abs().max().item(), " side1 rep0 vs rep1:", (res["1"][0][1] - res["1"][1][1]).abs().max().item())
st = model.store
worst = []
for name, (o, n, _) in st.offsets.items():
    dd = d[o:o + n].max().item()
    if dd > 0:
        worst.append((dd / max(b[o:o + n].abs().max().item(), 1e-12), name))
worst.sort(reverse=True)
for w in worst[:12]:
    print(f"  {w[0]:.3e} {w[1]}")
