"""Where does the bf16 path start to depend on the batch size?  Hidden states of clip 0 alone vs in a batch, stage by stage."""
import contextlib, io, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import ops
from speechmix_amd.model import SpeechMixEED
B, N = 8, 48000
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, compute_dtype="bf16", init_seed=0).eval()
eng = model.engine
g = torch.Generator().manual_seed(1234)
wave = (torch.randn(B, N, generator=g) * 0.1).clamp_(-1, 1).cuda()
model.store.refresh_shadow()
ops.TUNE_LOG = []
def run(w):
    Bc = w.shape[0]
    feat, csv = eng.cnn_fwd(w, Bc, N)
    x, ssv = eng.speech_fwd(w, Bc, N, False)
    T = ssv["T"]
    outs = {"conv0": csv["y"][0].view(Bc, -1, 512)[0], "cnn": feat.view(Bc, T, -1)[0]}
    for i, y in enumerate(csv["y"]):
        outs[f"conv{i}"] = y.view(Bc, -1, 512)[0]
    for i, h in enumerate(ssv["hidden"]):
        outs[f"hidden{i}"] = h.view(Bc, T, -1)[0]
    return {k: v.float().clone() for k, v in outs.items()}
a = run(wave)
b = run(wave[:1].contiguous())
for k in a:
    d = (a[k] - b[k]).abs()
    print(f"{k}: max diff {d.max().item():.3e} (scale {a[k].abs().max().item():.3e}), differing elements {int((d > 0).sum())} of {d.numel()}")
print("tuning decisions:", [(k[:7], m) for k, t1, t8, m, *_ in ops.TUNE_LOG][:20])
