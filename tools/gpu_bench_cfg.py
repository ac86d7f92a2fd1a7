"""Run a BASELINE.json config through the native step runner for a few steps (memory / shape / speed check)."""
import sys, os, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED, SpeechMixSelf
from speechmix_amd.trainer import StepRunner

cfg = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
with contextlib.redirect_stdout(io.StringIO()):
    if cfg == "4":
        model = SpeechMixEED("hubert_large_ll60k", "facebook/mbart-large-50", down_scale=8)
    elif cfg == "5":
        model = SpeechMixSelf("wav2vec2_large_960", "t5-large", share_layer_ratio=0.5, down_scale=8)
    else:
        model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2)
model.eval()
V = model.decoder_model.config.vocab_size
g = torch.Generator().manual_seed(0)
wave = (torch.randn(B, 160000, generator=g) * 0.1).clamp_(-1, 1).cuda()
labels = torch.randint(4, V, (B, 32), generator=g).cuda()
text = torch.randint(4, V, (B, 33), generator=g).cuda() if cfg == "5" else None
runner = StepRunner(model, lr=1e-5, optimizer=os.environ.get("SMX_OPT", "adafactor"))
print(f"cfg {cfg}: params {model.store.total/1e6:.1f} M, trainable ranges {len(runner.ranges)}", flush=True)
for i in range(2):
    loss = runner.step(wave, labels, text_input_ids=text)
torch.cuda.synchronize()
print("warm loss", loss.item(), "mem GB", torch.cuda.max_memory_allocated() / 1e9, flush=True)
t0 = time.perf_counter()
for i in range(steps):
    loss = runner.step(wave, labels, text_input_ids=text)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"cfg {cfg}: {dt*1e3:.1f} ms/step  {B*10/dt:.0f} audio-s/s  loss {loss.item():.4f}")
