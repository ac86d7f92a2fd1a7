"""Greedy decoding throughput at config 2 (wav2vec2-base + bart-base, 32 x 10 s clips, bf16): KV-cached path vs the
reference-style loop that re-runs the whole model per token (ref:eval.ipynb cell 6)."""
import contextlib, io, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED
from bench import synth_batch
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", share_layer_ratio=0, down_scale=2, compute_dtype="bf16")
model.eval()
lc = model.decoder_model.config
lc.eos_token_id = -1                       # random-init weights: never stop early, fixed token count
wave, labels = synth_batch(32, lc.vocab_size, 0, torch.device("cuda:0"))
n = 32
for _ in range(2):
    model.generate(wave, max_length=n)
torch.cuda.synchronize()
t0 = time.perf_counter(); out = model.generate(wave, max_length=n); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
t0 = time.perf_counter(); model.generate(wave, max_length=1); torch.cuda.synchronize(); t_one = time.perf_counter() - t0
per_tok = (t_all - t_one) / (n - 1)
print(f"cached greedy: {n} tokens x 32 clips in {t_all*1e3:.1f} ms (encoders + first token {t_one*1e3:.1f} ms, then {per_tok*1e3:.2f} ms per step "
      f"= {32 / per_tok:.0f} tokens/s); {32 * 10 / t_all:.0f} audio-s/s end to end")
# reference-style: full model forward on the growing prefix, arg-max of the last position
pre = torch.full((32, 1), lc.decoder_start_token_id, dtype=torch.int64, device="cuda:0")
with torch.no_grad():
    model(wave, decoder_input_ids=pre)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8):
        o = model(wave, decoder_input_ids=pre, return_model_detail=True)
        nxt = o["raw_logits"][:, -1].argmax(-1)
        pre = torch.cat([pre, nxt[:, None]], 1)
    torch.cuda.synchronize()
    t_re = (time.perf_counter() - t0) / 8
print(f"recompute loop (same kernels, whole model per token): {t_re*1e3:.1f} ms per step = {32 / t_re:.0f} tokens/s  -> cached path {t_re / per_tok:.1f}x")
