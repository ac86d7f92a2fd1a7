"""Greedy decoding speed at config 2 (32 clips, bart-base decoder, 32 tokens, eos disabled): eager loop vs graph replay."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, compute_dtype="bf16", init_seed=0).eval()
model.decoder_model.config.eos_token_id = -1
B, L = 32, 32
ids = torch.randint(4, 50000, (B, 249), generator=torch.Generator().manual_seed(0))
for mode in ("0", "1"):
    os.environ["SMX_DECODE_GRAPH"] = mode
    for _ in range(3):
        model.generate_from_text(ids, max_length=L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = model.generate_from_text(ids, max_length=L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"SMX_DECODE_GRAPH={mode}: {dt * 1e3:.1f} ms per call (encoder + {L} steps), {B * L / dt:.0f} tokens/s, {dt / L * 1e3:.2f} ms/step", flush=True)
