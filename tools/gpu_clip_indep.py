"""Clip independence: logits of clip 0 alone vs inside a batch, fp32 and bf16 compute paths, intermediate stages too."""
import contextlib, io, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED
cfgs = {"2": ("facebook/wav2vec2-base", "facebook/bart-base", 2), "4": ("hubert_large_ll60k", "facebook/mbart-large-50", 8)}
enc, lm, ds = cfgs[sys.argv[1]]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 160000
for dtype in ("fp32", "bf16"):
    with contextlib.redirect_stdout(io.StringIO()):
        model = SpeechMixEED(enc, lm, down_scale=ds, compute_dtype=dtype, init_seed=0).eval()
    V = model.decoder_model.config.vocab_size
    g = torch.Generator().manual_seed(1234)
    wave = (torch.randn(B, N, generator=g) * 0.1).clamp_(-1, 1).cuda()
    labels = torch.randint(4, V, (B, 32), generator=g).cuda()
    with torch.no_grad():
        a = model(wave, labels=labels, return_model_detail=True)
        b = model(wave[:1], labels=labels[:1], return_model_detail=True)
    for k in ("encoder_last_hidden_state", "inputs_embeds", "lm_encoder_last_hidden", "raw_logits"):
        x, y = a[k][:1].float(), b[k].float()
        print(f"[cfg {sys.argv[1]} {dtype}] {k}: max diff {(x - y).abs().max().item():.3e} of scale {x.abs().max().item():.3e}", flush=True)
    del model
    torch.cuda.empty_cache()
