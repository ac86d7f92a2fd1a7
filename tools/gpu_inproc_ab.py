"""In-process A/B of run-time switches on config 2 (train mode, eager steps): ONE model and runner, the settings alternate in blocks of steps, so
clock / box state is shared (process-level A/Bs of bench.py differ by up to 1 ms between identical configurations on some boxes).
    python tools/gpu_inproc_ab.py NAME [blocks] [steps_per_block]
NAME: attn_v3 | dgrad_wt | wgrad_side | lm_stream | colsum_side | pregen | head_side (see MODES)"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SMX_STEP_GRAPHS", "0")
import numpy as np
import torch
import bench as B
from speechmix_amd import engine as E
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner

name = sys.argv[1]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20


def env(k, v):
    def f():
        os.environ[k] = v
    return f


def attr(mod, k, v):
    def f():
        setattr(mod, k, v)
    return f


MODES = {"attn_v3": [("v2 (SMX_ATTN_V3=0)", env("SMX_ATTN_V3", "0")), ("v3 backward (default)", env("SMX_ATTN_V3", "bwd")), ("v3 forward + backward", env("SMX_ATTN_V3", "1"))],
         "dgrad_wt": [("rows-contiguous weight reads (default)", attr(E, "WT_MODE", False)), ("K-contiguous weight copies", attr(E, "WT_MODE", True))],
         "wgrad_side": [("grouped weight gradients on the compute stream (default)", env("SMX_WGRAD_SIDE", "0")), ("on a second stream, joined a layer later", env("SMX_WGRAD_SIDE", "1"))],
         "lm_stream": [("LM weight gradients on a second stream (default)", env("SMX_LM_WGRAD_STREAM", "1")), ("on the compute stream", env("SMX_LM_WGRAD_STREAM", "0"))],
         "colsum_side": [("column sums beside the grouped launch (default)", attr(E, "_COLSUM_SIDE", True)), ("in the data-gradient chain", attr(E, "_COLSUM_SIDE", False))],
         "pregen": [("attention masks generated beside the optimizer (default)", env("SMX_PREGEN_MASKS", "1")), ("generated in place", env("SMX_PREGEN_MASKS", "0"))],
         "head_side": [("LM head weight gradient on the second stream (default)", env("SMX_HEAD_WGRAD_SIDE", "1")), ("on the compute stream", env("SMX_HEAD_WGRAD_SIDE", "0"))]}[name]
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", share_layer_ratio=0, down_scale=2, compute_dtype="bf16", init_seed=0).train()
np.random.seed(1); torch.manual_seed(1)
runner = StepRunner(model, lr=5e-4, optimizer="adafactor", max_grad_norm=1.0)
wave, labels = B.synth_batch(32, model.decoder_model.config.vocab_size, 0, dev)
for _, setter in MODES:          # every setting once ahead of the timing (kernel picks of new keys are tuned live here)
    setter()
    for _ in range(4):
        runner.step(wave, labels)
torch.cuda.synchronize()
res = {n: [] for n, _ in MODES}
for b in range(blocks):
    for n, setter in MODES:
        setter()
        for _ in range(2):
            runner.step(wave, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step(wave, labels)
        torch.cuda.synchronize()
        res[n].append(1e3 * (time.perf_counter() - t0) / steps)
for n, v in res.items():
    print(f"{n:40s} median {sorted(v)[len(v) // 2]:.3f} ms  mean {sum(v) / len(v):.3f}  blocks {[round(x, 2) for x in v]}", flush=True)
