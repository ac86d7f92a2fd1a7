"""Where the eager and the graph-replayed step spend the GPU's time, stage by stage, in STEADY STATE (no synchronisation between
steps: the host keeps whatever lead it has): HIP events at the stage boundaries of both modes (Engine.marks / StepGraphs.trace),
medians over the steps.   python tools/gpu_stage_compare.py [steps]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speechmix_amd import graphs
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
from bench import synth_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12


def group(name):
    if name.startswith(("enc_fwd", "fwd:speech")):
        return "2 speech encoder layers fwd (+ front end in eager mode)"
    if name in ("front",):
        return "1 front end fwd (CNN, projection, positional conv)"
    if name in ("fwd:bridge", "fwd:lm", "bwd:lm", "stage:lm"):
        return "3 bridge + LM fwd + loss + LM bwd"
    if name in ("bwd:bridge", "stage:bridge", "pre_layers"):
        return "4 bridge bwd"
    if name.startswith(("bwd:enc_layer", "stage:enc_layer")):
        return "5 speech encoder layers bwd"
    if name in ("bwd:frontend", "stage:frontend", "tail"):
        return "6 front end bwd"
    return "7 optimizer + step boundary"


def run(mode):
    graphs.MODE, graphs.ENABLED = mode, mode != "0"
    with contextlib.redirect_stdout(io.StringIO()):
        model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", share_layer_ratio=0, down_scale=2, compute_dtype="bf16", init_seed=0)
    model.train()
    np.random.seed(1); torch.manual_seed(1)
    runner = StepRunner(model, lr=5e-4, optimizer="adafactor")
    wave, labels = synth_batch(32, model.decoder_model.config.vocab_size, 0, torch.device("cuda:0"))
    for _ in range(6):
        runner.step(wave, labels)
    torch.cuda.synchronize()
    eng = model.engine
    marks = []
    if mode == "0":
        eng.marks = marks
    else:
        assert runner._graphs is not None
        runner._graphs.trace = marks
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step(wave, labels)
        ev = torch.cuda.Event(enable_timing=True); ev.record()
        marks.append(("step_end", ev))
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    acc = {}
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        if n1 in ("start", "fwd:start"):
            n1 = "step_start"
        acc.setdefault(group(n1), []).append(e0.elapsed_time(e1))
    out = {k: sum(v) / steps for k, v in acc.items()}
    eng.marks = None
    del runner, model
    torch.cuda.empty_cache()
    return out, 1e3 * wall / steps, 1e3 * host / steps


res = {m: run(m) for m in ("0", "1")}
keys = sorted(set(res["0"][0]) | set(res["1"][0]))
print(f"{'stage (ms per step on the GPU)':62s} {'eager':>8s} {'replay':>8s}")
for k in keys:
    print(f"{k:62s} {res['0'][0].get(k, 0.0):8.3f} {res['1'][0].get(k, 0.0):8.3f}")
print(f"{'step (wall / steps)':62s} {res['0'][1]:8.3f} {res['1'][1]:8.3f}")
print(f"{'host loop time / steps':62s} {res['0'][2]:8.3f} {res['1'][2]:8.3f}")
