"""Upper bound of one lead of DESIGN.md §7 (round 5): the optimizer step of training step i beside the FRONT END of step i + 1.

TIMING ONLY - the results of these steps are wrong: the whole Adafactor step (not only its non-front-end tiles) is launched on a second stream
and the next forward waits for it right before its first encoder layer, so the front end of step i + 1 reads parameters that are being
updated.  What it measures is how much of the ~1.7 ms "optimizer + step boundary" stage disappears behind the 2.6 ms of CNN + feature
projection + positional conv when nothing orders them.    python tools/gpu_opt_overlap_probe.py [steps]"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SMX_STEP_GRAPHS", "0")
import torch
from speechmix_amd import ops
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
from bench import synth_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", share_layer_ratio=0, down_scale=2, compute_dtype="bf16")
model.train()
runner = StepRunner(model, lr=1e-4, optimizer="adafactor", max_grad_norm=1.0, seed=1)
wave, labels = synth_batch(32, model.decoder_model.config.vocab_size, 0, dev)
eng = model.engine
side = torch.cuda.Stream()
state = dict(on=False, ev=None)

af_step = runner.af.step


def af_step_side(*a, **k):
    if not state["on"]:
        return af_step(*a, **k)
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        af_step(*a, **k)
        state["ev"] = torch.cuda.Event()
        state["ev"].record()


runner.af.step = af_step_side
layer_fwd = eng.layer_fwd


def layer_fwd_wait(*a, **k):
    if state["ev"] is not None:
        torch.cuda.current_stream().wait_event(state["ev"])
        state["ev"] = None
    return layer_fwd(*a, **k)


eng.layer_fwd = layer_fwd_wait


def timed(n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        runner.step(wave, labels)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(6):
    runner.step(wave, labels)
res = {False: [], True: []}
for rep in range(3):
    for on in (False, True):
        state["on"] = on
        runner.step(wave, labels); runner.step(wave, labels)
        res[on].append(timed(steps))
print("ms per step, optimizer in stream order: ", [f"{x:.2f}" for x in res[False]])
print("ms per step, optimizer beside the next front end (results invalid): ", [f"{x:.2f}" for x in res[True]])
