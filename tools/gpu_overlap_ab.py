"""In-process A/B of the optimizer overlap modes (config 2, train mode, eager steps): ONE model and runner, the modes alternate in blocks of steps so
that clock / box state is shared - process-level A/Bs of bench.py differ by up to 1 ms between IDENTICAL configurations on some boxes.
    python tools/gpu_overlap_ab.py [blocks] [steps_per_block]
modes: off (one stream) | tail (update of everything but the front end beside the next front end) | tail+early (and the statistics pass of the
early-final gradients beside the front-end backward)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SMX_STEP_GRAPHS", "0")
import numpy as np
import torch
import bench as B
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", share_layer_ratio=0, down_scale=2, compute_dtype="bf16", init_seed=0).train()
np.random.seed(1); torch.manual_seed(1)
runner = StepRunner(model, lr=5e-4, optimizer="adafactor", max_grad_norm=1.0)
wave, labels = B.synth_batch(32, model.decoder_model.config.vocab_size, 0, dev)
split = runner._af_split
assert split is not None
for _ in range(8):
    runner.step(wave, labels)
torch.cuda.synchronize()
res = {"off": [], "tail": [], "tail+early": []}
for b in range(blocks):
    for mode in ("off", "tail", "tail+early"):
        runner._af_split = None if mode == "off" else split
        os.environ["SMX_OPT_EARLY_STATS"] = "1" if mode == "tail+early" else "0"
        for _ in range(2):
            runner.step(wave, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step(wave, labels)
        torch.cuda.synchronize()
        res[mode].append(1e3 * (time.perf_counter() - t0) / steps)
for mode, v in res.items():
    print(f"{mode:11s} median {sorted(v)[len(v) // 2]:.3f} ms  mean {sum(v) / len(v):.3f}  blocks {[round(x, 2) for x in v]}", flush=True)
