"""Data-parallel equivalence on the real kernels (run as two torch.distributed ranks sharing cuda:0 over gloo; started by
tests/test_gpu_r3.py): the gradient that two ranks x 2 clips (x grad_accum micro-batches) reduce must be the gradient of
the same clips as ONE batch, and one SGD step from it must give the same parameters.  fp32 path, eval mode (no random draws),
real wav2vec2-base -> bart-base widths on 1 s clips.  Prints one JSON line from rank 0."""
import contextlib, io, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

ga = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kind = sys.argv[2] if len(sys.argv) > 2 else "eed"       # "self": SpeechMixSelf, wav2vec2-base (half the layers) -> frozen t5-small, text pass
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from speechmix_amd.model import SpeechMixEED, SpeechMixSelf
from speechmix_amd.trainer import StepRunner


def build():
    with contextlib.redirect_stdout(io.StringIO()):
        if kind == "self":
            return SpeechMixSelf("facebook/wav2vec2-base", "t5-small", share_layer_ratio=0.5, down_scale=8, compute_dtype="fp32",
                                 init_seed=0).eval()
        return SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, compute_dtype="fp32", init_seed=0).eval()


per = 2                                   # clips per rank and micro-batch
N = world * ga * per
g = torch.Generator().manual_seed(7)
wave = (torch.randn(N, 16000, generator=g) * 0.1).clamp_(-1, 1).cuda()
model = build()
V = model.decoder_model.config.vocab_size
labels = torch.randint(4, V, (N, 6), generator=g).cuda()
labels[:, -1] = 1 if kind == "self" else 2
text = torch.randint(4, V, (N, 7), generator=g).cuda() if kind == "self" else None
p0 = model.store.master.clone()
kw = lambda a, b: {"text_input_ids": text[a:b]} if text is not None else {}

runner = StepRunner(model, lr=0.5, optimizer="sgd", max_grad_norm=0.0, grad_accum=ga)
for m in range(ga):                       # micro-batch m of rank r: clips (m * world + r) * per ...
    i0 = (m * world + rank) * per
    runner.step(wave[i0:i0 + per], labels[i0:i0 + per], **kw(i0, i0 + per))
torch.cuda.synchronize()
g_dp = model.store.grad.clone() / world   # the all-reduced SUM of the ranks' (accumulated) gradients
p_dp = model.store.master.clone()

# the same N clips as ONE batch: every rank runs the whole batch through a second runner, so the reduced sum / world is that
# batch's gradient exactly (x + x = 2 x in binary floating point)
ref = build()
rr = StepRunner(ref, lr=0.5, optimizer="sgd", max_grad_norm=0.0)
rr.step(wave, labels, **kw(0, N))
torch.cuda.synchronize()
g_full = ref.store.grad / world
gmax = g_full.abs().max().item()
err_g = (g_dp - g_full).abs().max().item() / gmax
err_p = (p_dp - ref.store.master).abs().max().item()
frozen = sum(1 for nm in model.store.offsets if not model.store.requires_grad(nm))
h = p_dp.view(torch.int32).to(torch.int64).sum().reshape(1)
hs = [torch.zeros_like(h) for _ in range(world)]
dist.all_gather(hs, h)
if rank == 0:
    print(json.dumps({"world": world, "grad_accum": ga, "clips": N, "grad_rel_err": err_g, "param_abs_err": err_p, "grad_max": gmax,
                      "in_sync": all(int(x) == int(hs[0]) for x in hs), "moved": (p_dp - p0).abs().max().item(), "kind": kind,
                      "frozen_tensors": frozen}), flush=True)
dist.destroy_process_group()
