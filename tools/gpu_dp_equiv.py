"""Data-parallel equivalence on the real kernels (run as two torch.distributed ranks sharing cuda:0 over gloo; started by
tests/test_gpu_r3.py): the gradient that two ranks x 2 clips (x grad_accum micro-batches) reduce must be the gradient of
the same clips as ONE batch, and one SGD step from it must give the same parameters.  fp32 path, eval mode (no random draws),
real wav2vec2-base -> bart-base widths on 1 s clips.  Prints one JSON line from rank 0."""
import contextlib, io, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

ga = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from speechmix_amd.model import SpeechMixEED, shift_tokens_right
from speechmix_amd.trainer import StepRunner


def build():
    with contextlib.redirect_stdout(io.StringIO()):
        return SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, compute_dtype="fp32", init_seed=0).eval()


per = 2                                   # clips per rank and micro-batch
N = world * ga * per
g = torch.Generator().manual_seed(7)
wave = (torch.randn(N, 16000, generator=g) * 0.1).clamp_(-1, 1).cuda()
model = build()
V = model.decoder_model.config.vocab_size
labels = torch.randint(4, V, (N, 6), generator=g).cuda()
labels[:, -1] = 2
p0 = model.store.master.clone()

runner = StepRunner(model, lr=0.5, optimizer="sgd", max_grad_norm=0.0, grad_accum=ga)
for m in range(ga):                       # micro-batch m of rank r: clips (m * world + r) * per ...
    i0 = (m * world + rank) * per
    runner.step(wave[i0:i0 + per], labels[i0:i0 + per])
torch.cuda.synchronize()
g_dp = model.store.grad.clone() / world   # the all-reduced SUM of the ranks' (accumulated) gradients
p_dp = model.store.master.clone()

# the same N clips as one batch on this rank alone (no reducer): engine forward / backward on a fresh copy of the weights
ref = build()
lc = ref.decoder_model.config
dec = shift_tokens_right(labels, lc.pad_token_id, lc.decoder_start_token_id)
ref._need_engine()
ref.engine.forward(wave, dec.contiguous(), labels.contiguous(), training=False, want_logits=False)
ref.engine.backward()
torch.cuda.synchronize()
g_full = ref.store.grad
gmax = g_full.abs().max().item()
err_g = (g_dp - g_full).abs().max().item() / gmax
err_p = (p_dp - (p0 - 0.5 * g_full)).abs().max().item()
h = p_dp.view(torch.int32).to(torch.int64).sum().reshape(1)
hs = [torch.zeros_like(h) for _ in range(world)]
dist.all_gather(hs, h)
if rank == 0:
    print(json.dumps({"world": world, "grad_accum": ga, "clips": N, "grad_rel_err": err_g, "param_abs_err": err_p, "grad_max": gmax,
                      "in_sync": all(int(x) == int(hs[0]) for x in hs), "moved": (p_dp - p0).abs().max().item()}), flush=True)
dist.destroy_process_group()
