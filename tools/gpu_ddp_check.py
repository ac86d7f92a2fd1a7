"""The reference's REAL multi-GPU wrapper around the HIP model (run as two torch.distributed ranks sharing cuda:0
over gloo, started by tests/test_gpu_r5b.py).  Under torchrun, HF Trainer wraps the module in
`DistributedDataParallel(model, find_unused_parameters=True)` (TF:trainer.py:720-737).  Here:
`SpeechMixEED(autograd_param_inputs=True)` (every parameter is an input of the step's single autograd node, so DDP's hooks see
every gradient) inside exactly that wrapper, LayerDrop 0.4 with DIFFERENT keep patterns on the two ranks (the missing-gradient case
of SURVEY section 7 hard part 6), and FreezingCallback-style `requires_grad` flips between steps.  Checked per step: no hang, and
DDP's averaged `.grad` of every parameter == the mean of the two ranks' LOCAL gradients (each rank's un-wrapped twin model with the
same weights, input and keep pattern; local flat gradients all-gathered).  fp32 path; prints one JSON line from rank 0."""
import contextlib, io, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from speechmix_amd.engine import RecordedHostRNG
from speechmix_amd.model import SpeechMixEED

ENC = dict(model_type="wav2vec2", hidden_size=64, num_hidden_layers=4, num_attention_heads=2, intermediate_size=128,
           conv_dim=[32] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2], num_conv_pos_embeddings=16,
           num_conv_pos_embedding_groups=4, layerdrop=0.4, mask_time_prob=0.0, hidden_dropout=0.0, attention_dropout=0.0,
           activation_dropout=0.0, feat_proj_dropout=0.0)
LM = dict(model_type="bart", vocab_size=120, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
          decoder_attention_heads=2, encoder_ffn_dim=128, decoder_ffn_dim=128, max_position_embeddings=128, dropout=0.0,
          attention_dropout=0.0, activation_dropout=0.0)


def build(**kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return SpeechMixEED(ENC, LM, down_scale=2, compute_dtype="fp32", init_seed=1, **kw).train()


model = build(autograd_param_inputs=True)
twin = build()
ddp = DDP(model, device_ids=[0], find_unused_parameters=True)
names = [n for n, _ in model.named_parameters()]
g = torch.Generator().manual_seed(11)
# keep patterns per (step, rank): rank 0 and rank 1 drop DIFFERENT layers; step 2 drops nothing on rank 1
keep = {0: [[True, False, True, True], [True, True, False, True]],
        1: [[False, True, True, False], [True, True, True, True]],
        2: [[True, True, True, False], [False, True, True, True]]}
worst, checked, frozen_ok, dropped_seen = 0.0, 0, True, 0
enc_names = [n for n in names if n.startswith("encoder_model.encoder.layers.0.")]
for step in range(3):
    wave = (torch.randn(world * 2, 8000, generator=g) * 0.1).cuda()
    labels = torch.randint(4, 120, (world * 2, 5), generator=g).cuda()
    w, lab = wave[2 * rank:2 * rank + 2], labels[2 * rank:2 * rank + 2]
    # FreezingCallback-style flips (ref:speechmix/module/utility.py:14-29): step 1 trains with encoder layer 0 frozen, step 2 releases it
    for m in (model, twin):
        for n, p in m.named_parameters():
            if n in enc_names:
                p.requires_grad = step != 1
    for m in (model, twin):
        m.engine.host_rng = RecordedHostRNG(keep=keep[step][rank])
    ddp.zero_grad(set_to_none=True)
    loss = ddp(w, labels=lab)["loss"]
    loss.backward()                                   # DDP's bucketed all-reduce over gloo runs inside
    twin.zero_grad(set_to_none=True)
    twin(w, labels=lab)["loss"].backward()
    torch.cuda.synchronize()
    dropped_seen += sum(1 for k in keep[step][rank] if not k)
    assert model.engine.last_dropped == [i for i, k in enumerate(keep[step][rank]) if not k]
    local = twin.store.grad.clone()
    parts = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    mean = sum(parts) / world
    tp = dict(model.named_parameters())
    for n in names:
        o, k, _ = model.store.offsets[n]
        p = tp[n]
        if not p.requires_grad:
            frozen_ok &= p.grad is None
            continue
        want = mean[o:o + k].view_as(p)
        got = p.grad
        assert got is not None, n
        scale = max(want.abs().max().item(), 1e-6)
        worst = max(worst, (got - want).abs().max().item() / scale)
        checked += 1
h = torch.tensor([worst], dtype=torch.float64)
dist.all_reduce(h, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"world": world, "steps": 3, "tensors_checked": checked, "worst_rel_err": h.item(), "frozen_grads_none": frozen_ok,
                      "layers_dropped_on_rank0": dropped_seen, "loss": float(loss.item())}), flush=True)
dist.destroy_process_group()
