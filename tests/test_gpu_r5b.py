"""Round-5 GPU tests, part b (run on the MI355X: `pytest -m gpu`).

* `speechmix_amd.optim.FusedAdafactor` (the `torch.optim.Optimizer` front of the fused multi-tensor Adafactor step) against
  `transformers.optimization.Adafactor` with HF Trainer's settings, stepped on the same model gradients;
* `DistributedDataParallel(model, find_unused_parameters=True)` around the HIP model - the reference's real multi-GPU wrapper -
  two ranks over gloo on the shared GPU, LayerDrop with different patterns per rank, requires_grad flips (item 7).
"""
import contextlib
import io
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENC = dict(model_type="wav2vec2", hidden_size=64, num_hidden_layers=3, num_attention_heads=2, intermediate_size=128,
           conv_dim=[32] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2], num_conv_pos_embeddings=16,
           num_conv_pos_embedding_groups=4, layerdrop=0.0, mask_time_prob=0.0, hidden_dropout=0.0, attention_dropout=0.0,
           activation_dropout=0.0, feat_proj_dropout=0.0)
LM = dict(model_type="bart", vocab_size=120, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
          decoder_attention_heads=2, encoder_ffn_dim=128, decoder_ffn_dim=128, max_position_embeddings=128, dropout=0.0,
          attention_dropout=0.0, activation_dropout=0.0)


def _build():
    from speechmix_amd.model import SpeechMixEED
    with contextlib.redirect_stdout(io.StringIO()):
        return SpeechMixEED(ENC, LM, down_scale=2, compute_dtype="fp32", init_seed=2).train()


def test_fused_adafactor_optimizer_reproduces_transformers_adafactor():
    """Five steps of `loss.backward(); clip; optimizer.step(); zero_grad` - HF's Adafactor (scale_parameter=False,
    relative_step=False: what Trainer's optim="adafactor" builds) after torch's clip_grad_norm_ on one model, FusedAdafactor with
    max_grad_norm folded in on its twin - leave the same parameters (fp32: 5e-4 of each tensor's range after five steps at lr 1e-2; the two models'
    gradients differ run to run in the last bits - fp32 atomics - and Adafactor's g / RMS(g) makes lr-sized steps of that wherever a tensor's gradient is
    nearly zero: the worst tensor, a cross-attention query bias, was seen at 1.8e-4 - 2.3e-4 of its range), with a frozen tensor
    untouched and the learning rate read from param_groups every step (a scheduler's hook)."""
    transformers = pytest.importorskip("transformers")
    from transformers.optimization import Adafactor
    from speechmix_amd.optim import FusedAdafactor
    a, b = _build(), _build()
    for m in (a, b):
        dict(m.named_parameters())["enc_to_dec_proj.bias"].requires_grad = False
    hf = Adafactor([p for p in a.parameters() if p.requires_grad], lr=1e-2, scale_parameter=False, relative_step=False, warmup_init=False)
    a.store.external_updates = True
    fu = FusedAdafactor(b, lr=1e-2, max_grad_norm=0.5)
    sched = torch.optim.lr_scheduler.LambdaLR(fu, lambda s: 1.0 / (1 + s))
    g = torch.Generator().manual_seed(0)
    p0 = b.store.master.clone()
    for step in range(5):
        wave = (torch.randn(3, 9000, generator=g) * 0.1).cuda()
        labels = torch.randint(4, 120, (3, 6), generator=g).cuda()
        for grp in hf.param_groups:
            grp["lr"] = 1e-2 / (1 + step)
        la = a(wave, labels=labels)["loss"]
        la.backward()
        torch.nn.utils.clip_grad_norm_([p for p in a.parameters() if p.requires_grad], 0.5)
        hf.step()
        a.zero_grad(set_to_none=True)
        lb = b(wave, labels=labels)["loss"]
        lb.backward()
        fu.step()
        sched.step()
        b.zero_grad(set_to_none=True)
        assert abs(la.item() - lb.item()) <= 2e-4 * max(1.0, abs(la.item())), (step, la.item(), lb.item())
    torch.cuda.synchronize()
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    moved = 0
    skipped = 0
    for n in pa:
        x, y = pa[n].detach(), pb[n].detach()
        o, k, _ = b.store.offsets[n]
        moved += int(not torch.equal(p0[o:o + k].view_as(y), y))
        if n.endswith("k_proj.bias"):
            # a key bias shifts every score of a query row alike: softmax cancels it, its gradient is exactly 0 in exact arithmetic
            # and rounding noise in fp32 - which Adafactor's update / RMS(update) turns into lr-sized steps of random sign
            skipped += 1
            continue
        scale = max(x.abs().max().item(), 1e-3)
        assert (x - y).abs().max().item() <= 5e-4 * scale + 1e-7, (n, (x - y).abs().max().item(), scale)
    assert skipped == 3 + 2 + 2 * 2
    o, k, _ = b.store.offsets["enc_to_dec_proj.bias"]
    assert torch.equal(p0[o:o + k], b.store.master[o:o + k])                      # frozen: untouched
    assert moved >= len(pa) - 3
    # the optimizer state travels through state_dict (Trainer writes optimizer.pt at every checkpoint)
    sd = fu.state_dict()
    fu2 = FusedAdafactor(_build(), lr=1e-2)
    fu2.load_state_dict(sd)
    assert torch.equal(fu2.plan.row, fu.plan.row) and torch.equal(fu2.plan.col, fu.plan.col) and (fu2.plan.steps == fu.plan.steps).all()


def test_ddp_wrapped_model_two_ranks_layerdrop_and_freezing():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "gpu_ddp_check.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(d)
    assert d["tensors_checked"] > 150 and d["layers_dropped_on_rank0"] >= 3 and d["frozen_grads_none"]
    assert d["worst_rel_err"] <= 1e-5, d


def test_two_ranks_on_one_gpu_with_the_replayed_step():
    """The N > 1 step with forward + backward replayed from captured graphs (SMX_STEP_GRAPHS=1): the replay reports every finished
    stage to dist.GradReducer BETWEEN two graphs, exactly where the eager backward does, so the bucketed reduction still overlaps
    the rest of backward.  Two ranks on cuda:0 over gloo through bench.py's own launch path, per-rank LayerDrop / SpecAugment
    draws, 6 set-up + 2 warm-up + 4 timed steps (the capture happens after five eager steps when world > 1): one JSON line,
    `host.step_graphs` true, parameters bit-identical on both ranks after the steps, a finite loss."""
    env = dict(os.environ, SMX_BENCH_SHARED_GPU="1", SMX_BENCH_CHECK_SYNC="1", SMX_STEP_GRAPHS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "4",
                        "--no-cpu-baseline", "--no-eval-leg", "--no-profile", "--seed", "3"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["host"]["step_graphs"] is True and d["host"]["graphs_per_step"] >= 20, d["host"]
    assert d.get("params_in_sync") is True, d
    assert d["final_loss"] == d["final_loss"] and 0 < d["final_loss"] < 50


def test_256_wide_gemm_kernels_match_the_128_kernel_bit_for_bit_on_ragged_shapes():
    """The persistent 256-wide kernels (tr_mode 8 ping-pong, 12 / 13 free-running with 256 / 192-row tiles) against the 128 x 128 kernel
    (tr_mode 1): same K order and the same epilogue arithmetic -> bit-identical outputs, for every epilogue class the step uses (bias +
    residual, GELU with dropout and the pre-activation / the saved derivative as the second output, the two data-gradient classes
    with a rows-contiguous weight, the fp32 weight gradient with a K split) on ragged M / N / K, each launch twice (the second one finds
    the LDS stages, the kernarg cache and - round 5 - the lane-exchanged stores, the early-issued K tile and the tail K tile of K % 64 != 0
    in a warm state).  A seeded excerpt of tools/gpu_gemm_fuzz.py (2 000 cases there, 36 here)."""
    import random
    from speechmix_amd import ops
    from speechmix_amd.ops import ACT_GELU, view
    dev = torch.device("cuda:0")
    rng = random.Random(11)
    torch.manual_seed(11)
    kinds = ["fwd", "fwd_act", "fwd_saved", "dgrad", "dgrad_actgrad", "dgrad_saved", "wgrad"]
    compared = 0
    for case in range(36):
        M = rng.choice([264, 1000, 4000, 7968, 15968]) + 8 * rng.randrange(0, 4)
        N = 8 * rng.randrange(8, 400)
        K = 8 * rng.randrange(4, 200)
        kind = kinds[case % len(kinds)]
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Wt = W.t().contiguous()
        bias = torch.randn(N, device=dev) * 0.1
        S = torch.randn(M, N, device=dev).bfloat16()
        ref, split = None, rng.choice([1, 3])
        for mode in (1, 8, 12, 13, 13):
            try:
                if kind == "wgrad":
                    kst = (M + 63) // 64
                    per = (kst + split - 1) // split
                    sp = (kst + per - 1) // per
                    G = torch.zeros(sp, N, K, dtype=torch.float32, device=dev)
                    ops.gemm(S, A, G, N, K, M, ops.BF16, a_rc=True, b_rc=True, av=view(N), bv=view(K), out_f32=True, split_k=sp,
                             split_stride=N * K if sp > 1 else 0, tr_mode=mode)
                    res = (G,)
                else:
                    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
                    aux = torch.zeros_like(Y)
                    kw = {"fwd": dict(bias=bias, resid=S, drop=(0.1, 4)), "fwd_act": dict(bias=bias, act=ACT_GELU, aux_out=aux, drop=(0.1, 5)),
                          "fwd_saved": dict(bias=bias, act=ACT_GELU | ops.ACT_SAVE_GRAD, aux_out=aux, drop=(0.1, 6)),
                          "dgrad": dict(b_rc=True, bv=view(N), resid=S), "dgrad_actgrad": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU),
                          "dgrad_saved": dict(b_rc=True, bv=view(N), aux_in=S, act=ACT_GELU | ops.ACT_SAVE_GRAD)}[kind]
                    ops.gemm(A, Wt if kw.get("b_rc") else W, Y, M, N, K, ops.BF16, tr_mode=mode, **kw)
                    res = (Y, aux)
            except RuntimeError:          # (a layout / class a kernel family does not instantiate: the launcher refuses, the tuner never offers it)
                continue
            if mode == 1:
                ref = res
            else:
                compared += 1
                for a_, b_ in zip(ref, res):
                    assert torch.equal(a_, b_), (case, kind, M, N, K, mode, (a_.float() - b_.float()).abs().max().item())
    assert compared >= 100, compared
