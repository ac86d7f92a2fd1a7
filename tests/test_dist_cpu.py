"""world_size-2 gloo tests of the data-parallel path (runs on CPU): bucketed flat-gradient all-reduce,
batch sharding, and that the reduction is independent of the order in which stages are reported."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from speechmix_amd.dist import GradReducer, shard_batch, stage_ranges
        # a synthetic flat layout with the same naming scheme as the real store
        offsets, off = {}, 0
        names = ["decoder_model.model.shared.weight", "decoder_model.model.decoder.layers.0.fc1.weight",
                 "length_adapters.0.weight", "enc_to_dec_proj.weight",
                 "encoder_model.encoder.layers.0.attention.q_proj.weight", "encoder_model.encoder.layers.1.attention.q_proj.weight",
                 "encoder_model.feature_extractor.conv_layers.0.conv.weight", "encoder_model.masked_spec_embed"]
        for i, n in enumerate(names):
            numel = 100 + 37 * i
            offsets[n] = (off, numel, (numel,))
            off = (off + numel + 63) // 64 * 64
        total = off
        g = torch.full((total,), float(rank + 1))
        torch.manual_seed(rank)
        g += torch.randn(total)
        ref = g.clone()
        dist.all_reduce(ref)
        red = GradReducer(g, stage_ranges(offsets, 2))
        red.begin_step()
        order = ["lm", "bridge", "enc_layer1", "enc_layer0", "frontend"] if rank == 0 else None
        # stages are reported in backward order; "frontend" is left to finish() to cover that path
        for s in ["lm", "bridge", "enc_layer1", "enc_layer0"]:
            red.stage_done(s)
        red.stage_done("lm")          # idempotent within a step
        red.finish()
        ok = True
        for n, (o, k, _) in offsets.items():
            ok &= torch.allclose(g[o:o + k], ref[o:o + k])
        lo, hi = shard_batch(64, rank, world)
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == (0, True, (0, 32)) and res[1] == (1, True, (32, 64))


def _worker_real_layout(rank, world, port, q):
    """GradReducer over the REAL config-2 FlatStore layout (wav2vec2-base + bart-base: 235.6 M parameters, q|k|v adjacency,
    alignment padding, tied embedding stored once) with per-rank-different LayerDrop sets and grad_accum = 2: micro-batch
    gradients add up locally, only the last micro-batch reports stages (StepRunner.step), dropped layers contribute zeros."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib, io
        from speechmix_amd.dist import GradReducer, stage_ranges
        from speechmix_amd.model import SpeechMixFixed
        from speechmix_amd.params import FlatStore
        torch.set_num_threads(2)
        with contextlib.redirect_stdout(io.StringIO()):
            # SpeechMixFixed(fixed_nlp) = frozen LM: its 139 M parameters must not be reduced at all
            model = SpeechMixFixed("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, fixed_nlp=True)
        store = FlatStore(model, "cpu", torch.float32)
        L = model.num_speech_encoder_layers
        stages = stage_ranges(store.offsets, L, trainable=store.requires_grad)
        red = GradReducer(store.grad, stages)
        covered = sum(b - a for _, rs in stages for a, b in rs)
        lm_elems = sum(n for nm, (o, n, _) in store.offsets.items() if nm.startswith("decoder_model."))
        trainable_elems = sum(n for nm, (o, n, _) in store.offsets.items() if store.requires_grad(nm))
        assert covered >= trainable_elems and covered < trainable_elems + 64 * len(store.offsets), (covered, trainable_elems)
        assert covered < store.total - lm_elems + 64 * len(store.offsets)            # the frozen LM is outside every bucket
        pre = "encoder_model.encoder.layers."
        layer_of = {nm: int(nm[len(pre):].split(".")[0]) for nm in store.offsets if nm.startswith(pre)}
        drops = {0: [{1, 5}, {5, 7}], 1: [{2}, {5, 11}]}          # [rank][micro-batch]: LayerDrop decisions differ per rank
        order = ["lm", "bridge"] + [f"enc_layer{i}" for i in range(L - 1, -1, -1)] + ["frontend"]
        red.begin_step()
        store.grad.zero_()
        for micro in range(2):
            for nm, (o, n, _) in store.offsets.items():
                if not store.requires_grad(nm) or layer_of.get(nm, -1) in drops[rank][micro]:
                    continue
                store.grad[o:o + n] += 0.5 * (rank + 1) * (micro + 1)           # (gscale = 1 / grad_accum folded in)
            if micro == 1:                                                       # stages are reported on the last micro-batch only
                for s in order[:-1]:
                    red.stage_done(s)
        red.finish()
        ok, checked = True, 0
        for nm, (o, n, _) in store.offsets.items():
            if not store.requires_grad(nm):
                ok &= bool((store.grad[o:o + n] == 0).all())
                continue
            want = 0.0
            for r in range(world):
                for micro in range(2):
                    if layer_of.get(nm, -1) not in drops[r][micro]:
                        want += 0.5 * (r + 1) * (micro + 1)
            ok &= bool(torch.allclose(store.grad[o:o + n], torch.full((n,), want)))
            checked += 1
        q.put((rank, bool(ok), checked > 150, covered))
    finally:
        dist.destroy_process_group()


def test_reducer_on_real_config2_layout_with_layerdrop_and_grad_accum():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real_layout, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1] and res[0][2] and res[0][3] == res[1][3], res


def _worker_modes(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from speechmix_amd import ops
        from speechmix_amd.dist import reduce_bucket, share_tuner_picks
        torch.manual_seed(7 + rank)
        g = torch.randn(1003)
        ref = g.clone()
        dist.all_reduce(ref)
        out = {}
        for mode in ("sum", "rs_ag", "bf16"):          # rs_ag on gloo: the shard arithmetic on all_reduce + all_gather (dist.py)
            t = g.clone()
            reduce_bucket(t, mode=mode)
            out[mode] = (t - ref).abs().max().item()
        # rs_ag with a bucket shorter than the world size, exactly divisible, and with a remainder of 1
        for n in (1, 2, 1002):
            torch.manual_seed(100 + rank)
            t = torch.randn(n)
            want = t.clone()
            dist.all_reduce(want)
            reduce_bucket(t, mode="rs_ag")
            out[f"rs_ag_{n}"] = (t - want).abs().max().item()
        # tuner picks: rank 1 adopts rank 0's choice for a shared key and keeps its own extra key
        ops._TUNED.clear()
        ops._tuned_set(("shape", 1), 8 if rank == 0 else 12)
        if rank == 1:
            ops._tuned_set(("only_rank1",), 13)
        share_tuner_picks()
        picks = (ops._tuned_get(("shape", 1)), ops._tuned_get(("only_rank1",)))
        q.put((rank, out, picks))
    finally:
        dist.destroy_process_group()


def test_allreduce_modes_and_shared_tuner_picks_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_modes, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, picks in res:
        assert out["sum"] == 0.0 and out["rs_ag"] == 0.0
        assert out["rs_ag_1"] == 0.0 and out["rs_ag_2"] == 0.0 and out["rs_ag_1002"] == 0.0
        assert 0.0 < out["bf16"] < 5e-2                       # bf16 rounding of the summed gradient, nothing worse
    assert res[0][2] == (8, None) and res[1][2] == (8, 13)


# ------------------------------------------------------------------------------------------------ 8 ranks
def _worker8(rank, world, port, q):
    """GradReducer + reduce_bucket(mode="rs_ag") at the world size of BASELINE config 3 (TF:trainer.py:720-737's DDP over 8 ranks): bucket
    sizes that do not divide by 8, a frozen LM, per-rank LayerDrop (a rank that dropped a layer contributes zeros for it and still takes
    part in its collectives) and ranks that report different prefixes of the stage order before finish()."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import speechmix_amd.dist as D
        D.ALLREDUCE_MODE = "rs_ag"
        log = []
        real_ar, real_ag = dist.all_reduce, dist.all_gather

        def ar(t, *a, **k):
            log.append(("all_reduce", t.numel()))
            return real_ar(t, *a, **k)

        def ag(lst, t, *a, **k):
            log.append(("all_gather", t.numel(), len(lst)))
            return real_ag(lst, t, *a, **k)
        dist.all_reduce, dist.all_gather = ar, ag
        L = 3
        offsets, off = {}, 0
        names = ["decoder_model.model.shared.weight", "decoder_model.model.decoder.layers.0.fc1.weight", "length_adapters.0.weight",
                 "enc_to_dec_proj.weight"] + [f"encoder_model.encoder.layers.{i}.attention.q_proj.weight" for i in range(L)] + \
                ["encoder_model.feature_extractor.conv_layers.0.conv.weight", "encoder_model.masked_spec_embed"]
        for i, n in enumerate(names):
            numel = 1003 + 37 * i                          # never a multiple of 8
            offsets[n] = (off, numel, (numel,))
            off = (off + numel + 63) // 64 * 64
        total = off
        frozen = lambda n: n.startswith("decoder_model.")                   # SpeechMixFixed-style: the LM is not reduced
        gen = torch.Generator().manual_seed(100 + rank)
        g = torch.randn(total, generator=gen)
        dropped = rank % L                                                   # this rank's LayerDrop draw
        o, k, _ = offsets[f"encoder_model.encoder.layers.{dropped}.attention.q_proj.weight"]
        g[o:o + k] = 0.0
        # the single-process answer: the sum of every rank's buffer
        want = torch.zeros(total)
        for r in range(world):
            gr = torch.randn(total, generator=torch.Generator().manual_seed(100 + r))
            oo, kk, _ = offsets[f"encoder_model.encoder.layers.{r % L}.attention.q_proj.weight"]
            gr[oo:oo + kk] = 0.0
            want += gr
        mine = g.clone()
        stages = D.stage_ranges(offsets, L, trainable=lambda n: not frozen(n))
        red = D.GradReducer(g, stages)
        red.begin_step()
        order = [s for s, _ in stages]
        for s in order[:1 + rank % 4]:                                       # a rank-dependent prefix; finish() takes the rest in order
            red.stage_done(s)
        red.finish()
        ok = True
        for n, (o, k, _) in offsets.items():
            if frozen(n):
                ok &= bool(torch.equal(g[o:o + k], mine[o:o + k]))          # untouched
            else:
                ok &= bool(torch.allclose(g[o:o + k], want[o:o + k], rtol=1e-5, atol=1e-5))
        q.put((rank, ok, log))
    finally:
        dist.destroy_process_group()


def test_rs_ag_buckets_gloo_world8_non_divisible_frozen_layerdrop():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(ok for _, ok, _ in res), [r for r, ok, _ in res if not ok]
    logs = [lg for _, _, lg in res]
    assert all(lg == logs[0] for lg in logs), "ranks issued different collective sequences"
    kinds = {e[0] for e in logs[0]}
    assert kinds == {"all_reduce", "all_gather"} and len(logs[0]) >= 10          # shard exchange + non-divisible remainders
