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
