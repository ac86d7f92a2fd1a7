"""Host-side logic added in round 2 (CPU): checkpoint key conversion (fairseq / s3prl names, pytorch_model.bin), the
SpeechMixAdapter constructor, the reference's length filter / length-grouped order, the learning-rate schedule, update
ranges under LayerDrop, trainable-only reduction buckets, and bench.py starting its own ranks."""
import json
import os
import re
import subprocess
import sys

import pytest
import torch

from tests.golden_util import load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ checkpoints
def _hf_to_fairseq(k):
    """Inverse of the HF conversion table (TF:models/wav2vec2/convert_wav2vec2_original_pytorch_checkpoint_to_pytorch.py),
    written out independently of speechmix_amd.checkpoint: builds a fairseq-named state dict from an HF-named one."""
    m = re.match(r"feature_extractor\.conv_layers\.(\d+)\.(conv|layer_norm)\.(weight|bias)$", k)
    if m:
        i, kind, leaf = m.groups()
        return f"feature_extractor.conv_layers.{i}.{0 if kind == 'conv' else 2}.{leaf}"
    k = k.replace("feature_projection.projection", "post_extract_proj").replace("feature_projection.layer_norm", "layer_norm")
    k = k.replace("encoder.pos_conv_embed.conv.parametrizations.weight.original0", "encoder.pos_conv.0.weight_g")
    k = k.replace("encoder.pos_conv_embed.conv.parametrizations.weight.original1", "encoder.pos_conv.0.weight_v")
    k = k.replace("encoder.pos_conv_embed.conv.bias", "encoder.pos_conv.0.bias")
    k = k.replace("attention.", "self_attn.").replace("feed_forward.intermediate_dense", "fc1").replace("feed_forward.output_dense", "fc2")
    k = re.sub(r"(encoder\.layers\.\d+)\.layer_norm", r"\1.self_attn_layer_norm", k)
    return "mask_emb" if k == "masked_spec_embed" else k


def test_fairseq_and_bin_checkpoints_load_by_key_name(tmp_path):
    from speechmix_amd import checkpoint as ck
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    enc = {k[len("encoder_model."):]: v for k, v in sd.items() if k.startswith("encoder_model.")}
    fair = {"w2v_encoder.w2v_model." + _hf_to_fairseq(k): v for k, v in enc.items()}
    fair["w2v_encoder.w2v_model.quantizer.vars"] = torch.zeros(3)           # parameters SpeechMix never uses
    fair["w2v_encoder.w2v_model.final_proj.weight"] = torch.zeros(2, 2)
    for k in fair:                                                          # every name maps back
        if "quantizer" in k or "final_proj" in k:
            assert ck.hf_key_from_fairseq(k) is None
        else:
            assert ck.hf_key_from_fairseq(k) in enc, k
    # "layer"-norm feature extractor: fairseq wraps the LayerNorm in a Sequential -> one more index
    assert ck.hf_key_from_fairseq("feature_extractor.conv_layers.3.2.1.weight") == "feature_extractor.conv_layers.3.layer_norm.weight"
    pt = tmp_path / "wav2vec_small.pt"
    torch.save({"model": fair, "cfg": None}, pt)                            # fairseq nests the weights under "model"
    lm_dir = tmp_path / "lm"
    lm_dir.mkdir()
    lm = {k[len("decoder_model."):]: v for k, v in sd.items() if k.startswith("decoder_model.")}
    torch.save(lm, lm_dir / "pytorch_model.bin")                            # what ref:eval.py:10-style checkpoints hold
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, speech_checkpoint=str(pt), nlp_checkpoint=str(lm_dir))
    own = model.state_dict()
    for k, v in sd.items():
        if k.startswith(("encoder_model.", "decoder_model.")):
            assert torch.equal(own[k].cpu(), v), k
    # whole-model state dict as ref:speechmix/model.py saves it (s3prl wrapper: `encoder_model.model.` + fairseq names)
    whole = {("encoder_model.model." + _hf_to_fairseq(k[len("encoder_model."):]) if k.startswith("encoder_model.") else k): v
             for k, v in sd.items()}
    model2 = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2)
    missing, unexpected = model2.load_state_dict(whole, strict=False)
    assert not unexpected and all(k == "weights_sum" for k in missing), (missing, unexpected)
    for k, v in sd.items():
        assert torch.equal(model2.state_dict()[k].cpu(), v), k


def test_speechmix_adapter_constructor_structure():
    from speechmix_amd.model import SpeechMixAdapter
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    model = SpeechMixAdapter(m["enc_cfg"], m["lm_cfg"], down_scale=2)
    lc = m["lm_cfg"]
    n = lc["encoder_layers"] + lc["decoder_layers"]
    assert len(model.adapters) == n
    names = dict(model.named_parameters())
    d = lc["d_model"]
    for i in range(n):                         # nn.Sequential(LayerNorm, Linear, ReLU, Linear) state-dict names
        assert names[f"adapters.{i}.0.weight"].shape == (d,)
        assert names[f"adapters.{i}.1.weight"].shape == (d // 2, d)
        assert names[f"adapters.{i}.3.weight"].shape == (d, d // 2)
        assert names[f"adapters.{i}.3.bias"].shape == (d,)
    frozen = [k for k, p in names.items() if not p.requires_grad]
    assert frozen and all(".layers." in k and k.startswith("decoder_model.model.") for k in frozen)
    assert names["decoder_model.model.shared.weight"].requires_grad            # embeddings stay trainable, as in the reference
    assert all(names[f"adapters.{i}.1.weight"].requires_grad for i in range(n))
    assert sorted(model.list_no_grad) == sorted(frozen)


# ------------------------------------------------------------------------------------------------ input side
def test_length_filter_and_length_grouped_order():
    from speechmix_amd.data import DataCollatorWithPadding, bucketed_batches, filter_by_length, length_grouped_indices
    lengths = [16000, 16001, 15999, 320000, 319999, 50000]
    assert filter_by_length(lengths, 20) == [1, 4, 5]            # strict on both sides (ref:train.py:279-281)
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(16001, 320000, (257,), generator=g).tolist()
    order = length_grouped_indices(lens, 8, generator=torch.Generator().manual_seed(1))
    assert sorted(order) == list(range(257))                      # a permutation
    assert lens[order[0]] == max(lens)                            # the longest clip leads
    mb = min(257 // 32, 50) * 8
    for i in range(0, 257, mb):                                   # every mega-batch is sorted longest-first (after its head)
        seg = [lens[j] for j in order[i:i + mb]][1:]
        assert seg == sorted(seg, reverse=True)
    # padding waste: batches of 8 consecutive clips vs. random batches
    def waste(o):
        return sum(max(lens[j] for j in o[i:i + 8]) * len(o[i:i + 8]) - sum(lens[j] for j in o[i:i + 8]) for i in range(0, 257, 8))
    assert waste(order) < 0.25 * waste(list(range(257)))

    class Tok:
        pad_token_id, bos_token_id = 1, 0
    ds = [{"input_values": torch.zeros(n), "labels": [5, 6, 2]} for n in lens[:64]]
    seen = []
    for rank in range(2):
        for b in bucketed_batches(ds, DataCollatorWithPadding(Tok()), 4, lengths=lens[:64], rank=rank, world=2,
                                  generator=torch.Generator().manual_seed(3)):
            assert b["input_values"].shape[0] == 4
            seen.append(b["input_values"].shape[1])
    assert len(seen) == 16                                        # 64 clips = 8 global batches x 2 ranks
    # no generator given, two ranks: the permutation comes from (seed, epoch), not from each rank's own global RNG - the ranks
    # partition every global batch (no clip twice, none missing) and the strided split balances the lengths between them
    ds2 = [{"input_values": torch.full((n,), float(i)), "labels": [5, 6, 2]} for i, n in enumerate(lens[:64])]
    per_rank = []
    for rank in range(2):
        torch.manual_seed(100 + rank)                             # different global RNG states per rank must not matter
        ids, tot = [], 0
        for b in bucketed_batches(ds2, DataCollatorWithPadding(Tok()), 4, lengths=lens[:64], rank=rank, world=2, seed=5, epoch=1):
            ids += [int(r[0].item()) for r in b["input_values"]]
            tot += b["input_values"].shape[1]
        per_rank.append((ids, tot))
    assert sorted(per_rank[0][0] + per_rank[1][0]) == list(range(64))
    assert abs(per_rank[0][1] - per_rank[1][1]) < 0.1 * per_rank[0][1]


def test_linear_schedule_and_update_ranges_under_layerdrop():
    from speechmix_amd.trainer import linear_schedule_with_warmup, trainable_ranges
    f = linear_schedule_with_warmup(5e-4, 500, 10000)
    assert f(1) == 0.0 and abs(f(251) - 2.5e-4) < 1e-12 and abs(f(501) - 5e-4) < 1e-12
    assert abs(f(5251) - 2.5e-4) < 1e-12 and f(10001) == 0.0

    class P:
        def __init__(self, rg): self.requires_grad = rg

    class Store:
        offsets = {"a": (0, 64, (64,)), "encoder_model.encoder.layers.0.w": (64, 64, (64,)),
                   "encoder_model.encoder.layers.1.w": (128, 64, (64,)), "frozen": (192, 64, (64,)), "z": (256, 64, (64,))}
        params = {k: P(k != "frozen") for k in offsets}
        def requires_grad(self, n): return self.params[n].requires_grad
    st = Store()
    layer_of = {"encoder_model.encoder.layers.0.w": 0, "encoder_model.encoder.layers.1.w": 1}
    assert trainable_ranges(st) == [(0, 192), (256, 320)]
    assert trainable_ranges(st, {0}, layer_of) == [(0, 64), (128, 192), (256, 320)]     # dropped layer: not updated
    assert trainable_ranges(st, {0, 1}, layer_of) == [(0, 64), (256, 320)]


def test_reduction_buckets_cover_trainable_parameters_only():
    from speechmix_amd.dist import stage_ranges
    offsets = {"decoder_model.a": (0, 100, (100,)), "decoder_model.b": (128, 100, (100,)), "adapters.0.1.weight": (256, 64, (64,)),
               "enc_to_dec_proj.weight": (320, 64, (64,)), "encoder_model.encoder.layers.0.w": (384, 64, (64,)),
               "encoder_model.feature_projection.w": (448, 64, (64,))}
    frozen = {"decoder_model.a", "decoder_model.b"}
    st = dict(stage_ranges(offsets, 1, trainable=lambda n: n not in frozen))
    assert st["lm"] == [(256, 320)]                    # the frozen LM is not reduced; adapters ride in the LM stage
    assert st["bridge"] == [(320, 384)] and st["enc_layer0"] == [(384, 448)] and st["frontend"] == [(448, 512)]
    st_all = dict(stage_ranges(offsets, 1))
    assert st_all["lm"][0][0] == 0


# ------------------------------------------------------------------------------------------------ bench launch path
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 workers itself (it asserted)."""
    env = dict(os.environ, SMX_BENCH_SPAWN_ONLY="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    # every rank's object, wherever it landed on the shared pipe (the objects hold no nested braces)
    lines = [json.loads(m) for m in re.findall(r'\{"spawn_check"[^{}]*\}', r.stdout)]
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert all(l["world"] == 2 and l["gpus"] == 2 and l["rank_sum"] == 1.0 for l in lines)


def test_s3prl_layout_fairseq_namespace_and_empty_loads(tmp_path):
    """Round-2 advisor finding: s3prl-converted upstream checkpoints keep the weights under `model_weight` (next to task_cfg /
    model_cfg), real fairseq files carry an argparse.Namespace, and a checkpoint that matches nothing must not leave the model
    silently at its random initialisation."""
    import argparse
    import pytest
    from speechmix_amd import checkpoint as ck
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case("eed_w2v2_bart")
    enc = {k[len("encoder_model."):]: v for k, v in sd.items() if k.startswith("encoder_model.")}
    fair = {_hf_to_fairseq(k): v for k, v in enc.items()}
    s3 = tmp_path / "s3prl_wav2vec2.pt"
    torch.save({"task_cfg": {"sample_rate": 16000}, "model_cfg": {"encoder_layers": 4}, "model_weight": fair}, s3)
    got = ck.read_state_file(str(s3))
    assert set(got) == set(fair)
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, speech_checkpoint=str(s3))
    own = model.state_dict()
    for k, v in sd.items():
        if k.startswith("encoder_model."):
            assert torch.equal(own[k].cpu(), v), k
    fs = tmp_path / "fairseq_with_args.pt"
    torch.save({"args": argparse.Namespace(arch="wav2vec2", encoder_layers=4), "model": fair}, fs)
    assert set(ck.read_state_file(str(fs))) == set(fair)
    junk = tmp_path / "other_model.pt"
    torch.save({"model": {"totally.unrelated.weight": torch.zeros(3, 3)}}, junk)
    with pytest.raises(RuntimeError, match="none of its"):
        SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, speech_checkpoint=str(junk))
    empty = tmp_path / "empty_dir"
    empty.mkdir()
    with pytest.raises(FileNotFoundError):
        ck.read_checkpoint(str(empty))


def test_adafactor_plan_layout_gives_every_partial_sum_its_own_slot():
    """The fixed-order reductions of the fused Adafactor step (csrc/adafactor.hip) rely on the host plan: a tensor's tiles are
    contiguous, every row tile owns one row of its segment's [n_rt][C] block of column partials, every column tile one row of
    its [n_ct][R] block of row sums, and no two tiles (or blocks) overlap in the scratch buffer."""
    from speechmix_amd.ops import AdafactorPlan
    shapes = [(768, 3072), (3072, 768), (512, 512, 3), (3072,), (1, 5027), (768, 48, 128), (50265, 768), (100, 4100), (768,), (64, 64)]
    offs, total = [], 0
    for s in shapes:
        offs.append(total)
        n = 1
        for d in s:
            n *= d
        total += (n + 63) // 64 * 64
    plan = AdafactorPlan(list(zip(offs, shapes)), torch.device("cpu"))
    tiles, segs = plan.tiles.tolist(), plan.segs.tolist()
    import numpy as np
    tt = np.frombuffer(plan.tensors.numpy().tobytes(), dtype=np.dtype([("off", "<i8"), ("nb", "<i4"), ("R", "<i4"), ("C", "<i4"),
                                                                       ("row_off", "<i4"), ("col_off", "<i4"), ("rm_off", "<i4"),
                                                                       ("factored", "<i4"), ("tile0", "<i4"), ("ntile", "<i4"), ("_pad", "<i4")]))
    used = np.zeros(plan.cpart.numel(), dtype=np.int32)
    covered = 0
    for t, T in enumerate(tt):
        mine = tiles[T["tile0"]:T["tile0"] + T["ntile"]]
        assert mine and all(tl[0] == t for tl in mine)
        assert T["tile0"] == covered
        covered += T["ntile"]
        if T["factored"]:
            cells = np.zeros((T["nb"], T["R"], T["C"]), dtype=np.int32)
            for (_, b, r0, nr, c0, nc, full_rows, full_cols, cp_off, rp_off) in mine:
                cells[b, r0:r0 + nr, c0:c0 + nc] += 1
                if not full_cols:
                    used[cp_off:cp_off + nc] += 1
                if not full_rows:
                    used[rp_off + r0:rp_off + r0 + nr] += 1
            assert (cells == 1).all()                                # the tiles partition every matrix
    assert covered == len(tiles)
    assert used.max() <= 1                                           # no slot written twice
    for (t, b, cp_off, n_rt, rp_off, n_ct) in segs:
        T = tt[t]
        if n_rt > 1:
            assert (used[cp_off:cp_off + n_rt * T["C"]] == 1).all()  # the fold reads exactly what the tiles wrote
        if n_ct > 1:
            assert (used[rp_off:rp_off + n_ct * T["R"]] == 1).all()
        assert n_rt <= AdafactorPlan.MAX_RT
