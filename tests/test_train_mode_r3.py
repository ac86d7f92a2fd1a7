"""Train-mode behaviour pinned to the reference (tests/golden/make_golden_r3.py: HFSpeechMixEED in `.train()` with every
dropout probability 0, under np.random.seed / torch.manual_seed).

CPU part (this file, not gpu): the SpecAugment span draw restated in speechmix_amd.engine.compute_mask_indices reproduces
HF's `_compute_mask_indices` BIT FOR BIT from the same legacy np.random stream (24 cases incl. ragged attention-mask
lengths, and the stream position afterwards), the LayerDrop draws come from torch's CPU generator in HF's order, and the
oracle handed the recorded decisions reproduces the reference's train-mode hidden states / logits / loss / gradients
(2e-5 abs, as test_oracle_golden.py).  The GPU part is tests/test_gpu_r3.py."""
import numpy as np
import pytest
import torch

from speechmix_amd.engine import HFHostRNG, RecordedHostRNG, compute_mask_indices
from tests.golden_util import GOLDEN, load_case
from tests.test_oracle_golden import _close, _run_with_grads


def _mask_cases():
    z = np.load(f"{GOLDEN}/mask_indices.npz")
    for i, row in enumerate(z["cases"]):
        B, T, prob, length, mmin, seed, nxt = row
        lens = z["lengths"][i]
        lens = None if lens[0] < 0 else [int(v) for v in lens if v >= 0]
        yield i, (int(B), int(T)), float(prob), int(length), int(mmin), int(seed), float(nxt), lens, z[f"m{i}"]


def test_specaugment_spans_are_hf_bit_for_bit_from_the_global_numpy_stream():
    n = 0
    for i, shape, prob, length, mmin, seed, nxt, lens, gold in _mask_cases():
        np.random.seed(seed)
        got = compute_mask_indices(shape, prob, length, HFHostRNG(), lens, mmin)
        assert got.dtype == bool and np.array_equal(got, gold), (i, shape, prob, length, lens, seed)
        assert np.random.rand() == nxt, "the draw count differs from HF's (stream position after the call)"
        priv = compute_mask_indices(shape, prob, length, HFHostRNG.seeded(seed), lens, mmin)         # private RandomState, same order
        assert np.array_equal(priv, gold)
        n += 1
    assert n == 24


def test_specaugment_rejects_what_hf_rejects():
    with pytest.raises(ValueError):
        compute_mask_indices((2, 8), 0.5, 10, HFHostRNG.seeded(0))
    with pytest.raises(ValueError):
        compute_mask_indices((2, 8), 0.5, 0, HFHostRNG.seeded(0))


@pytest.mark.parametrize("case", ["eed_train_specaug", "eed_train_layerdrop"])
def test_layerdrop_draws_are_hf_torch_rand_stream(case):
    z = np.load(f"{GOLDEN}/{case}.npz")
    seed, draws, keep = int(z["seed"]), z["layerdrop_draws"], z["layer_keep"]
    _, _, _, m = load_case(case)
    torch.manual_seed(seed)
    rng = HFHostRNG()
    got = np.array([rng.layerdrop() for _ in range(len(draws))])
    assert np.array_equal(got.astype(np.float32), draws.astype(np.float32))
    assert np.array_equal(got >= m["enc_cfg"]["layerdrop"], keep)                       # TF: skip <=> draw < layerdrop
    priv = HFHostRNG.seeded(seed)
    assert np.array_equal(np.array([priv.layerdrop() for _ in range(len(draws))]).astype(np.float32), draws.astype(np.float32))
    rec = RecordedHostRNG(keep=keep)
    assert [rec.layerdrop() >= 0.5 for _ in keep] == [bool(k) for k in keep]


@pytest.mark.parametrize("case", ["eed_train_specaug", "eed_train_layerdrop"])
def test_oracle_with_recorded_decisions_matches_reference_train_mode(case):
    sd, inp, gold, m = load_case(case)
    kw = dict(layer_keep=inp["layer_keep"].numpy())
    if "spec_mask" in inp:
        kw["spec_mask"] = inp["spec_mask"]
    sd, out, trace = _run_with_grads(sd, m, inp, **kw)
    _close(trace["feature_projection"], gold["feature_projection"], what="feature_projection")
    _close(out["encoder_last_hidden_state"], inp["encoder_hidden"], what="encoder_last_hidden_state")
    _close(out["inputs_embeds"], gold["inputs_embeds"], what="inputs_embeds")
    _close(out["raw_logits"], gold["raw_logits"], what="raw_logits")
    assert torch.equal(out["logits"], gold["logits"])
    assert abs(out["loss"].item() - gold["loss"].item()) < 1e-5
    out["loss"].backward()
    n = 0
    for k, g in gold.items():
        if k.startswith("grad::"):
            got = sd[k[6:]].grad
            if got is None:                          # a dropped layer: the reference leaves .grad None (stored as zeros)
                assert float(g.abs().max()) == 0.0, k
            else:
                _close(got, g, what=k)
            n += 1
    assert n >= 6
    if case == "eed_train_specaug":
        assert gold["grad::encoder_model.masked_spec_embed"].abs().max() > 1e-6
    else:
        assert int((~inp["layer_keep"]).sum()) >= 1 and len(inp["none_grads"]) >= 1
