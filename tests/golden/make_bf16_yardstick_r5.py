"""The bf16 YARDSTICK of the full-dimension reference fixtures (round 5).

What error does bf16 storage + fp32 accumulation cost on the reference's own arithmetic?  The CPU oracle (pinned to the reference's
outputs to ~2e-6 by tests/test_full_dimension_r4.py) is run with every weight and activation in torch.bfloat16 on the SAME seeded
weights and inputs as tests/golden/full_cfg*.npz and compared with those reference fixtures through the test's own `compare()`:
logits, loss, worst hidden state, worst gradient entry, worst gradient norm.  The HIP bf16 path is then held to 1.5 x these
figures (tests/test_full_dimension_r4.py BF16 rows) - a criterion that comes from the arithmetic, not from the run under test.

Second part (the losses of configs 4 and 5, item 8 / weak item 2): the loss is ONE scalar per run, so one draw of its bf16 error says
little - the oracle's fp32 and bf16 losses on THREE further seeded inputs of each (forward only) are recorded, and the GPU test
compares the HIP bf16 path's loss errors on the same three inputs with them.

    python tests/golden/make_bf16_yardstick_r5.py [case ...]       # -> tests/golden/bf16_yardstick_r5.json   (CPU, ~20 min)
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import full_dim_util as U          # noqa: E402

OUT = os.path.join(U.GOLDEN, "bf16_yardstick_r5.json")
KEYS = ("logits", "logits_max", "logits_lse", "loss", "loss_value", "hidden_worst", "grad_worst", "grad_norm_worst")


def seeded_inputs(c, vocab, seed):
    """Inputs of the loss-spread part: the case's shapes, another seed."""
    g = torch.Generator().manual_seed(seed)
    wave = (torch.randn(c["B"], c["N"], generator=g) * 0.1).clamp_(-1, 1)
    labels = torch.randint(4, vocab, (c["B"], c["L"]), generator=g)
    labels[:, -1] = 2
    text = torch.randint(4, vocab, (c["B"], c["L"] + 1), generator=g) if c["kind"] == "self" else None
    return wave, labels, text


def main():
    names = sys.argv[1:] or list(U.CASES)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        c = U.CASES[name]
        t0 = time.time()
        fx = U.load_fixture(name)
        ours = U.build_ours(c)
        wave, labels, text = U.case_inputs(c, ours.decoder_model.config.vocab_size)
        got, grads = U.oracle_run(c, ours, wave, labels, text, dt=torch.bfloat16)
        got["logits"] = got["logits"].float()
        got["hidden"] = {k: v.float() for k, v in got["hidden"].items()}
        r = U.compare(fx, got, {k: v.float() for k, v in grads.items()})
        res[name] = {k: float(r[k]) for k in KEYS if k in r}
        res[name].update(hidden_worst_name=r["hidden_worst_name"], grad_worst_name=r.get("grad_worst_name"),
                         grad_norm_worst_name=r.get("grad_norm_worst_name"), seconds=round(time.time() - t0, 1))
        print(name, res[name], flush=True)
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        del ours, got, grads
    for case in ("full_cfg4_2x2s", "full_cfg5_2x2s"):
        if sys.argv[1:] and "loss_spread" not in sys.argv[1:] and case not in sys.argv[1:]:
            continue
        c = U.CASES[case]
        ours = U.build_ours(c)
        spread = {}
        for seed in (101, 202, 303):
            wave, labels, text = seeded_inputs(c, ours.decoder_model.config.vocab_size, seed)
            l32 = U.oracle_loss(c, ours, wave, labels, text, torch.float32)
            l16 = U.oracle_loss(c, ours, wave, labels, text, torch.bfloat16)
            spread[str(seed)] = dict(loss_fp32=l32, loss_bf16=l16, err=abs(l16 - l32))
            print("loss spread", case, seed, spread[str(seed)], flush=True)
        res[case + "_loss_spread"] = spread
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
