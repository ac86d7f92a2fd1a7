"""BASELINE.json configs 4 and 5 at their FULL dimensions: hubert-large-ll60k -> mbart-large-50
(d 1024, 24 stable-LN layers, "layer"-norm CNN with conv bias, FFN 4096, V 250 054, down_scale 8) and SpeechMixSelf
wav2vec2-large -> t5-large (12 of 24 layers kept, T5 d 1024 / 24+24 layers / FFN 4096 / V 32 128, LM frozen, text pass).
The oracle does not run ~1 G parameters in seconds, so these are the size-independent properties of
tests/test_gpu_fullsize.py - bit-identical reruns, clip-permutation equivariance, clip independence, and the
data-parallel identity (batch gradient = mean of its half-batch gradients) - at 8 clips x 10 s; the values themselves are
pinned by the tiny twins of test_gpu_e2e.py / test_gpu_r2.py (same code paths: "layer" CNN, stable LN, mBART pre-LN, T5)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, SAMPLES, LABEL_LEN = 8, 160000, 32


def _setup(cfg):
    from tools.gpu_bench_cfg import build
    model = build(cfg).eval()
    V = model.decoder_model.config.vocab_size
    g = torch.Generator().manual_seed(1234)
    wave = (torch.randn(B, SAMPLES, generator=g) * 0.1).clamp_(-1, 1).cuda()
    labels = torch.randint(4, V, (B, LABEL_LEN), generator=g)
    labels[:, -1] = model.decoder_model.config.eos_token_id
    text = torch.randint(4, V, (B, 33), generator=g).cuda() if cfg == "5" else None
    return model, wave, labels.cuda(), text


def _forward(model, wave, labels, text):
    with torch.no_grad():
        out = model(wave, labels=labels, return_model_detail=True, **({"text_input_ids": text} if text is not None else {}))
    return out["loss"].float().clone(), out["raw_logits"].float().clone()


@pytest.mark.parametrize("cfg", ["4", "5"])
def test_large_configs_full_size_properties(cfg):
    from speechmix_amd.trainer import StepRunner
    model, wave, labels, text = _setup(cfg)
    sl = (lambda t, idx: t[idx] if t is not None else None)
    loss, logits = _forward(model, wave, labels, text)
    assert logits.shape[:2] == (B, LABEL_LEN) and torch.isfinite(logits).all() and torch.isfinite(loss)
    loss2, logits2 = _forward(model, wave, labels, text)
    assert torch.equal(logits, logits2)                                                # bit-identical reruns
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(7)).cuda()
    loss_p, logits_p = _forward(model, wave[perm], labels[perm], sl(text, perm))
    scale = logits.abs().max().item()
    assert (logits_p - logits[perm]).abs().max().item() <= 2e-3 * scale               # same arithmetic per clip
    assert abs(loss_p.item() - loss.item()) <= 1e-3 * abs(loss.item())
    # a clip's logits do not depend on its neighbours (different tile walks / kernel choices at 2 clips: summation order)
    _, logits_s = _forward(model, wave[:2], labels[:2], sl(text, slice(0, 2)))
    assert (logits_s - logits[:2]).abs().max().item() <= 3e-2 * scale
    # data-parallel identity: gradient of 8 clips = mean of the gradients of its halves
    runner = StepRunner(model, lr=0.0, optimizer="sgd", max_grad_norm=0.0)
    l_all = runner.step(wave, labels, text_input_ids=text).item()
    g_all = model.store.grad.clone()
    h = B // 2
    l_a = runner.step(wave[:h], labels[:h], text_input_ids=sl(text, slice(0, h))).item()
    g_half = model.store.grad.clone()
    l_b = runner.step(wave[h:], labels[h:], text_input_ids=sl(text, slice(h, B))).item()
    g_half += model.store.grad
    g_half *= 0.5
    assert abs(0.5 * (l_a + l_b) - l_all) <= 2e-3 * abs(l_all)
    assert torch.isfinite(g_all).all() and g_all.abs().max().item() > 0
    rel = ((g_all - g_half).norm() / g_all.norm()).item()
    cos = torch.nn.functional.cosine_similarity(g_all, g_half, dim=0).item()
    print(f"[config {cfg}] loss {l_all:.4f}, half-batch identity: rel {rel:.3e} cos {cos:.6f}")
    # bf16 storage makes every activation a step function of its fp32 value: a different fp32 summation order (another K
    # split at another batch size) flips a few roundings by one ulp in the first conv GEMM (478 of 2.4 M elements measured),
    # and every later GEMM sums ~1 500 such inputs, so the flips spread: 37 % of the CNN's outputs differ by one ulp between a
    # 1-clip and an 8-clip batch, 1.6 % of the hidden states' range after the encoder (measured round 2).  The fp32
    # path is bit-identical across batch sizes (tests/test_gpu_fullsize_parity.py).  Measured here: rel 6.1e-2 / cos 0.9984
    # (config 4), bounds = 3x.
    assert rel < 0.18 and cos > 0.985, (rel, cos)
    if cfg == "5":                                   # the frozen LM receives no gradient at all
        lm = [(o, n) for nm, (o, n, _) in model.store.offsets.items() if nm.startswith("decoder_model.")]
        assert all(float(g_all[o:o + n].abs().max()) == 0.0 for o, n in lm[:50])
