"""Training-mode dropout (TF:nn.functional.dropout sites of wav2vec2 / BART / T5) on the HIP path.

The reference draws its masks from torch's Philox stream, which no other implementation can reproduce bit for
bit; what CAN be pinned is (1) the mask statistics and determinism, (2) that every fused site (GEMM epilogue, norm
output, attention probabilities) applies exactly the mask the stand-alone kernel produces for the same (p, seed),
(3) attention forward/backward against a torch fp32 reference that is handed that mask, and (4) that the analytic
gradient of the whole step equals the directional derivative of the loss with the masks held fixed."""
import math

import numpy as np
import pytest
import torch

from tests.golden_util import load_case

pytestmark = pytest.mark.gpu


def _mask(shape, p, seed, dev):
    """keep-mask * 1/(1-p) of a site with flat row-major indexing, straight from the device kernel."""
    from speechmix_amd import ops
    ones = torch.ones(shape, dtype=torch.float32, device=dev)
    out = torch.empty_like(ones)
    ops.dropout(ones, out, ones.numel(), p, seed, ops.F32)
    return out


def _close(got, want, dtype, slack=1.0):
    """fp32: 1e-5 absolute + 1e-5 relative; bf16: one output rounding (2^-8 relative) plus the bf16 operands' spread."""
    rel, ab = (1e-5, 1e-5) if dtype == "fp32" else (8e-3 * slack, 4e-3 * slack)
    err = (got.float() - want.float()).abs()
    ok = err <= ab + rel * want.float().abs()
    return bool(ok.all()), err.max().item()


def test_mask_statistics_and_determinism():
    dev = torch.device("cuda:0")
    n = 1 << 22
    for p in (0.05, 0.1, 0.5):
        m1 = _mask((n,), p, 1234, dev)
        m2 = _mask((n,), p, 1234, dev)
        m3 = _mask((n,), p, 1235, dev)
        assert torch.equal(m1, m2)
        vals = torch.unique(m1).cpu().tolist()
        assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - p)) < 1e-6
        keep = (m1 > 0).float().mean().item()
        assert abs(keep - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4, (p, keep)
        # independent across seeds: agreement rate of two masks = keep^2 + p^2
        agree = ((m1 > 0) == (m3 > 0)).float().mean().item()
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 3e-3
        # no visible structure along rows of a [M, N] site: per-column keep rates stay within 5 sigma
        cols = (m1.view(-1, 1024) > 0).float().mean(0)
        assert (cols - (1 - p)).abs().max().item() < 5 * math.sqrt(p * (1 - p) / (n // 1024))
    assert torch.equal(_mask((64,), 0.0, 5, dev), torch.ones(64, device=dev))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fused_sites_apply_the_standalone_mask(dtype):
    from speechmix_amd import ops
    from speechmix_amd.ops import ACT_GELU
    dev = torch.device("cuda:0")
    dt = ops.F32 if dtype == "fp32" else ops.BF16
    tdt = ops.torch_dtype(dt)
    g = torch.Generator(device="cpu").manual_seed(0)
    M, N, K, p, seed = 200, 136, 72, 0.25, 99
    a = torch.randn(M, K, generator=g).to(dev, tdt)
    w = torch.randn(N, K, generator=g).to(dev, tdt) * 0.2
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev, tdt)
    mask = _mask((M, N), p, seed, dev)
    # GEMM epilogue: resid + drop(gelu(a w^T + b))
    plain = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(a, w, plain, M, N, K, dt, bias=bias, act=ACT_GELU, out_f32=True)
    got = torch.empty(M, N, dtype=tdt, device=dev)
    ops.gemm(a, w, got, M, N, K, dt, bias=bias, act=ACT_GELU, resid=resid, drop=(p, seed))
    want = plain * mask + resid.float()
    assert _close(got, want, dtype)[0], _close(got, want, dtype)
    dropped = (mask == 0)
    assert torch.equal(got[dropped].float(), resid[dropped].float())      # dropped entries are exactly the residual
    # backward-through-activation mode: dy W * gelu'(pre) * mask
    pre = torch.randn(M, N, generator=g).to(dev, tdt)
    dy = torch.randn(M, K, generator=g).to(dev, tdt)
    wt = torch.randn(K, N, generator=g).to(dev, tdt) * 0.2             # dgrad reads W [K(out), N(in)] rows-contiguous
    nodrop = torch.empty(M, N, dtype=tdt, device=dev)
    ops.gemm(dy, wt, nodrop, M, N, K, dt, b_rc=True, bv=ops.view(N), aux_in=pre, act=ACT_GELU)
    withdrop = torch.empty(M, N, dtype=tdt, device=dev)
    ops.gemm(dy, wt, withdrop, M, N, K, dt, b_rc=True, bv=ops.view(N), aux_in=pre, act=ACT_GELU, drop=(p, seed))
    assert _close(withdrop, nodrop.float() * mask, dtype)[0]
    # norm output
    D = 136
    x = torch.randn(M, D, generator=g).to(dev, tdt)
    gamma = torch.randn(D, generator=g).to(dev)
    beta = torch.randn(D, generator=g).to(dev)
    y0, y1 = torch.empty(M, D, dtype=tdt, device=dev), torch.empty(M, D, dtype=tdt, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.norm_fwd(x, y0, gamma, beta, mean, rstd, M, D, dt)
    ops.norm_fwd(x, y1, gamma, beta, mean, rstd, M, D, dt, drop=(p, seed))
    assert _close(y1, y0.float() * mask, dtype)[0]
    # norm backward with a dropped output == norm backward of the pre-masked gradient
    dyn = torch.randn(M, D, generator=g).to(dev, tdt)
    dym = (dyn.float() * mask).to(tdt)
    outs = []
    for dy_in, drop in ((dym, None), (dyn, (p, seed))):
        dx = torch.empty(M, D, dtype=tdt, device=dev)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        ops.norm_bwd(dy_in, x, dx, gamma, beta, mean, rstd, dg, db, M, D, dt, drop=drop)
        outs.append((dx.float(), dg, db))
    for u, v in zip(*outs):
        assert (u - v).abs().max().item() < (1e-4 if dtype == "fp32" else 3e-2) * max(1.0, v.abs().max().item())


@pytest.mark.parametrize("dtype,D,causal", [("fp32", 16, False), ("fp32", 64, True), ("bf16", 64, False), ("bf16", 64, True)])
def test_attention_probability_dropout_matches_torch_reference(dtype, D, causal):
    """softmax -> dropout -> @V, forward and backward (TF:models/wav2vec2/modeling_wav2vec2.py:540-548)."""
    from speechmix_amd import ops
    dev = torch.device("cuda:0")
    dt = ops.F32 if dtype == "fp32" else ops.BF16
    tdt = ops.torch_dtype(dt)
    B, H, T, p, seed = 2, 3, 77, 0.2, 4242
    d = H * D
    g = torch.Generator(device="cpu").manual_seed(1)
    qkv = (torch.randn(B * T, 3 * d, generator=g) * 0.7).to(dev, tdt)
    do = torch.randn(B * T, d, generator=g).to(dev, tdt)
    scale = D ** -0.5
    desc = ops.AttnDesc(B, H, T, T, D, causal, scale, drop=(p, seed))
    desc.set("Q", qkv, 0, T * 3 * d, 3 * d)
    desc.set("K", qkv, d, T * 3 * d, 3 * d)
    desc.set("V", qkv, 2 * d, T * 3 * d, 3 * d)
    o = torch.empty(B * T, d, dtype=tdt, device=dev)
    lse = torch.empty(B * H * T, device=dev)
    desc.set("O", o, 0, T * d, d)
    ops.attention_fwd(desc, lse, dt)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * H * T, device=dev)
    desc.set("dO", do, 0, T * d, d)
    desc.set("dQ", dqkv, 0, T * 3 * d, 3 * d)
    desc.set("dK", dqkv, d, T * 3 * d, 3 * d)
    desc.set("dV", dqkv, 2 * d, T * 3 * d, 3 * d)
    ops.attention_bwd(desc, lse, delta, dt)
    # torch fp32 reference with the very same mask
    Tp = (T + 3) // 4 * 4                       # mask rows are indexed with the key count rounded up to a multiple of 4
    mask = _mask((B, H, T, Tp), p, seed, dev)[..., :T]
    x = qkv.float().view(B, T, 3, H, D).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)   # [3,B,H,T,D]
    s = (x[0] @ x[1].transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.ones(T, T, device=dev).triu(1).bool(), float("-inf"))
    pr = torch.softmax(s, -1) * mask
    ref = (pr @ x[2]).permute(0, 2, 1, 3).reshape(B * T, d)
    ref.backward(do.float())
    dref = x.grad.permute(1, 3, 0, 2, 4).reshape(B * T, 3 * d)
    tol = 2e-5 if dtype == "fp32" else 4e-2
    assert (o.float() - ref.detach()).abs().max().item() < tol
    assert (dqkv.float() - dref).abs().max().item() < tol * (1 if dtype == "fp32" else 3)


def _train_model(case, **kw):
    from speechmix_amd.model import SpeechMixEED
    sd, inp, gold, m = load_case(case)
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], share_layer_ratio=m.get("share_layer_ratio", 0),
                         down_scale=m["down_scale"], compute_dtype="fp32", **kw)
    model.load_state_dict(sd, strict=False)
    model.train()
    return model, inp


def _loss(model, inp, seed=7):
    eng = model.engine
    eng.host_rng = type(eng.host_rng).seeded(seed) if hasattr(type(eng.host_rng), 'seeded') else eng.host_rng   # LayerDrop / SpecAugment draws
    eng.drop_rng = np.random.default_rng(seed)     # dropout site seeds
    return model(inp["input_values"], labels=inp["labels"])["loss"]


@pytest.mark.parametrize("case", ["eed_w2v2_bart", "eed_hubert_mbart"])
def test_training_mode_gradient_is_the_derivative_of_the_masked_forward(case):
    """With the RNG streams pinned the training-mode step is a deterministic function of the parameters; its
    analytic gradient (every site regenerating its mask in backward) must reproduce central differences along random
    directions confined to the encoder, the bridge and the LM in turn."""
    model, inp = _train_model(case)
    model._need_engine()
    ec = model.engine.ec
    assert ec.hidden_dropout > 0 and ec.attention_dropout > 0 and model.engine.lc.dropout > 0
    l_eval = None
    model.eval()
    with torch.no_grad():
        l_eval = model(inp["input_values"], labels=inp["labels"])["loss"].item()
    model.train()
    l1, l2 = _loss(model, inp).item(), _loss(model, inp).item()
    assert abs(l1 - l2) < 2e-6                           # same seeds -> same masks (atomic loss reduction: last-bit noise)
    assert abs(l1 - l_eval) > 1e-3                       # and dropout is really on
    assert abs(_loss(model, inp, seed=8).item() - l1) > 1e-4
    loss = _loss(model, inp)
    loss.backward()
    st = model.store
    grad = st.grad.clone()
    g = torch.Generator(device="cpu").manual_seed(3)
    checked = 0
    for prefix in ("encoder_model.", "length_adapters.|enc_to_dec_proj.", "decoder_model."):
        sel = torch.zeros_like(st.master, dtype=torch.bool)
        for name, (o, n, _) in st.offsets.items():
            if any(name.startswith(px) for px in prefix.split("|")) and st.requires_grad(name):
                sel[o:o + n] = True
        assert sel.any()
        gs = torch.where(sel, grad, torch.zeros_like(grad))
        # (a) isotropic random direction: catches components the analytic gradient leaves out (small signal: loose)
        # (b) randomly re-weighted gradient direction (large signal): tight
        ra = torch.where(sel, (torch.randn(st.master.numel(), generator=g) * 2e-3).to(sel.device), torch.zeros_like(grad))
        u = (0.5 + torch.rand(st.master.numel(), generator=g)).to(sel.device)
        gn2 = (gs.double() ** 2).sum().item()
        rb = gs * u * min(1e-2 / gn2, 0.05 / math.sqrt(gn2))       # <= 1e-2 loss change, <= 0.05 step norm (stay linear)
        for tag, r, rtol, atol in (("random", ra, 5e-2, 3e-6), ("grad-aligned", rb, 2e-2, 0.0)):
            analytic = (grad.double() * r.double()).sum().item()

            def central(scale):
                with torch.no_grad():
                    st.master.add_(r, alpha=scale)
                    lp = _loss(model, inp).item()
                    st.master.sub_(r, alpha=2 * scale)
                    lm = _loss(model, inp).item()
                    st.master.add_(r, alpha=scale)
                return (lp - lm) / (2 * scale)
            fd = (4 * central(0.5) - central(1.0)) / 3       # Richardson step: cancels the cubic term of the big step (the random
            # direction needs it too: its step has norm ~2 over the LM's parameters, and which masks / dropped layers the
            # pinned streams yield - HF's own draw order since round 3 - decides how curved the loss is along it)
            print(f"[{case}] {prefix} {tag}: analytic {analytic:.6e} central-diff {fd:.6e}")
            assert abs(analytic - fd) < rtol * max(abs(fd), abs(analytic)) + atol, (prefix, tag, analytic, fd)
            checked += 1
    assert checked == 6
