"""Debug aid (round 5): does any allocation made while a step is captured overlap a speech-encoder hidden state that is still alive?"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd import graphs, engine as E
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
ENC = dict(model_type="wav2vec2", hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256,
           conv_dim=[64] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2], num_conv_pos_embeddings=16,
           num_conv_pos_embedding_groups=4, layerdrop=0.3)
LM = dict(model_type="bart", vocab_size=200, d_model=128, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
          decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256, max_position_embeddings=128)
graphs.MODE, graphs.ENABLED = "1", True
g = torch.Generator().manual_seed(0)
wave = (torch.randn(4, 12000, generator=g) * 0.1).cuda()
labels = torch.randint(4, 200, (4, 6), generator=g).cuda()
# a first model, as the test has one (its garbage is what the collector finds)
for rep in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        model = SpeechMixEED(ENC, LM, down_scale=2, compute_dtype="bf16", init_seed=0, weighted_sum=True).eval()
    r = StepRunner(model, lr=0.0, optimizer="sgd", max_grad_norm=0.0, seed=9)
    r.use_graphs = rep == 1
    eng = model.engine
    log = []
    orig_new = E.Engine.new

    def new(self, *shape, dt=None, _o=orig_new):
        t = _o(self, *shape, dt=dt)
        if self._cap is not None:
            log.append((t.data_ptr(), t.numel() * t.element_size(), tuple(shape), torch.cuda.current_stream().cuda_stream))
        return t
    E.Engine.new = new
    import speechmix_amd.ops as O
    orig_wb = O.weighted_sum_bwd
    seen = {}

    def wb(hidden, w, dy, dots, dw, sw, n, dtype, _o=orig_wb):
        if eng._cap is not None:
            seen["hidden"] = [(h.data_ptr(), h.numel() * h.element_size()) for h in hidden]
            seen["dy"] = dy.data_ptr(); seen["dots"] = dots.data_ptr(); seen["sw"] = sw.data_ptr()
            seen["t"] = list(hidden) + [dy, dots, sw]
        return _o(hidden, w, dy, dots, dw, sw, n, dtype)
    O.weighted_sum_bwd = wb
    for step in range(5):
        loss = r.step(wave, labels)
        torch.cuda.synchronize()
        o, k, _ = model.store.offsets["weights_sum"]
        print(rep, step, "graphed", r._graphs is not None, "loss", float(loss), "dW(weights_sum)", model.store.grad[o:o + k].tolist(), flush=True)
    E.Engine.new = orig_new
    if rep == 1 and r._graphs is not None:
        c = r._graphs.carry
        hid = {k: v for k, v in c.items() if k.startswith("fwd_out") or k.startswith("fwd_in")}
        for name, t in sorted(hid.items()):
            a0, a1 = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
            hits = [(i, p, n, sh) for i, (p, n, sh, st) in enumerate(log) if p < a1 and p + n > a0 and p != a0]
            print(name, hex(a0), t.numel() * t.element_size(), "overlapping later allocations:", hits[:6])
        print(len(log), "allocations through Engine.new during the capture")
        print("hidden ptrs at capture:", [hex(p) for p, _ in seen["hidden"]])
        for l, (p, n) in enumerate(seen["hidden"]):
            idx = [i for i, (q, m, sh, st) in enumerate(log) if q == p]
            later = [(i, sh, hex(st)) for i, (q, m, sh, st) in enumerate(log) if q < p + n and q + m > p and i > (idx[0] if idx else -1)]
            print("hidden", l, hex(p), "allocated as log entry", idx, "later overlapping allocations:", later[:5])
        for l, h in enumerate(seen["t"][:5]):
            print("hidden", l, "max abs now", float(h.float().abs().max()))
