"""Round-6 host logic on the CPU (no GPU, no compute through the library): the flat store's version counters and offset lookup, the
Adafactor plan's tensor -> tile map that the phased optimizer step ranges over, the front-end range the runner hands to it, and the
engine-side bookkeeping of LayerDrop under gradient accumulation."""
import contextlib
import io

import torch

ENC = dict(model_type="wav2vec2", hidden_size=64, num_hidden_layers=3, num_attention_heads=2, intermediate_size=128,
           conv_dim=[32] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2], num_conv_pos_embeddings=16,
           num_conv_pos_embedding_groups=4)
LM = dict(model_type="bart", vocab_size=120, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
          decoder_attention_heads=2, encoder_ffn_dim=128, decoder_ffn_dim=128, max_position_embeddings=128)


def _model():
    from speechmix_amd.model import SpeechMixEED
    with contextlib.redirect_stdout(io.StringIO()):
        return SpeechMixEED(ENC, LM, down_scale=2, compute_dtype="fp32", init_seed=0)


def test_flat_store_versions_and_offset_lookup():
    from speechmix_amd.params import FlatStore
    m = torch.nn.Module()
    m.a = torch.nn.Parameter(torch.zeros(10, 7))
    m.b = torch.nn.Parameter(torch.zeros(130))
    m.c = torch.nn.Parameter(torch.zeros(3, 3), requires_grad=False)
    st = FlatStore(m, "cpu", torch.float32)
    oa, ob, oc = (st.offsets[n][0] for n in ("a", "b", "c"))
    assert st.name_at(oa) == "a" and st.name_at(oa + 69) == "a" and st.name_at(ob + 129) == "b" and st.name_at(oc + 8) == "c"
    assert st.name_at(oa + 70) is None                       # alignment padding between a and b
    assert st.name_at(st.total + 5) is None
    w0, h0 = st.wver, st.hard_ver
    st.mark_shadow_fresh()                                    # an optimizer step wrote masters and compute copies together
    assert (st.wver, st.hard_ver) == (w0 + 1, h0)
    assert st.refresh_shadow() is False                      # fp32 store: the compute copies ARE the masters - nothing to re-cast
    assert st.requires_grad("a") and not st.requires_grad("c")


def test_adafactor_plan_maps_tensors_to_contiguous_tile_ranges():
    from speechmix_amd.ops import AdafactorPlan
    shapes = [(64, 10), (64,), (128, 64, 3), (128,), (300, 520), (520,), (9000,), (2048, 256)]
    offs, off = [], 0
    for sh in shapes:
        n = 1
        for d in sh:
            n *= d
        offs.append((off, sh))
        off = (off + n + 63) // 64 * 64
    plan = AdafactorPlan(offs, torch.device("cpu"))
    t0 = [plan.tile0_of(i) for i in range(len(shapes) + 1)]
    assert t0[0] == 0 and t0[-1] == plan.ntiles and all(a < b for a, b in zip(t0[:-1], t0[1:]))
    tiles = plan.tiles.numpy()
    for i in range(len(shapes)):
        assert (tiles[t0[i]:t0[i + 1], 0] == i).all()         # a tensor's tiles are contiguous: a tile range is a set of whole tensors
    assert t0[7] - t0[6] == 2                                 # a 9 000-element vector: two 8 192-element tiles


def test_front_end_tensors_are_one_range_of_the_optimizer_order():
    """What StepRunner hands to AdafactorPlan.step(split=...): the speech encoder's tensors outside its layers sit side by side in the flat order
    (the optimizer's tail may run beside the next step's front end only if that range can be updated first)."""
    from speechmix_amd.params import FlatStore
    st = FlatStore(_model(), "cpu", torch.float32)           # (the model's own store is built with the engine, on the GPU)
    names = [nm for nm, _ in sorted(st.offsets.items(), key=lambda kv: kv[1][0])]
    ep = "encoder_model."
    assert any(n.startswith(ep + "encoder.layers.") for n in names)
    front = [n.startswith(ep) and not n.startswith(ep + "encoder.layers.") for n in names]
    idx = [i for i, f in enumerate(front) if f]
    assert idx and len(idx) == idx[-1] + 1 - idx[0] and len(idx) < len(names)
    used_by_front = ("masked_spec_embed", "feature_extractor.", "feature_projection.", "encoder.pos_conv_embed.", "encoder.layer_norm.")
    assert all(names[i][len(ep):].startswith(used_by_front) for i in idx)


def test_dropped_layer_bookkeeping_under_gradient_accumulation():
    import types
    from speechmix_amd.engine import Engine
    eng = types.SimpleNamespace(last_dropped=[], dropped_since_zero=set(), param_event=None)
    eng.note_dropped = types.MethodType(Engine.note_dropped, eng)
    eng.wait_params = types.MethodType(Engine.wait_params, eng)
    eng.last_dropped = [0, 2]
    eng.note_dropped(True)                                    # first micro-batch after zero_grad
    assert eng.dropped_since_zero == {0, 2}
    eng.last_dropped = [2]
    eng.note_dropped(False)                                   # accumulated: layer 0 now holds a gradient
    assert eng.dropped_since_zero == {2}
    eng.last_dropped = [0, 1]
    eng.note_dropped(False)
    assert eng.dropped_since_zero == set()
    eng.last_dropped = [1]
    eng.note_dropped(True)                                    # the next update starts over
    assert eng.dropped_since_zero == {1}
    eng.param_event = None
    eng.wait_params()                                         # nothing pending: a no-op on any device
