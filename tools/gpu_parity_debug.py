"""Worst gradient tensors of a full-dimension config (fp32 path) vs the oracle.  python tools/gpu_parity_debug.py CFG"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.gpu_fullsize_cfg_parity as P
cfg = sys.argv[1]
model = P.build(cfg, "fp32")
ec, lc = model.encoder_model.config.to_dict(), model.decoder_model.config.to_dict()
kind = P.CFGS[cfg][0]
wave, labels, text = P.inputs(2, 32000, 8, lc["vocab_size"], kind == "self")
trainable = {k for k, p in model.named_parameters() if p.requires_grad}
sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
leaves, r, secs = P.oracle_run(cfg, sd, ec, lc, wave, labels, text, model.num_speech_encoder_layers, trainable=trainable)
kw = {"text_input_ids": text} if text is not None else {}
out = model(wave, labels=labels, return_model_detail=True, **kw)
out["loss"].backward()
torch.cuda.synchronize()
named = dict(model.named_parameters())
rows = []
gmax = max(float(v.grad.abs().max()) for v in leaves.values() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None)
for k, v in leaves.items():
    if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None and k in named and named[k].grad is not None:
        g = v.grad.float(); got = named[k].grad.detach().float().cpu()
        d = (got - g).abs()
        rows.append((d.max().item() / max(g.abs().max().item(), 1e-3 * gmax), k, d.max().item(), g.abs().max().item(), tuple(g.shape),
                     int(d.argmax())))
rows.sort(reverse=True)
print("gmax", gmax)
for e, k, dm, gm, shp, am in rows[:25]:
    print(f"{e:.3e}  {k}  abs {dm:.3e}  own max {gm:.3e}  shape {shp}  argmax idx {am}")
