"""Golden vectors for the Adafactor restatement: HF's own class (transformers.optimization.Adafactor, configured as the
reference's Trainer configures it for optim="adafactor": scale_parameter=False, relative_step=False, lr given) stepped
three times on tensors of the shapes the model has (matrix, 3-D conv weight, vector, 1 x n row), gradients seeded.
    python tests/golden/make_adafactor_golden.py   ->  tests/golden/adafactor.npz  (data only)"""
import os
import numpy as np
import torch
from transformers.optimization import Adafactor

torch.manual_seed(7)
shapes = {"matrix": (24, 40), "conv3d": (6, 5, 3), "vector": (33,), "rowvec": (1, 50), "posconv": (4, 3, 16)}
out = {}
params = {k: torch.nn.Parameter(torch.randn(*s) * 0.5) for k, s in shapes.items()}
opt = Adafactor(list(params.values()), lr=5e-4, scale_parameter=False, relative_step=False, warmup_init=False)
for k, p in params.items():
    out[f"p0::{k}"] = p.detach().numpy().copy()
for step in range(3):
    for k, p in params.items():
        g = torch.randn_like(p) * (10.0 if (k == "vector" and step == 1) else 0.1)   # one large step exercises the RMS clip
        p.grad = g
        out[f"g{step}::{k}"] = g.numpy().copy()
    opt.step()
    for k, p in params.items():
        out[f"p{step + 1}::{k}"] = p.detach().numpy().copy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "adafactor.npz"), **out)
print("wrote", len(out), "arrays")
