"""Helpers to load the committed golden fixtures (tests/golden/*.npz + manifest.json)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    out = {}
    for name in ("manifest.json", "manifest_r2.json", "manifest_r3.json", "manifest_r4.json"):      # later rounds' fixtures live beside round 1's
        path = os.path.join(GOLDEN, name)
        if os.path.exists(path):
            with open(path) as f:
                out.update(json.load(f))
    return out


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd, out, inp = {}, {}, {}
    for k in z.files:
        t = torch.from_numpy(z[k])
        if k.startswith("w::"):
            sd[k[3:]] = t
        elif k.startswith("o::"):
            out[k[3:]] = t
        else:
            inp[k] = t
    return sd, inp, out, manifest().get(name, {})
