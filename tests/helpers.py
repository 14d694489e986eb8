"""Shared test helpers: golden-fixture loading and error metrics."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    """Returns {key: tensor}; nested dicts saved as 'group/key' come back as {'group': {key: tensor}}."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        t = torch.from_numpy(np.array(z[k]))
        if "/" in k:
            g, kk = k.split("/", 1)
            out.setdefault(g, {})[kk] = t
        else:
            out[k] = t
    return out


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name + ".json")))


def prefixed(sd, prefix="m"):
    return {prefix + "." + k: v for k, v in sd.items()}


def rel_err(a, b):
    """max-norm relative error  max|a-b| / max|b|."""
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
