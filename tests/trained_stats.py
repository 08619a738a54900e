"""Seeded random state dicts with the per-tensor statistics of the reference's trained checkpoints
(tests/golden/trained_weight_stats.json, made by tests/golden/make_trained_stats.py in the build container)."""
import json
import os

import torch

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_weight_stats.json")


def state_dict_like_trained(name, template, seed=0):
    """`template`: a state dict with the right keys / shapes (the oracle's seeded weights).  Every floating-point tensor is
    replaced by seeded noise with the trained tensor's mean and standard deviation, clipped to its [min, max]
    (running variances therefore stay positive); integer entries (num_batches_tracked) are kept."""
    with open(_PATH) as f:
        table = json.load(f)[name]
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in template.items():
        st = table.get(k)
        if st is None or not v.dtype.is_floating_point:
            out[k] = v.clone()
            continue
        assert list(v.shape) == st["shape"], (k, tuple(v.shape), st["shape"])
        t = torch.randn(v.shape, generator=g, dtype=torch.float64) * st["std"] + st["mean"]
        out[k] = t.clamp_(st["min"], st["max"]).to(v.dtype)
    return out
