#!/usr/bin/env python3
"""Per-tensor statistics of the reference's TRAINED checkpoints (build container only: reads /root/reference; the
checkpoints themselves are never committed).  -> tests/golden/trained_weight_stats.json

The GPU parity tests draw seeded random tensors and rescale them tensor by tensor to these statistics
(tests/trained_stats.py), so that the kernels also run on weights / folded BatchNorm scales of the magnitude a trained
network has (running variances of 0.02-0.1, i.e. a x3-x7 gain in front of every ELU) and not only on default-init ones."""
import json
import os
import sys

import torch

REF = "/root/reference/src"
FILES = {"phasenet": "phase_net/phase_net.pt", "fusionnet": "fusion_net/fusion_net.pt"}


def stats(t):
    t = t.double().flatten()
    d = {"shape": None, "mean": float(t.mean()), "std": float(t.std()) if t.numel() > 1 else 0.0, "min": float(t.min()), "max": float(t.max())}
    return d


def main():
    out = {}
    for name, rel in FILES.items():
        sd = torch.load(os.path.join(REF, rel), map_location="cpu")
        out[name] = {}
        for k, v in sd.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            d = stats(v)
            d["shape"] = list(v.shape)
            out[name][k] = d
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trained_weight_stats.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
