#!/usr/bin/env python3
"""Development aid: run one fused frame with every float32 device allocation wrapped in NaN guard bands (and NaN-filled
itself).  Reports (a) guard bands that were written (out-of-bounds stores), (b) NaNs in the outputs (values that depend
on out-of-bounds or uninitialised reads)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd"), os.path.join(ROOT, "tests")]
from oracle import pipeline_cpu, synth  # noqa: E402
from test_pipeline_gpu import _models  # noqa: E402

GUARD = 32 * 1024          # floats on each side
records = []
_empty = torch.empty


def guarded_empty(*size, **kw):
    shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
    dev = torch.device(kw.get("device", "cpu"))
    if dev.type != "cuda" or kw.get("dtype", torch.float32) != torch.float32 or kw.get("pin_memory"):
        return _empty(*size, **kw)
    n = 1
    for d in shape:
        n *= d
    if n > 64 * 1024 * 1024:      # (the 192 MiB split-K workspace)
        return _empty(*size, **kw)
    base = torch.full((n + 2 * GUARD,), float("nan"), dtype=torch.float32, device=dev)
    records.append((base, n, tuple(shape)))
    return base[GUARD:GUARD + n].view(*shape)


torch.empty = guarded_empty
torch.empty_like = lambda t, **kw: guarded_empty(tuple(t.shape), dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device))

device = torch.device("cuda:0")
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 96)
weights = pipeline_cpu.seeded_weights(3)
run = _models(device, weights)
f0, _, f2 = (torch.from_numpy(x).to(device) for x in synth.translating_pair(5, h, w))
res = run(f0, f2, output_baseline=True)
torch.cuda.synchronize()
for k, v in res.items():
    if torch.is_tensor(v):
        print(f"{k}: NaNs {int(torch.isnan(v).sum())} of {v.numel()}")
bad = 0
for base, n, shape in records:
    lo, hi = base[:GUARD], base[GUARD + n:]
    wl, wh = int((~torch.isnan(lo)).sum()), int((~torch.isnan(hi)).sum())
    if wl or wh:
        bad += 1
        idx_l = (~torch.isnan(lo)).nonzero().flatten()
        idx_h = (~torch.isnan(hi)).nonzero().flatten()
        print(f"OOB write around tensor {shape}: {wl} floats before (nearest {GUARD - int(idx_l.max()) if wl else '-'} back), "
              f"{wh} after (first at +{int(idx_h.min()) if wh else '-'}, last at +{int(idx_h.max()) if wh else '-'})")
print(f"{len(records)} guarded allocations, {bad} with written guard bands")
