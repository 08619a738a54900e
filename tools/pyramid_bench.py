#!/usr/bin/env python3
"""Pyramid-only timing at 1080p: analysis of 6 images (all levels, both residuals) and synthesis of 3, as the fused frame
issues them.  Prints ms and the algorithmic HBM rate (SURVEY 8d byte counts).  Under
`rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/pyramid_bench.py` the per-kernel table follows."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd.train.pyramid import Pyramid  # noqa: E402
from vfi_amd.values import DecompValues  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
ITERS = int(os.environ.get("ITERS", 10))
height = int(math.ceil((math.log2(min(H, W)) - 3) * 2) + 2)
dev = torch.device("cuda:0")
pyr = Pyramid(height, 4, math.sqrt(2), dev)
img = torch.rand((6, H, W), device=dev)
v = pyr.filter(img)
sub = DecompValues(v.high_level[:3], [p[:12] for p in v.phase], [a[:12] for a in v.amplitude], v.low_level[:3])
rec = pyr.inv_filter(sub)
torch.cuda.synchronize()
print("round trip max err", float((rec - img[:3]).abs().max()))
plan = pyr.pyr.plan(H, W, 6)
for name, fn, n in (("analysis N=6", lambda: pyr.filter(img), 6), ("synthesis N=3", lambda: pyr.inv_filter(sub), 3)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ITERS):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / ITERS * 1e3
    byt = plan._bytes(n, (1 << (height - 2)) - 1, True, True)
    print(f"{name}: {ms:.3f} ms   algorithmic {byt / 1e6:.0f} MB -> {byt / ms / 1e6:.0f} GB/s = {byt / ms / 1e6 / 80:.1f} % of 8 TB/s", flush=True)
