#!/usr/bin/env python3
"""Development aid: per-shape time of every convolution of one 1080p frame (HIP events around each launch)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
import bench  # noqa: E402
from vfi_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
runners, _ = bench.build_runner(dev, 1)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
f0, f2 = torch.rand(3, h, w, device=dev), torch.rand(3, h, w, device=dev)
shapes = []
orig = _lib.call


def call(name, *args, work=None):
    if name.startswith("vfi_conv2d") and work is not None:
        # (x, xs, w, b, r, rs, y, ys, n, cin, h, w, cout, ks, ...)
        n, cin, hh, ww, cout, ks = args[8:14]
        work = (work[0], work[1], f"{work[2].split('<')[0]} ks{ks} N{n} {cin}->{cout} @{hh}x{ww}")
    if name == "vfi_resize_bilinear":
        n, c, hi, wi, ho, wo = args[6:12]
        work = ("byte", 4.0 * n * c * (hi * wi + ho * wo * (2 if args[2] else 1)), f"resize N{n} C{c} {hi}x{wi}->{ho}x{wo}")
    orig(name, *args, work=work)


_lib.call = call
ops._lib.call = call
for _ in range(2):
    runners[0](f0, f2, output_baseline=True)
torch.cuda.synchronize()
_lib.PROFILE = _lib.Recorder()
runners[0](f0, f2, output_baseline=True)
agg = _lib.PROFILE.summary()
_lib.PROFILE = None
tot = sum(v["seconds"] for v in agg.values())
print(f"total {tot * 1e3:.2f} ms over {sum(v['calls'] for v in agg.values())} calls")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["seconds"])[:400]:
    rate = v["work"] / v["seconds"] / 1e12 if v["kind"] == "flop" else v["work"] / v["seconds"] / 1e9
    print(f"{v['seconds'] * 1e3:8.3f} ms  x{v['calls']:3d}  {rate:8.1f} {'TF/s' if v['kind'] == 'flop' else 'GB/s'}  {k}")
