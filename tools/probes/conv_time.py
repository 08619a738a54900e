#!/usr/bin/env python3
"""Development aid: time 3x3 layers given as n,cin,cout,h,w[,act[,res]] arguments (res: with a residual input).  The device's clocks move with its power state,
so the layers are timed round-robin (ROUNDS rounds of LAUNCHES launches each, HIP events) and the table gives the minimum and
the median round of each: compare columns of one run, not numbers of different runs."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import ops  # noqa: E402

ROUNDS, LAUNCHES = int(os.environ.get("ROUNDS", 7)), int(os.environ.get("LAUNCHES", 30))
dev = torch.device("cuda:0")
cases = []
for spec in sys.argv[1:]:
    f = spec.split(",")
    n, cin, cout, h, w = (int(v) for v in f[:5])
    act = f[5] if len(f) > 5 else "elu"
    act = None if act == "none" else act
    res = torch.randn((n, cout, h, w), device=dev) if len(f) > 6 and f[6] == "res" else None
    if len(f) > 6 and f[6] == "pool":
        res = "pool"
    x = torch.randn((n, cin, h, w), device=dev)
    pc = ops.PackedConv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.zeros(cout), device=dev)
    out = torch.empty((n, cout, h, w), device=dev)
    cases.append((spec, n, cin, cout, h, w, act, x, pc, out, [], res))
for rnd in range(ROUNDS + 1):
    for spec, n, cin, cout, h, w, act, x, pc, out, times, res in cases:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(LAUNCHES):
            if res == "pool":
                ops.conv2d_pool2(x, pc, False, "zeros", act)
            else:
                ops.conv2d(x, pc, "reflect", act, out=out, residual=res)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times.append(e0.elapsed_time(e1) / LAUNCHES)
for spec, n, cin, cout, h, w, act, x, pc, out, times, res in cases:
    gf = 2.0 * n * cin * cout * 36 * (h * w / 16) / 1e9
    lo, med = min(times), statistics.median(times)
    print(f"N{n} {cin}->{cout} @{h}x{w} {act}{' +pool' if res == 'pool' else ' +res' if res is not None else ''}: min {lo:.3f} ms ({gf / lo / 157.3:.3f})  median {med:.3f} ms ({gf / med / 157.3:.3f} of 157.3 TFLOP/s, F(4x4) count)")
