#!/usr/bin/env python3
"""Development aid: repeat the same 3x3 convolutions and report bitwise run-to-run differences."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (n, cin, cout, h, w, pad) in [(3, 64, 64, 64, 96, "reflect"), (1, 6, 32, 64, 96, "zeros"), (3, 88, 64, 23, 34, "reflect"),
                                  (1, 512, 512, 8, 12, "zeros"), (2, 64, 64, 270, 480, "zeros"), (12, 64, 64, 6, 9, "reflect")]:
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pc = ops.PackedConv(wt, b)
    x = torch.randn(n, cin, h, w, device=dev)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect" if pad == "reflect" else "constant"), wt, b)
    first = None
    bad = 0
    for it in range(30):
        if it % 2:   # interleave another input through the same buffers
            ops.conv2d(torch.randn_like(x), pc, pad_mode=pad, act="relu")
        y = ops.conv2d(x, pc, pad_mode=pad, act=None)
        torch.cuda.synchronize()
        if first is None:
            first = y.clone()
        elif not torch.equal(y, first):
            bad += 1
    print((n, cin, cout, h, w, pad), "max err vs torch", (first - ref).abs().max().item(), "nondeterministic runs:", bad, flush=True)
