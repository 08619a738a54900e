#!/usr/bin/env python3
"""Development aid: ONE 3x3 layer, a few launches -- the program to put behind `rocprofv3 --pmc ...` (profiles/r02_conv_pmc.md).
usage: conv_one_layer.py [n cin cout h w [pad]]   (default: PhaseNet's 64 -> 64 at 1080p x 3)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import ops  # noqa: E402

a = sys.argv[1:]
n, cin, cout, h, w = (int(v) for v in a[:5]) if len(a) >= 5 else (3, 64, 64, 1080, 1920)
pad = a[5] if len(a) > 5 else "reflect"
dev = torch.device("cuda:0")
x = torch.randn((n, cin, h, w), device=dev)
pc = ops.PackedConv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.zeros(cout), device=dev)
out = torch.empty((n, cout, h, w), device=dev)
for _ in range(6):
    ops.conv2d(x, pc, pad, "relu", out=out)
torch.cuda.synchronize()
