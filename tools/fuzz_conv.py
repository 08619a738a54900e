#!/usr/bin/env python3
"""Development aid: random-shape 3x3 convolutions (Winograd path) against torch on the CPU."""
import os, sys, random
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import ops
dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    n, cin, cout = rnd.randint(1, 3), rnd.randint(1, 70), rnd.randint(1, 70)
    h, w = rnd.randint(2, 40), rnd.choice([rnd.randint(2, 40), rnd.randint(60, 140), 32, 64, 96, 128])
    pad = rnd.choice(["zeros", "reflect"])
    act = rnd.choice([None, "relu", "elu", "tanh", "sigmoid"])
    use_res = rnd.random() < 0.3
    g = torch.Generator().manual_seed(it)
    x = torch.randn((n, cin, h, w), generator=g)
    wgt = torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    res = torch.randn((n, cout, h, w), generator=g) if use_res else None
    xp = F.pad(x.double(), (1, 1, 1, 1), mode="reflect") if pad == "reflect" else F.pad(x.double(), (1, 1, 1, 1))
    ref = F.conv2d(xp, wgt.double(), b.double())
    ref = {None: lambda t: t, "relu": F.relu, "elu": F.elu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act](ref)
    if res is not None:
        ref = ref + res.double()
    pc = ops.PackedConv(wgt, b, device=dev)
    out = ops.conv2d(x.to(dev), pc, pad, act, residual=None if res is None else res.to(dev))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    worst = max(worst, err)
    if err > 1e-4 or not torch.isfinite(out).all():
        print("FAIL", (n, cin, cout, h, w, pad, act, use_res), err)
print("cases done, worst abs error", worst)
