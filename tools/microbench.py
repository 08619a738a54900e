#!/usr/bin/env python3
"""Per-op timing on one GPU (development aid, not the contract bench)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def conv_cases():
    # (name, n, cin, cout, h, w, ks, pad)
    H, W = 1088, 1920
    return [
        ("ke.conv1.0 6->32 @1", 1, 6, 32, H, W, 3, "zeros"),
        ("ke.conv1.2 32->32 @1", 1, 32, 32, H, W, 3, "zeros"),
        ("ke.conv2 64->64 @1/2", 1, 64, 64, H // 2, W // 2, 3, "zeros"),
        ("ke.heads 64->448 @1/2", 1, 64, 448, H // 2, W // 2, 3, "zeros"),
        ("ke.head3 64->25 @1/2", 1, 64, 25, H // 2, W // 2, 3, "zeros"),
        ("ke.head7 25->25 @1", 1, 25, 25, H, W, 3, "zeros"),
        ("ke.conv3 128->128 @1/4", 1, 128, 128, H // 4, W // 4, 3, "zeros"),
        ("ke.conv4 256->256 @1/8", 1, 256, 256, H // 8, W // 8, 3, "zeros"),
        ("ke.conv5 512->512 @1/16", 1, 512, 512, H // 16, W // 16, 3, "zeros"),
        ("ke.deconv5 512->512 @1/32", 1, 512, 512, H // 32, W // 32, 3, "zeros"),
        ("pn.l7a 88->64 @1080p", 3, 88, 64, 1080, 1920, 3, "reflect"),
        ("pn.l7b 64->64 @1080p", 3, 64, 64, 1080, 1920, 3, "reflect"),
        ("pn.pred 64->8 1x1", 3, 64, 8, 1080, 1920, 1, "zeros"),
        ("fn.enc0 18->32 5x5", 1, 18, 32, 1080, 1920, 5, "reflect"),
        ("fn.enc1 32->64 5x5 @1/2", 1, 32, 64, 540, 960, 5, "reflect"),
        ("fn.dec0 128->64 5x5 @1/4", 1, 128, 64, 270, 480, 5, "reflect"),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="conv")
    ap.add_argument("--filter", default="")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.what == "unet":
        import types
        from vfi_amd import _lib
        from vfi_amd.fusion_net.fusion_adacofnet import AdaCoFNet
        net = AdaCoFNet(types.SimpleNamespace(kernel_size=5, dilation=1, gpu_id=0)).to(dev)
        n = int(args.filter or 1)
        f0, f2 = torch.rand((n, 3, 1080, 1920), device=dev), torch.rand((n, 3, 1080, 1920), device=dev)
        for _ in range(2):
            net(f0, f2)
        torch.cuda.synchronize()
        _lib.PROFILE = _lib.Recorder()
        net(f0, f2)
        torch.cuda.synchronize()
        tot = 0.0
        for name, work, e0, e1 in _lib.PROFILE.rows:
            t = e0.elapsed_time(e1)
            tot += t
            if work and work[0] == "flop":
                print(f"{work[2]:34s} {t:7.3f} ms {work[1]/1e9:8.1f} GF {work[1]/t/1e9:7.1f} TF/s")
            else:
                print(f"{name:34s} {t:7.3f} ms")
        print("total", tot)
        _lib.PROFILE = None
    if args.what == "phasenet":
        import math, types
        from vfi_amd import _lib
        from vfi_amd.phase_net.phase_net import PhaseNet
        from vfi_amd.train.pyramid import Pyramid
        pyr = Pyramid(17, 4, math.sqrt(2), dev)
        net = PhaseNet(pyr, dev)
        img = torch.rand((6, 1080, 1920), device=dev)
        def run():
            vals, bufs = pyr.filter(img, concat_frames=2, phase_scale=1.0 / math.pi)
            return net(net.normalize_vals(vals, concat=bufs))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        _lib.PROFILE = _lib.Recorder()
        run()
        torch.cuda.synchronize()
        import collections
        agg = collections.OrderedDict()
        tot = 0.0
        for name, work, e0, e1 in _lib.PROFILE.rows:
            t = e0.elapsed_time(e1)
            tot += t
            key = work[2] if work else name
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1; a[1] += t; a[2] += work[1] if work and work[0] == "flop" else 0.0
        for k, (c, t, f) in agg.items():
            print(f"{k:34s} calls {c:3d} {t:8.3f} ms" + (f" {f/t/1e9:7.1f} TF/s" if f else ""))
        print("total", tot)
        # per-level conv times
        lv = [(n, w, e0.elapsed_time(e1)) for n, w, e0, e1 in _lib.PROFILE.rows if w and w[0] == "flop"]
        for i in range(0, len(lv), 3):
            print("block", i // 3, " ".join(f"{t:.3f}ms/{w[1]/t/1e9:.0f}TF" for _, w, t in lv[i:i + 3]))
        _lib.PROFILE = None
    if args.what == "adacof":
        bench_adacof(dev)
    if args.what == "median":
        import numpy as np
        h, w = 1080, 1920
        yy, xx = np.meshgrid(np.linspace(0, 6, h), np.linspace(0, 9, w), indexing="ij")
        smooth = torch.from_numpy((np.sin(yy) * np.cos(xx) + 0.1 * np.sin(7 * xx)).astype(np.float32)[None]).to(dev)
        noise = torch.randn((1, h, w), device=dev)
        for name, x in (("smooth", smooth), ("noise", noise)):
            t = timeit(lambda: ops.median_filter(x, 50), iters=5, warm=1)
            print(f"median50 1080p {name}: {t*1e3:.3f} ms", flush=True)
    if args.what == "conv":
        tot_f, tot_t = 0.0, 0.0
        for name, n, cin, cout, h, w, ks, pad in conv_cases():
            if args.filter and args.filter not in name:
                continue
            x = torch.randn((n, cin, h, w), device=dev)
            pc = ops.PackedConv(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5, torch.zeros(cout), device=dev)
            out = torch.empty((n, cout, h, w), device=dev)
            t = timeit(lambda: ops.conv2d(x, pc, pad, "relu", out=out), iters=args.iters)
            fl = 2.0 * n * cin * cout * ks * ks * h * w
            tot_f += fl; tot_t += t
            print(f"{name:28s} {t*1e3:8.3f} ms  {fl/t/1e12:7.2f} TFLOP/s", flush=True)
        print(f"{'total':28s} {tot_t*1e3:8.3f} ms  {tot_f/tot_t/1e12:7.2f} TFLOP/s")
        for name, n, cin, cout, hs, ws in [("ups head7 25->25 ->1088x1920", 1, 25, 25, 544, 960), ("ups occ 64->1", 1, 64, 1, 544, 960),
                                           ("ups up2 64->64 ->544x960", 1, 64, 64, 272, 480), ("ups up4 256->256 ->136x240", 1, 256, 256, 68, 120)]:
            x = torch.randn((n, cin, hs, ws), device=dev)
            pc = ops.PackedConv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.zeros(cout), device=dev)
            out = torch.empty((n, cout, 2 * hs, 2 * ws), device=dev)
            t = timeit(lambda: ops.conv2d(x, pc, "zeros", "relu", out=out, upsample2x=True), iters=args.iters)
            fl = 2.0 * n * cin * cout * 9 * 4 * hs * ws
            print(f"{name:28s} {t*1e3:8.3f} ms  {fl/t/1e12:7.2f} TFLOP/s", flush=True)


def bench_adacof(dev):
    from vfi_amd.adacof.cupy_module.adacof import adacof_fused
    n, h, w = 1, 1088, 1920
    g = torch.Generator(device="cpu").manual_seed(1)
    f0 = torch.rand((n, 3, h, w), generator=g).to(dev)
    f2 = torch.rand((n, 3, h, w), generator=g).to(dev)
    W = torch.softmax(torch.randn((2, n, 25, h, w), generator=g), 2).to(dev).contiguous()
    a = (torch.randn((4, n, 25, h, w), generator=g) * 2).clamp(-8, 8).to(dev)
    occ = torch.rand((n, 1, h, w), generator=g).to(dev)
    x0 = torch.cat((f0, f0[:, :1]), 1).permute(0, 2, 3, 1).contiguous()
    x2 = torch.cat((f2, f2[:, :1]), 1).permute(0, 2, 3, 1).contiguous()
    t = timeit(lambda: adacof_fused(x0, x2, W[0], a[0], a[1], W[1], a[2], a[3], occ, 1, True, True, rgbx=True))
    by = n * h * w * (150 * 4 + 4 + 24 + 36 + 4)
    print(f"adacof_fused_rgbx 1088x1920: {t*1e3:.3f} ms  {by/t/1e9:.0f} GB/s ({by/t/8e12*100:.1f}% of 8 TB/s)", flush=True)
    L = torch.randn((2, n, 25, h, w), generator=g).to(dev)
    t = timeit(lambda: adacof_fused(x0, x2, L[0], a[0], a[1], L[1], a[2], a[3], occ, 1, False, True, rgbx=True, weights_are_logits=True))
    by = n * h * w * (150 * 4 + 4 + 24 + 12 + 4)
    print(f"adacof_fused_rgbx logits, no sides (as the frame runs it): {t*1e3:.3f} ms  {by/t/1e9:.0f} GB/s ({by/t/8e12*100:.1f}% of 8 TB/s)", flush=True)
    for sides, mask in ((True, True), (False, False)):
        t = timeit(lambda: adacof_fused(f0, f2, W[0], a[0], a[1], W[1], a[2], a[3], occ, 1, sides, mask))
        by = n * h * w * (150 * 4 + 4 + 24 + 12 * (3 if sides else 1) + (4 if mask else 0))
        print(f"adacof_fused 1088x1920 sides={sides} mask={mask}: {t*1e3:.3f} ms  {by/t/1e9:.0f} GB/s ({by/t/8e12*100:.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
