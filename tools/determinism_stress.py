#!/usr/bin/env python3
"""Development aid: hunts run-to-run differences of the fused frame.

  1. the round-1 observation: clip loop on two streams (tests/test_clip_io.py) followed by two eager references, a
     hipGraph capture and replays (tests/test_pipeline_gpu.py, original order) -- every stage output compared bitwise;
  2. N eager repetitions of the same frame on one stream, on two alternating streams, and with two frames in flight,
     at a small size and at 1080p; reports the first stage whose bits differ from the first run.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd"), os.path.join(ROOT, "tests")]
from oracle import pipeline_cpu, synth  # noqa: E402
from vfi_amd.adacof.models import Model  # noqa: E402
from vfi_amd.fusion_net.fusion_net import FusionNet  # noqa: E402
from vfi_amd.fusion_net import interpolate_video as iv  # noqa: E402
from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator  # noqa: E402

dev = torch.device("cuda:0")
STAGES = ("ada_pred", "flow_var_map", "phase_pred", "phase_uncertainty", "ada_uncertainty", "base", "baseline", "final")


def models(seed):
    w = pipeline_cpu.seeded_weights(seed)
    adacof = Model(types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0))
    adacof.load(w["adacof"]); adacof.eval()
    fusion = FusionNet().to(dev); fusion.load_state_dict(w["fusionnet"]); fusion.eval()
    return adacof, fusion, w["phasenet"]


def diff(a, b):
    return [k for k in STAGES if k in a and not torch.equal(a[k], b[k])]


def snap(out):
    return {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}


def clip_io_pass():
    from PIL import Image
    tmp = tempfile.mkdtemp()
    src, dst = os.path.join(tmp, "in"), os.path.join(tmp, "out")
    os.makedirs(src)
    for i in range(4):
        f = synth.translating_pair(3, 64, 96, shift=(1.5 * i, -1.0 * i))[2]
        Image.fromarray((f.transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)).save(os.path.join(src, f"{i:03d}.png"))
    adacof, fusion, pn = models(0)
    runners = [FusionInterpolator(adacof, fusion, pn, dev) for _ in range(2)]
    args = types.SimpleNamespace(gpu_id=0, input_video=src, output_video=dst, index_from=0, zpad=3)
    iv.interpolate_video(args, runners=runners)


def graph_pass(tag):
    adacof, fusion, pn = models(3)
    run = FusionInterpolator(adacof, fusion, pn, dev)
    a0, _, a2 = (torch.from_numpy(x).to(dev) for x in synth.translating_pair(5, 64, 96))
    b0, _, b2 = (torch.from_numpy(x).to(dev) for x in synth.translating_pair(6, 64, 96))
    want_a = snap(run(a0, a2, output_baseline=True))
    want_b = snap(run(b0, b2, output_baseline=True))          # round-1 order: second eager reference BEFORE the capture
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=dev)
    f0, f2 = torch.empty_like(a0), torch.empty_like(a2)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        f0.copy_(a0); f2.copy_(a2)
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = run(f0, f2, output_baseline=True)
        g.replay(); s.synchronize()
        got_a = snap(out)
        f0.copy_(b0); f2.copy_(b2)
        g.replay(); s.synchronize()
        got_b = snap(out)
    torch.cuda.synchronize()
    later_b = snap(run(b0, b2, output_baseline=True))
    torch.cuda.synchronize()
    print(f"[{tag}] replay(a) vs eager(a): {diff(got_a, want_a)}  replay(b) vs early eager(b): {diff(got_b, want_b)}  "
          f"replay(b) vs later eager(b): {diff(got_b, later_b)}", flush=True)


def repeat(h, w, reps, mode):
    adacof, fusion, pn = models(2)
    runners = [FusionInterpolator(adacof, fusion, pn, dev) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    pairs = [tuple(torch.from_numpy(x).to(dev) for x in synth.translating_pair(10 + i, h, w)[::2]) for i in range(2)]
    torch.cuda.synchronize()
    first = [None, None]
    bad = {}
    for it in range(reps):
        outs = []
        for j in range(2):
            p = pairs[j]
            if mode == "one_stream":
                outs.append(snap(runners[0](p[0], p[1], output_baseline=True)))
            elif mode == "alternate":           # two streams, never concurrently
                with torch.cuda.stream(streams[j]):
                    outs.append(snap(runners[j](p[0], p[1], output_baseline=True)))
                streams[j].synchronize()
            else:                               # two frames in flight
                with torch.cuda.stream(streams[j]):
                    outs.append(snap(runners[j](p[0], p[1], output_baseline=True)))
        torch.cuda.synchronize()
        for j in range(2):
            if first[j] is None:
                first[j] = outs[j]
            else:
                d = diff(outs[j], first[j])
                if d:
                    bad.setdefault(d[0], []).append((it, j, d))
    print(f"repeat {h}x{w} x{reps} {mode}: " + (f"DIFFERENCES first-stage -> {bad}" if bad else "bit-identical"), flush=True)
    return first


if __name__ == "__main__":
    what = sys.argv[1:] or ["graph", "small", "big"]
    if "graph" in what:
        graph_pass("cold")
        clip_io_pass()
        for i in range(3):
            graph_pass(f"after clip loop {i}")
    if "small" in what:
        ref = {}
        for mode in ("one_stream", "alternate", "in_flight"):
            ref[mode] = repeat(64, 96, 40, mode)
        for mode in ("alternate", "in_flight"):
            print("small", mode, "vs one_stream:", [diff(ref[mode][j], ref["one_stream"][j]) for j in range(2)], flush=True)
    if "big" in what:
        ref = {}
        for mode in ("one_stream", "in_flight"):
            ref[mode] = repeat(1080, 1920, 6, mode)
        print("1080p in_flight vs one_stream:", [diff(ref["in_flight"][j], ref["one_stream"][j]) for j in range(2)], flush=True)
