import math, os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from oracle import pyramid_cpu, synth, layout_cpu
from vfi_amd.train.pyramid import Pyramid
dev = torch.device("cuda:0")
SIZES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(128, 160), (96, 96), (64, 96)]
for (h, w) in SIZES:
    f0, _, f2 = synth.translating_pair(7, h, w)
    img = torch.from_numpy(np.concatenate([f0, f2], 0))
    height = layout_cpu.calc_pyr_height(h, w)
    ref = pyramid_cpu.Pyramid(height).filter(img)
    got = Pyramid(height, 4, math.sqrt(2), dev).filter(img.to(dev))
    for k in range(len(ref.phase)):
        a_ref, p_ref = ref.amplitude[k], ref.phase[k]
        a, p = got.amplitude[k].cpu(), got.phase[k].cpu()
        z_ref, z = torch.polar(a_ref, p_ref), torch.polar(a, p)
        scale = max(1e-3, a_ref.max().item())
        flips = int(((p - p_ref).abs() > 3.0).sum())
        print(f"{h}x{w} level {k} shape {tuple(p.shape)}: coeff err {float((z - z_ref).abs().max()) / scale:.2e}  amp err {float((a - a_ref).abs().max()) / scale:.2e}  phase flips {flips}")
    print("low", float((got.low_level.cpu() - ref.low_level).abs().max()), "high", float((got.high_level.cpu() - ref.high_level).abs().max()))
