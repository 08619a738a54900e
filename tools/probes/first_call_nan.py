"""Probe: the FIRST pyramid analysis (PhaseNet concat layout, amplitude maxima) of a fresh process -- where are the NaNs?"""
import math, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd.train.pyramid import Pyramid
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 160)
height = int(math.ceil((math.log2(min(h, w)) - 3) * 2) + 2)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
img = torch.rand((6, h, w), generator=g).to(dev)
pyr = Pyramid(height, 4, math.sqrt(2), dev)
pyr.set_full_size(h, w)
for rep in range(2):
    v, bufs, amax = pyr.filter(img, concat_frames=2, phase_scale=1.0 / math.pi, amp_max_eps=1e-8)
    torch.cuda.synchronize()
    for k, (p, a) in enumerate(zip(v.phase, v.amplitude)):
        for name, t in (("phase", p), ("amp", a)):
            nan = torch.isnan(t)
            if nan.any():
                planes = [(c, j) for c in range(t.shape[0]) for j in range(t.shape[1]) if nan[c, j].any()]
                c, j = planes[0]
                rows = nan[c, j].any(1).nonzero().flatten().tolist()
                cols = nan[c, j].any(0).nonzero().flatten().tolist()
                print(f"rep {rep} list index {k} {name} {tuple(t.shape)}: NaN planes {planes}; plane {planes[0]}: rows {rows[:6]}..{rows[-3:]} ({len(rows)}), cols {cols[:6]}..{cols[-3:]} ({len(cols)})")
    print(f"rep {rep} done; amp_max NaN {int(torch.isnan(amax).sum())} high NaN {int(torch.isnan(v.high_level).sum())}", flush=True)
