"""Probe: pyramid analysis + synthesis of the same input (a) sequentially on one stream and (b) with two plans in flight on
two streams -- bitwise comparison of every output (the wave kernels have no workgroup barriers: is anything timing dependent?)."""
import math, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd.train.pyramid import Pyramid
from vfi_amd.values import DecompValues
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
height = int(math.ceil((math.log2(min(H, W)) - 3) * 2) + 2)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
imgs = [torch.rand((6, H, W), generator=g).to(dev) for _ in range(2)]
def run(pyr, img):
    v = pyr.filter(img)
    sub = DecompValues(v.high_level[:3], [p[:12] for p in v.phase], [a[:12] for a in v.amplitude], v.low_level[:3])
    rec = pyr.inv_filter(sub)
    return [v.high_level, v.low_level, rec] + list(v.phase) + list(v.amplitude)
names = ["high", "low", "rec"] + [f"phase{k}" for k in range(height - 2)] + [f"amp{k}" for k in range(height - 2)]
pyrs = [Pyramid(height, 4, math.sqrt(2), dev) for _ in range(3)]
ref = [[t.clone() for t in run(pyrs[2], imgs[i % 2])] for i in range(2)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
bad = {}
for rep in range(6):
    outs = []
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            outs.append(run(pyrs[i], imgs[i]))
    torch.cuda.synchronize()
    for i in range(2):
        for n, a, b in zip(names, outs[i], ref[i]):
            if not torch.equal(a, b):
                d = (a - b).abs()
                bad.setdefault(n, []).append((rep, i, int((d > 0).sum()), float(d.max())))
print("mismatches:", {k: v[:3] for k, v in bad.items()} if bad else "none")
