"""Probe: run the same fused frame twice in a fresh process and compare every pyramid call's outputs between the two runs."""
import math, os, sys, types, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from oracle import pipeline_cpu, synth
from vfi_amd.adacof.models import Model
from vfi_amd.fusion_net.fusion_net import FusionNet
from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator
from vfi_amd.train import pyramid as pyrmod
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 160)
dev = torch.device("cuda:0")
weights = pipeline_cpu.seeded_weights(0)
f0, f1, f2 = (torch.from_numpy(x) for x in synth.translating_pair(7, h, w))
adacof = Model(types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0))
adacof.load(weights["adacof"]); adacof.eval()
fusion = FusionNet().to(dev); fusion.load_state_dict(weights["fusionnet"]); fusion.eval()
run = FusionInterpolator(adacof, fusion, weights["phasenet"], dev)
log = []
def flat(x, out):
    if torch.is_tensor(x): out.append(x.detach().clone())
    elif isinstance(x, (list, tuple)): [flat(y, out) for y in x]
    elif hasattr(x, "phase"): [flat(y, out) for y in (x.high_level, x.phase, x.amplitude, x.low_level)]
    return out
def wrap(cls, name):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        r = orig(self, *a, **k)
        torch.cuda.synchronize()
        log.append((name, flat(r, [])))
        return r
    setattr(cls, name, f)
for n in ("filter", "inv_filter", "band_filter", "band_filter_pair"):
    wrap(pyrmod.Pyramid, n)
runs = []
for rep in range(2):
    log.clear()
    got = run(f0.to(dev), f2.to(dev), output_baseline=True)
    torch.cuda.synchronize()
    runs.append([(n, [t.clone() for t in ts]) for n, ts in log])
for (n0, a), (n1, b) in zip(runs[0], runs[1]):
    bad = [(i, tuple(x.shape), int((x != y).sum()), float((x - y).abs().max())) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]
    print(n0, "identical" if not bad else bad[:6])
    for i, x in enumerate(a):
        nan = torch.isnan(x)
        if nan.any() and x.dim() == 4:
            planes = [(c, j) for c in range(x.shape[0]) for j in range(x.shape[1]) if nan[c, j].any()]
            c, j = planes[0]
            rows = nan[c, j].any(1).nonzero().flatten().tolist(); cols = nan[c, j].any(0).nonzero().flatten().tolist()
            print(f"   NaN in run 0, output {i} {tuple(x.shape)}: planes {planes}; first: rows {rows[:4]}..{rows[-2:]} ({len(rows)}) cols {cols[:4]}..{cols[-2:]} ({len(cols)})")
