"""Probe: ONE fused frame in a fresh process against the oracle (stage PSNRs) -- hunting an intermittent first-call error."""
import math, os, sys, types, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from oracle import pipeline_cpu, synth
from vfi_amd.adacof.models import Model
from vfi_amd.fusion_net.fusion_net import FusionNet
from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 160)
dev = torch.device("cuda:0")
weights = pipeline_cpu.seeded_weights(0)
f0, f1, f2 = (torch.from_numpy(x) for x in synth.translating_pair(7, h, w))
ref = pipeline_cpu.interp(f0, f2, weights, output_baseline=True)
adacof = Model(types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0))
adacof.load(weights["adacof"]); adacof.eval()
fusion = FusionNet().to(dev); fusion.load_state_dict(weights["fusionnet"]); fusion.eval()
run = FusionInterpolator(adacof, fusion, weights["phasenet"], dev)
psnr = lambda a, b: 10 * math.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-30))
for rep in range(3):
    got = run(f0.to(dev), f2.to(dev), output_baseline=True)
    torch.cuda.synchronize()
    print(rep, {k: round(psnr(got[k].cpu(), ref[k]), 1) for k in ("phase_pred", "baseline", "ada_uncertainty", "final")}, flush=True)
