"""Probe: dump two stage outputs of the fused frame (seed 11, the tests' pair) as float16-difference-safe .npy files under
gpurun_out/, to compare two builds / VFI_PYR_WAVE settings off the box.  usage: frame_dump.py H W tag"""
import os, sys, types, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from oracle import pipeline_cpu, synth
from vfi_amd.adacof.models import Model
from vfi_amd.fusion_net.fusion_net import FusionNet
from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator
h, w, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda:0")
f0, f1, f2 = (torch.from_numpy(x) for x in synth.translating_pair(11, h, w))
weights = pipeline_cpu.seeded_weights(0)
adacof = Model(types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0))
adacof.load(weights["adacof"]); adacof.eval()
fusion = FusionNet().to(dev); fusion.load_state_dict(weights["fusionnet"]); fusion.eval()
got = FusionInterpolator(adacof, fusion, weights["phasenet"], dev)(f0.to(dev), f2.to(dev), output_baseline=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for k in ("phase_pred", "ada_uncertainty"):
    np.save(os.path.join(ROOT, "gpurun_out", f"dump_{tag}_{k}.npy"), got[k].cpu().numpy())
print("dumped", tag)
