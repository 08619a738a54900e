"""Probe: N fused frames sequentially vs two in flight (tests/test_pipeline_gpu.py::test_two_frames_in_flight_equal_sequential_1080p)
with EVERY stage output compared bitwise: which stage is timing dependent?"""
import os, sys, types, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from oracle import pipeline_cpu
from vfi_amd.adacof.models import Model
from vfi_amd.fusion_net.fusion_net import FusionNet
from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
dev = torch.device("cuda:0")
weights = pipeline_cpu.seeded_weights(4)
args = types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0)
adacof = Model(args); adacof.load(weights["adacof"]); adacof.eval()
fusion = FusionNet().to(dev); fusion.load_state_dict(weights["fusionnet"]); fusion.eval()
g = torch.Generator().manual_seed(5)
pairs = [(torch.rand((3, h, w), generator=g).to(dev), torch.rand((3, h, w), generator=g).to(dev)) for _ in range(2)]
n = 6
seq_runner = FusionInterpolator(adacof, fusion, weights["phasenet"], dev)
seq = [{k: v.clone() for k, v in seq_runner(*pairs[i % 2], output_baseline=True).items()} for i in range(n)]
torch.cuda.synchronize()
runners = [FusionInterpolator(adacof, fusion, weights["phasenet"], dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
outs = []
for i in range(n):
    with torch.cuda.stream(streams[i % 2]):
        outs.append(runners[i % 2](*pairs[i % 2], output_baseline=True))
torch.cuda.synchronize()
for i in range(n):
    bad = {k: (int((outs[i][k] != seq[i][k]).sum()), float((outs[i][k] - seq[i][k]).abs().max())) for k in seq[i] if not torch.equal(outs[i][k], seq[i][k])}
    print(i, bad if bad else "identical")
print("seq self-consistency:", all(torch.equal(seq[0][k], seq[2][k]) for k in seq[0]))
