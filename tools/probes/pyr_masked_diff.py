"""Probe: the ada-uncertainty pyramid round trip (masked analysis of two image sets, |phase| / |amp| differences of the six
coarsest levels, masked synthesis) on the GPU against the oracle, per stage -- where does a disagreement come from?"""
import math, os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from oracle import pyramid_cpu, synth, layout_cpu
from vfi_amd import ops
from vfi_amd.train.pyramid import Pyramid
from vfi_amd.values import DecompValues
dev = torch.device("cuda:0")
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 160)
f0, f1, f2 = (torch.from_numpy(x) for x in synth.translating_pair(7, h, w))
a_img, p_img = (0.5 * (f0 + f2)).contiguous(), f1.contiguous()          # stand-ins for ada_pred / rgb_pred: two similar images
if len(sys.argv) > 3 and sys.argv[3] == "frame":                        # ... or the fused path's own two predictions (seed 11, as the tests)
    import types
    from oracle import pipeline_cpu
    from vfi_amd.adacof.models import Model
    from vfi_amd.fusion_net.fusion_net import FusionNet
    from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator
    f0, f1, f2 = (torch.from_numpy(x) for x in synth.translating_pair(11, h, w))
    weights = pipeline_cpu.seeded_weights(0)
    adacof = Model(types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0))
    adacof.load(weights["adacof"]); adacof.eval()
    fusion = FusionNet().to(dev); fusion.load_state_dict(weights["fusionnet"]); fusion.eval()
    got = FusionInterpolator(adacof, fusion, weights["phasenet"], dev)(f0.to(dev), f2.to(dev))
    a_img, p_img = got["ada_pred"][0].cpu().contiguous(), got["phase_pred"][0].cpu().contiguous()
height = layout_cpu.calc_pyr_height(h, w)
cpu = pyramid_cpu.Pyramid(height, 4, np.sqrt(2))
vals = cpu.filter(torch.cat((a_img, p_img), 0).float())
va, vp = layout_cpu.separate_vals(vals, 2)
diff = layout_cpu.get_first_value_levels(layout_cpu.subtract_values(vp, va), 6)
freq_ref = cpu.inv_filter(diff)
pyr = Pyramid(height, 4, math.sqrt(2), dev)
pyr.set_full_size(h, w)
nlev = height - 2
coarse = min(6, nlev)
mask = ((1 << coarse) - 1) << (nlev - coarse)
vb = pyr.filter(torch.cat((a_img, p_img), 0).to(dev), level_mask=mask, want_high=False)
half = lambda t: (t[:12], t[12:])
dp, da = [0] * nlev, [0] * nlev
for k in range(nlev - coarse, nlev):
    dp[k] = ops.absdiff(*half(vb.phase[k])[::-1])
    da[k] = ops.absdiff(*half(vb.amplitude[k])[::-1])
    rp, ra = diff.phase[k], diff.amplitude[k]
    gp, ga = dp[k].cpu().reshape(rp.shape), da[k].cpu().reshape(ra.shape)
    bad = (gp - rp).abs() > 1
    if bad.any():
        for idx in bad.nonzero()[:6].tolist():
            i = tuple(idx)
            print(f"   flip at {i}: |dphase| ref {float(rp[i]):.6f} gpu {float(gp[i]):.6f}  |damp| {float(ra[i]):.4e}  amp(ada) {float(va.amplitude[k][i]):.4e} phase(ada) {float(va.phase[k][i]):.6f} phase(ph) {float(vp.phase[k][i]):.6f}")
    print(f"level {k} {tuple(rp.shape)}: |dphase| max err {float((gp - rp).abs().max()):.3e} (count > 1: {int(((gp - rp).abs() > 1).sum())})  |damp| max err {float((ga - ra).abs().max()):.3e}")
dlow = ops.absdiff(vb.low_level[3:], vb.low_level[:3])
print("dlow err", float((dlow.cpu() - diff.low_level).abs().max()))
freq = pyr.inv_filter(DecompValues(0, dp, da, dlow)).cpu()
print("freq max err", float((freq - freq_ref).abs().max()), "rms err", float((freq - freq_ref).pow(2).mean().sqrt()), "ref max", float(freq_ref.abs().max()))
# synthesis alone on the ORACLE's difference values
dd = DecompValues(0, [x.to(dev) if torch.is_tensor(x) else 0 for x in diff.phase], [x.to(dev) if torch.is_tensor(x) else 0 for x in diff.amplitude], diff.low_level.to(dev))
freq2 = pyr.inv_filter(dd).cpu()
print("synthesis of the oracle's values: max err", float((freq2 - freq_ref).abs().max()))
