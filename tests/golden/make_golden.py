#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the
GPU box).  The fixtures it writes are DATA (seeded inputs + the outputs the reference's
own code produced for them); no reference source is copied.

    python tests/golden/make_golden.py [--only NAME ...]

What is imported from the reference, and how:
  * src.fusion_net.fusion_net.FusionNet                      -- imports unmodified
  * src.phase_net.phase_net.PhaseNet, src.train.utils.*      -- import after registering EMPTY
    placeholder modules for packages that are absent from this image and are never
    called on the exercised path (skimage, steerable)
  * src.fusion_net.fusion_adacofnet.{KernelEstimation,AdaCoFNet} -- same, with an empty
    `cupy` placeholder providing only the `memoize` decorator (never called)
  * AdaCoF sampling: the reference has no CPU path (adacof.py:356-357).  Its kernel is a
    C-syntax string that the reference's own pure-Python specialiser `cupy_kernel()`
    (adacof.py:261-299) turns into plain C for a given shape.  That text is written to a
    TEMP dir (never into the repo), compiled with g++ together with a host loop that
    iterates the launch grid, run on the seeded inputs and deleted again.
"""
import argparse
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

REF = os.environ.get("VFI_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _placeholders():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    sk = mod("skimage")
    sk.io = mod("skimage.io")
    sk.color = mod("skimage.color")
    st = mod("steerable")
    st.SCFpyr_PyTorch = mod("steerable.SCFpyr_PyTorch", SCFpyr_PyTorch=object)
    st.utils = mod("steerable.utils")

    def memoize(for_each_device=False):
        return lambda fn: fn
    mod("cupy", memoize=memoize)
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
        except Exception:
            mp = mod("matplotlib")
            mp.pyplot = mod("matplotlib.pyplot")
    sys.path.insert(0, REF)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", os.path.relpath(path), f"{os.path.getsize(path)/1024:.0f} KiB")


# ----------------------------------------------------------------------------------------
# AdaCoF sampling (a12)
# ----------------------------------------------------------------------------------------
_HOST_LOOP = r"""
#include <cstdlib>
struct dim3_ { int x, y, z; };
static thread_local dim3_ blockIdx, threadIdx;
static dim3_ blockDim = {512, 1, 1}, gridDim = {1, 1, 1};
#define __global__
%(KERNEL)s
extern "C" void run(int n, const float* input, const float* weight, const float* offset_i,
                    const float* offset_j, float* output) {
    gridDim.x = (n + 512 - 1) / 512;                       /* adacof.py:349 */
    for (int b = 0; b < gridDim.x; ++b)
        for (int t = 0; t < blockDim.x; ++t) {
            blockIdx.x = b; threadIdx.x = t;
            kernel_AdaCoF_updateOutput(n, input, weight, offset_i, offset_j, output);
        }
}
"""


def adacof_inputs(seed, n, c, h, w, f, dil, big):
    rng = np.random.default_rng(seed)
    pad = (f - 1) * dil // 2
    hin, win = h + (f - 1) * dil, w + (f - 1) * dil
    del pad
    inp = rng.random((n, c, hin, win), dtype=np.float32)
    logits = rng.standard_normal((n, f * f, h, w)).astype(np.float32)
    wgt = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    amp = 12.0 if big else 2.0
    a = (rng.standard_normal((n, f * f, h, w)) * amp).astype(np.float32)
    b = (rng.standard_normal((n, f * f, h, w)) * amp).astype(np.float32)
    # exact integers, negative fractions, -0.0, huge offsets that clamp on every side
    a[:, 0] = np.round(a[:, 0]); b[:, 0] = np.round(b[:, 0])
    a[:, 1] = -np.abs(a[:, 1]); b[:, 1] = -np.abs(b[:, 1])
    a[:, 2, 0::2] = -0.0; b[:, 2, :, 0::2] = 0.0
    a[:, 3] = 1000.0 * np.sign(a[:, 3]); b[:, 3] = -1000.0 * np.sign(b[:, 3])
    a[:, 4] = np.clip(a[:, 4], -0.999, 0.999); b[:, 4] = np.clip(b[:, 4], -0.999, 0.999)
    return inp, wgt.astype(np.float32), a, b


def gen_adacof():
    _placeholders()
    import ctypes
    from src.adacof.cupy_module import adacof as ref  # reference module
    for tag, (n, c, h, w, f, dil, big) in {
        "f5d1": (1, 3, 37, 53, 5, 1, True),
        "f5d1_b2": (2, 3, 24, 40, 5, 1, False),
        "f11d2": (1, 3, 29, 31, 11, 2, True),
        "f3d1_c1": (1, 1, 16, 19, 3, 1, True),
    }.items():
        inp, wgt, a, b = adacof_inputs(len(tag) * 7 + f, n, c, h, w, f, dil, big)
        out = np.zeros((n, c, h, w), np.float32)
        tens = {k: torch.from_numpy(v) for k, v in
                dict(input=inp, weight=wgt, offset_i=a, offset_j=b, output=out).items()}
        text = ref.cupy_kernel("kernel_AdaCoF_updateOutput", f, dil, tens)  # adacof.py:261-299
        text = text.replace('extern "C" __global__', "static")
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "k.cpp")
            with open(src, "w") as fh:
                fh.write(_HOST_LOOP % {"KERNEL": text})
            so = os.path.join(td, "k.so")
            # nvcc/NVRTC contract a*b+c into fma by default; g++ -O2 on x86-64 does not
            # (no -mfma), which gives the plain fp32 evaluation.
            subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, src])
            lib = ctypes.CDLL(so)
            fp = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            lib.run(out.size, fp(inp), fp(wgt), fp(a), fp(b), fp(out))
        save("adacof_sampling_" + tag, input=inp, weight=wgt, offset_i=a, offset_j=b,
             dilation=dil, output=out)


# ----------------------------------------------------------------------------------------
# Networks: the reference's own classes loaded with the oracle's SEEDED weights (the trained
# AdaCoF checkpoint is not in the reference snapshot, and trained weights are not committed);
# fixtures hold inputs + the reference's outputs only.
# ----------------------------------------------------------------------------------------
def _oracle():
    root = os.path.dirname(os.path.dirname(HERE))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import nets_cpu, synth
    return nets_cpu, synth


def _flat(prefix, vals):
    d = {prefix + "high": vals.high_level.numpy(), prefix + "low": vals.low_level.numpy()}
    for k, (p, a) in enumerate(zip(vals.phase, vals.amplitude)):
        d[f"{prefix}phase{k}"] = p.numpy()
        d[f"{prefix}amp{k}"] = a.numpy()
    return d


def _sub(prefix, vals, step):
    """Strided subsample + per-tensor sums of a DecompValues (keeps big fixtures small)."""
    d = {}
    items = [("high", vals.high_level), ("low", vals.low_level)]
    items += [(f"phase{k}", p) for k, p in enumerate(vals.phase)] + [(f"amp{k}", a) for k, a in enumerate(vals.amplitude)]
    for name, t in items:
        d[prefix + name] = t[..., ::step, ::step].numpy()
        d[prefix + name + "_sum"] = t.double().sum(dim=(1, 2, 3)).numpy()
    return d


def gen_phasenet():
    """Inputs come from oracle.synth.synthetic_vals(seed,...) and are NOT stored."""
    _placeholders()
    nets, synth = _oracle()
    from src.phase_net.phase_net import PhaseNet
    from src.train import utils as rutils
    from src.train.pyramid import DecompValues as RefVals
    for tag, (h, w, step) in {"32x48": (32, 48, 1), "96x112": (96, 112, 3)}.items():
        height = int(np.ceil((np.log2(min(h, w)) - 3) * 2) + 2)          # utils.py:168-171
        pyr = types.SimpleNamespace(height=height, nbands=4)
        net = PhaseNet(pyr, torch.device("cpu"), num_img=2)
        print(net.load_state_dict(nets.phasenet_random_state_dict(seed=1)))
        net.eval()
        vals_batch = RefVals(*synth.synthetic_vals(h + w, 6, h, w, height))
        vals_list = rutils.separate_vals(vals_batch, 2)                   # utils.py:83-127
        vals_in = rutils.get_concat_layers_inf(pyr, vals_list)            # utils.py:47-80
        with torch.no_grad():
            normed = net.normalize_vals(vals_in)
            out = net(normed)
        d = dict(h=h, w=w, height=height, weight_seed=1, input_seed=h + w, step=step)
        d.update(_sub("out_", out, step))
        if step == 1:
            d.update(_flat("norm_", normed))
        d["max_low"] = net.max_low_level.numpy()
        for k, mx in enumerate(net.max_amplitudes):
            d[f"max_amp{k}"] = mx.numpy()
        save("phasenet_" + tag, **d)


def gen_layout():
    """Pins separate_vals / get_concat_layers_inf / get_last|first_value_levels / subtract_values
    on an integer-coded pyramid (every element carries its own (img, band, level, y, x) code)."""
    _placeholders()
    from src.train import utils as rutils
    from src.train.pyramid import DecompValues
    _, synth = _oracle()
    h, w, height = 16, 24, 6
    bands, low = synth.level_sizes(h, w, height)
    code = lambda n, a, b, base: (base + torch.arange(n * a * b, dtype=torch.float32).reshape(n, 1, a, b))
    vals = DecompValues(high_level=code(6, h, w, 1e5), low_level=code(6, *low, 2e5),
                        phase=[code(24, a, b, 1e6 * (k + 1)) for k, (a, b) in enumerate(bands)],
                        amplitude=[code(24, a, b, -1e6 * (k + 1)) for k, (a, b) in enumerate(bands)])
    pyr = types.SimpleNamespace(height=height, nbands=4)
    lst = rutils.separate_vals(vals, 2)
    cat = rutils.get_concat_layers_inf(pyr, lst)
    last = rutils.get_last_value_levels(lst[0], use_levels=1)
    first = rutils.get_first_value_levels(lst[1], use_levels=2)
    sub = rutils.subtract_values(lst[0], lst[1])
    d = dict(h=h, w=w, height=height)
    d.update(_flat("vals_", vals)); d.update(_flat("sep0_", lst[0])); d.update(_flat("sep1_", lst[1]))
    d.update(_flat("cat_", cat)); d.update(_flat("last_", last)); d.update(_flat("first_", first))
    d.update(_flat("sub_", sub))
    save("layout_helpers", **d)


def gen_fusionnet():
    _placeholders()
    nets, _ = _oracle()
    from src.fusion_net.fusion_net import FusionNet
    net = FusionNet()
    print(net.load_state_dict(nets.fusionnet_random_state_dict(seed=2)))
    net.eval()
    for tag, (h, w) in {"64x64": (64, 64), "40x72": (40, 72)}.items():
        rng = np.random.default_rng(h * w)
        r = lambda c: torch.from_numpy(rng.random((1, c, h, w), dtype=np.float32))
        base, ada, ph, other, maps = r(3), r(3), r(3), r(6), r(3)
        with torch.no_grad():
            o0 = net(base, ada, ph, other, maps, variant=0)
            o1 = net(base, ada, ph, other, maps, variant=1)
        save("fusionnet_" + tag, seed=2, base=base.numpy(), adacof=ada.numpy(), phase=ph.numpy(),
             other=other.numpy(), maps=maps.numpy(), out_variant0=o0.numpy(), out_variant1=o1.numpy())


def gen_adacofnet():
    """KernelEstimation heads + the AdaCoFNet glue (pad to /32, normalise, flow-variance mask, crop)
    with the sampling call replaced by the value the fixture `adacof` pins (oracle C)."""
    _placeholders()
    nets, _ = _oracle()
    from oracle import adacof_cpu
    from src.fusion_net import fusion_adacofnet as ref

    class Sampler:
        @staticmethod
        def apply(inp, wgt, a, b, dil):
            return torch.from_numpy(adacof_cpu.adacof_forward(inp.numpy(), wgt.contiguous().numpy(),
                                                             a.contiguous().numpy(), b.contiguous().numpy(), dil))
    args = types.SimpleNamespace(kernel_size=5, dilation=1, gpu_id=0)
    net = ref.AdaCoFNet(args)
    net.moduleAdaCoF = Sampler.apply          # the CUDA-only op (adacof.py:356-357 raises on CPU)
    print(net.load_state_dict(nets.adacofnet_random_state_dict(seed=3)))
    net.eval()
    for tag, (h, w) in {"64x96": (64, 96), "40x50": (40, 50)}.items():
        rng = np.random.default_rng(h + 3 * w)
        f0 = torch.from_numpy(rng.random((1, 3, h, w), dtype=np.float32))
        f2 = torch.from_numpy(rng.random((1, 3, h, w), dtype=np.float32))
        with torch.no_grad():
            t1, t2, fr, mask = net(f0, f2)
            d = dict(seed=3, frame0=f0.numpy(), frame2=f2.numpy(), t1=t1.numpy(), t2=t2.numpy(),
                     frame1=fr.numpy(), mask=mask.numpy())
            if h % 32 == 0 and w % 32 == 0:
                heads = net.get_kernel(ref.moduleNormalize(f0), ref.moduleNormalize(f2))
                for name, t in zip(("w1", "a1", "b1", "w2", "a2", "b2", "occ"), heads):
                    d["head_" + name] = t[..., ::4, ::4].numpy()      # strided subsample
                    d["head_" + name + "_sum"] = t.double().sum(dim=(2, 3)).numpy()
        save("adacofnet_" + tag, **d)


def gen_phasenet_only():
    """BASELINE.json configs[0]: the PhaseNet-only flow of src/phase_net/interpolate_twoframe.py:62-104 at 256x256
    (per colour channel: filter x2 -> get_concat_layers -> normalize_vals -> PhaseNet -> inv_filter), run with the
    REFERENCE's own `Pyramid` adapter (src/train/pyramid.py), `get_concat_layers` (src/train/utils.py:19-44) and
    `PhaseNet` (src/phase_net/phase_net.py).  The one stand-in is the absent third-party transform
    `steerable.SCFpyr_PyTorch`, whose build() / reconstruct() are served by oracle/pyramid_cpu.py -- the same
    substitution as the CUDA-only sampler in gen_adacofnet.  The Lab conversion (skimage, absent) stays outside: the
    flow starts from Lab images the tests regenerate from seeds, so the fixture stores the output only."""
    _placeholders()
    nets, synth = _oracle()
    from oracle import color_cpu, layout_cpu, pyramid_cpu

    class OracleSCF:
        def __init__(self, height, nbands, scale_factor, device):
            self.height, self.nbands, self.scale_factor = height, nbands, scale_factor
            self._specs = {}

        def _spec(self, h, w):
            if (h, w) not in self._specs:
                self._specs[(h, w)] = pyramid_cpu.PyramidSpec(h, w, self.height, self.nbands, self.scale_factor)
            return self._specs[(h, w)]

        def build(self, x):
            return pyramid_cpu.build(self._spec(*x.shape[-2:]), x[:, 0])

        def reconstruct(self, coeff):
            return pyramid_cpu.reconstruct(self._spec(*coeff[0].shape[-2:]), coeff)

    import src.train.pyramid as rpyr
    rpyr.SCFpyr_PyTorch = OracleSCF
    from src.train.utils import get_concat_layers
    from src.phase_net.phase_net import PhaseNet
    h = w = 256
    f0, _, f2 = (torch.from_numpy(x) for x in synth.translating_pair(4, h, w))
    img_1, img_2 = color_cpu.rgb2lab_single(f0), color_cpu.rgb2lab_single(f2)
    pyr = rpyr.Pyramid(height=layout_cpu.calc_pyr_height(h, w), nbands=4, scale_factor=np.sqrt(2), device="cpu")
    assert pyr.height == 12
    net = PhaseNet(pyr, torch.device("cpu"))
    print(net.load_state_dict(nets.phasenet_random_state_dict(seed=1)))
    net.eval()
    result = []
    for c in range(3):                                                   # interpolate_twoframe.py:82-104
        vals_1 = pyr.filter(img_1[c].unsqueeze(0))
        vals_2 = pyr.filter(img_2[c].unsqueeze(0))
        vals_normalized = net.normalize_vals(get_concat_layers(pyr, vals_1, vals_2))
        with torch.no_grad():
            vals_r = net(vals_normalized)
        result.append(pyr.inv_filter(vals_r).detach())
    lab_pred = torch.cat(result, 0)
    save("phasenet_only_256", pair_seed=4, weight_seed=1, h=h, w=w, height=pyr.height, lab_pred=lab_pred.numpy())


GENERATORS = {"adacof": gen_adacof, "phasenet": gen_phasenet, "layout": gen_layout,
              "fusionnet": gen_fusionnet, "adacofnet": gen_adacofnet, "phasenet_only": gen_phasenet_only}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    for name, fn in GENERATORS.items():
        if args.only and name not in args.only:
            continue
        print("==", name)
        fn()


if __name__ == "__main__":
    main()
