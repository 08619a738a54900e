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


GENERATORS = {"adacof": gen_adacof}


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
