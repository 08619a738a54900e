"""Evaluation driver -- counterpart of reference src/evaluation/evaluate.py:28-300 (the caller above the hot path).

    python -m vfi_amd.evaluation.evaluate --fusion --test_sets Clip1 Clip2 --dim 512 ...

Same flags and flow as the reference (`parser` :28-73, `eval` :219-278):
  1. build the AdaCoF `Model` and the `FusionNet` ONCE (:225-242) -- plus, unlike the reference, the PhaseNet state and
     the pyramid plans (the reference rebuilds those per frame inside `interp`);
  2. per test set: `interpolate.interpolate_dataset` (:245-252) -> PNGs under <base_dir>/<img_output>/<set>/{fusion,
     phasenet,adacof,baseline}/, skipped when they exist;
  3. per test set: `evaluate_dataset` (:76-212) scores every prediction against the ground-truth frame
     Testset/<set>/<i>.png with `evaluate_image` and caches the array as <base_dir>/result_<set>.npy (:266-278).
Differences in execution, not results: predictions and targets are decoded to the GPU and scored there
(vfi_amd.evaluation.evaluate_image: PSNR/SSIM/ad-hoc measures as deterministic device reductions; LPIPS column NaN --
needs pretrained VGG weights that ship neither with the reference nor with this image); with torch.distributed
initialised (one process per GPU, rank r on GPU LOCAL_RANK) the test sets are dealt round-robin to the ranks -- no
collective: every rank writes its own result_<set>.npy and the caches make a second pass on any rank complete; an
explicit --base_dir is required then, and a set without predictions raises instead of caching an empty array.  The matplotlib figures of
visualizations.py (:254-258,299-300) are out of scope; `eval` returns the per-set arrays and prints NaN-aware means.
"""
import argparse
import glob
import os
import random
import warnings
from datetime import datetime
from types import SimpleNamespace

import numpy as np
import torch

from .. import shard
from ..adacof.models import Model
from ..fusion_net.fusion_net import FusionNet
from . import evaluate_image, interpolate

MEASURES = ("ssim", "lpips", "psnr", "ssd", "l1", "mse", "variance")       # evaluate_image.py:30 order

parser = argparse.ArgumentParser(description="Evaluation")
# Evaluation Parameters (evaluate.py:31-45)
parser.add_argument("--gpu_id", type=int, default=0)
parser.add_argument("--adacof", action="store_true")
parser.add_argument("--phase", action="store_true")
parser.add_argument("--fusion", action="store_true")
parser.add_argument("--baseline", action="store_true")
parser.add_argument("--base_dir", type=str, default=os.path.join("Evaluation", datetime.today().strftime("%Y-%m-%d-%H:%M:%S")))
parser.add_argument("--img_output", type=str, default="interpolated")
parser.add_argument("--max_num", type=int, default=10)
parser.add_argument("--seed", type=int, default=1000)
parser.add_argument("--test_sets", type=str, nargs="+",
                    default=["airboard_1", "airplane_landing", "airtable_3", "basketball_1", "water_ski_2", "yoyo",
                             "MODE_SH0280", "MODE_SH0440", "MODE_SH0450", "MODE_SH0740", "MODE_SH0780", "MODE_SH1010",
                             "MODE_SH1270", "Flashlight", "firework", "lights", "sun"])
parser.add_argument("--testset_root", type=str, default="Testset",
                    help="folder holding <set>/<frame>.png (the reference hard-codes 'Testset', evaluate.py:141,250)")
parser.add_argument("--middle_frame_target", action="store_true",
                    help="score prediction i against frame i+1 (the true middle frame) instead of the reference's "
                         "target_folder[start_index + i] (evaluate.py:151,158), which is the triplet's FIRST frame")
# AdaCoF parameters (:48-56)
parser.add_argument("--adacof_model", type=str, default="vfi_amd.adacof.models.adacofnet")
parser.add_argument("--adacof_checkpoint", type=str, default="./src/adacof/checkpoint/kernelsize_5/ckpt.pth")
parser.add_argument("--adacof_config", type=str, default="./src/adacof/checkpoint/kernelsize_5/config.txt")
parser.add_argument("--adacof_kernel_size", type=int, default=5)
parser.add_argument("--adacof_dilation", type=int, default=1)
# PhaseNet parameters (:59-61)
parser.add_argument("--phasenet_checkpoint", type=str, default="./src/phase_net/phase_net.pt")
parser.add_argument("--phasenet_replace_high_level", action="store_true")
# Fusion parameters (:64-73)
parser.add_argument("--fusion_checkpoint", type=str, default="./src/fusion_net/fusion_net.pt")
parser.add_argument("--fusion_adacof_model", type=str, default="vfi_amd.fusion_net.fusion_adacofnet")
parser.add_argument("--fusion_model", type=int, default=1)
parser.add_argument("--fusion_replace_high_level", action="store_true")
parser.add_argument("--vimeo_testset", action="store_true")
parser.add_argument("--mode", type=str, default="alpha")
parser.add_argument("--dim", type=int, default=512)


def _to_tensor(path, device):
    """torchvision TF.to_tensor(Image.open(path)) on the device: (3,H,W) float in [0,1]."""
    from PIL import Image
    arr = np.array(Image.open(path))
    if arr.ndim == 2:
        arr = np.stack([arr] * 3, -1)
    return torch.from_numpy(np.ascontiguousarray(arr[..., :3])).to(device).permute(2, 0, 1).float().div_(255)


def evaluate_dataset(args, dataset_path):
    """evaluate.py:76-212: compares the interpolated images of one test set with the ground truth (frame i+1 of the
    triplet (i, i+2)).  Returns a list of (methods, 7) arrays, methods in the order adacof, phase, fusion, baseline."""
    device = torch.device("cuda:{}".format(args.gpu_id))
    name = os.path.basename(dataset_path)
    root = getattr(args, "testset_root", "Testset")
    if getattr(args, "vimeo_testset", False):
        sub = lambda kind: sorted(glob.glob(os.path.join(args.base_dir, args.img_output, kind, dataset_path, "*", "im2.png")))
    else:
        sub = lambda kind: sorted(glob.glob(os.path.join(args.base_dir, args.img_output, name, kind, "*")))
    folders = []
    for flag, kind in (("adacof", "adacof"), ("phase", "phasenet"), ("fusion", "fusion"), ("baseline", "baseline")):
        if getattr(args, flag, False):
            folders.append(sub(kind))
    if not folders or not folders[-1]:
        return []
    num_img, first_img = len(folders[-1]), folders[-1][0]                        # :112-130 (the last enabled method wins)
    if getattr(args, "vimeo_testset", False):
        with open(os.path.join(root, "vimeo_interp_test", "tri_testlist.txt")) as f:
            triplets = [x.strip() for x in f.readlines()]
        targets = sorted(os.path.join(root, "vimeo_interp_test", "target", t, "im2.png")
                         for t in triplets if t.startswith(dataset_path))
        start_index = 0
    else:
        targets = sorted(glob.glob(os.path.join(root, dataset_path, "*")))
        start_index = int(os.path.splitext(os.path.basename(first_img))[0]) - 1   # :151 (max_num window offset)
        # NB the reference indexes the ground truth with start_index + i (:158): prediction "<i+1>.png" (frames i, i+2,
        # interpolate.py:121-122) is scored against frame i.  Kept as the default so the numbers match the reference's.
        if getattr(args, "middle_frame_target", False):
            start_index += 1
    results = []
    for i in range(num_img):
        target = _to_tensor(targets[start_index + i], device)
        rows = [evaluate_image.evaluate_image(args, _to_tensor(f[i], device), target) for f in folders]
        results.append(np.stack(rows))
    return results


def build_models(args):
    """evaluate.py:225-242: AdaCoF `Model` (fusion variant, dotted path) + FusionNet, built and loaded once."""
    device = torch.device("cuda:{}".format(args.gpu_id))
    adacof_model = Model(SimpleNamespace(gpu_id=args.gpu_id, model=args.fusion_adacof_model, kernel_size=args.adacof_kernel_size,
                                         dilation=args.adacof_dilation, config=args.adacof_config))
    adacof_model.eval()
    adacof_model.load(torch.load(args.adacof_checkpoint, map_location=torch.device("cpu"))["state_dict"])
    fusion_net = FusionNet().to(device)
    fusion_net.load_state_dict(torch.load(args.fusion_checkpoint, map_location="cpu"))   # (reference: no map_location)
    fusion_net.eval()
    return adacof_model, fusion_net


def _run_token():
    """What the ranks of ONE launch share and no other launch has: torchrun's run id and rendezvous port.  The hand-off
    marker carries it, so a marker left behind in a reused --base_dir is never taken for this run's."""
    return "{}_{}".format(os.environ.get("TORCHELASTIC_RUN_ID", "run"), os.environ.get("MASTER_PORT", "0"))


def _wait_for(path, failed_path, timeout_s=None, poll_s=2.0):
    """File-based hand-off between ranks (no collective on the data path): blocks until `path` exists; gives up when the
    producing rank has left `failed_path` behind or after VFI_EVAL_HANDOFF_TIMEOUT_S (default 2 h)."""
    import time
    if timeout_s is None:
        timeout_s = float(os.environ.get("VFI_EVAL_HANDOFF_TIMEOUT_S", 2 * 3600.0))
    t0 = time.monotonic()
    while not os.path.exists(path):
        if os.path.exists(failed_path):
            raise RuntimeError("evaluate: rank 0 failed while interpolating ({})".format(failed_path))
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError("evaluate: waited {:.0f} s for {}".format(timeout_s, path))
        time.sleep(poll_s)


def eval(args, rank=None, world=None):           # noqa: A001  (the reference's name)
    local_rank = None
    if rank is None:
        rank, local_rank, world = shard.env_world()
    if world > 1:
        # one process per GPU: rank r works on GPU LOCAL_RANK (the reference is single-process and uses --gpu_id);
        # the default --base_dir embeds this process's start time, so the ranks would not agree on it
        if local_rank is not None:
            args.gpu_id = local_rank % max(torch.cuda.device_count(), 1)
        if args.base_dir == parser.get_default("base_dir"):
            raise SystemExit("evaluate: --base_dir must be given explicitly when WORLD_SIZE > 1 (the default is per process)")
    torch.cuda.set_device(args.gpu_id)
    random.seed(args.seed)
    root = getattr(args, "testset_root", "Testset")
    img_output_dir = os.path.join(args.base_dir, args.img_output)
    os.makedirs(img_output_dir, exist_ok=True)
    adacof_model, fusion_net = build_models(args)
    if getattr(args, "vimeo_testset", False):
        # the triplet list is one flat file: rank 0 interpolates it, the others wait for its marker before they score
        marker = os.path.join(img_output_dir, ".vimeo_interpolated." + _run_token())
        failed = marker + ".failed"
        if rank == 0:
            try:
                interpolate.interpolate_dataset(args, adacof_model, fusion_net)
            except BaseException:
                open(failed, "w").close()               # the waiting ranks stop instead of polling until their timeout
                raise
            open(marker, "w").close()
        else:
            _wait_for(marker, failed)
        args.test_sets = [os.path.basename(x) for x in sorted(glob.glob(os.path.join(root, "vimeo_interp_test", "target", "*")))]
    mine = [t for i, t in enumerate(args.test_sets) if i % world == rank]
    if not getattr(args, "vimeo_testset", False):
        for testset in mine:                                                          # :249-252
            interpolate.interpolate_dataset(args, adacof_model, fusion_net, os.path.join(root, testset), max_num=args.max_num)
    results_np = {}
    for testset in mine:                                                              # :266-278
        result_path = os.path.join(args.base_dir, "result_{}.npy".format(os.path.basename(testset)))
        if os.path.exists(result_path):
            result_np = np.load(result_path)
        else:
            rows = evaluate_dataset(args, testset)
            if not rows:
                # the reference raises IndexError here (evaluate.py:112-130 index an empty list); an empty array must never
                # become a cache entry that later runs would take for a result
                raise IndexError("evaluate: no interpolated images found for test set '{}' under {}".format(testset, img_output_dir))
            result_np = np.array(rows)
            np.save(result_path, result_np)
        results_np[testset] = result_np
        if result_np.size:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                mean = np.nanmean(result_np, axis=0)                                  # (methods, 7); LPIPS column stays NaN
            for m, row in enumerate(mean):
                print("Result for {} [method {}]: ".format(testset, m) +
                      "  ".join("{} {:.5g}".format(k, v) for k, v in zip(MEASURES, row)))
    return results_np


def main():
    eval(parser.parse_args())


if __name__ == "__main__":
    main()
