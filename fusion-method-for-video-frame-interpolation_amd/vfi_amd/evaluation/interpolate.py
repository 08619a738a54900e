"""Dataset loops -- counterpart of reference src/evaluation/interpolate.py:57-209 (fusion branch): same output
naming, folder layout and skip-if-exists idempotence; each triplet is one call of `fusion_interp.interp`."""
import glob
import os
import random
from types import SimpleNamespace

import torch

from ..fusion_net import interpolate_twoframe as fusion_interp


def interpolate_fusion(args, adacof_model, fusion_net, a, b, output, output_phase, output_adacof, output_baseline):
    """interpolate.py:57-90: interpolate files a, b -> output (+ phase / adacof / baseline), skipped if output exists."""
    if os.path.exists(output):
        return False
    print("Interpolating {} and {} to {} with fusion method".format(a, b, output))
    with torch.no_grad():
        fusion_interp.interp(SimpleNamespace(
            gpu_id=args.gpu_id, adacof_model=args.fusion_adacof_model, adacof_kernel_size=args.adacof_kernel_size,
            adacof_dilation=args.adacof_dilation, first_frame=a, second_frame=b, output_frame=output,
            adacof_checkpoint=args.adacof_checkpoint, adacof_config=getattr(args, "adacof_config", None),
            checkpoint=args.fusion_checkpoint, model=getattr(args, "fusion_model", 1),
            loaded_adacof_model=adacof_model, loaded_fusion_net=fusion_net,
            high_level=getattr(args, "fusion_replace_high_level", False), mode=getattr(args, "mode", None),
            output_phase=True, output_frame_phase=output_phase, output_adacof=True, output_frame_adacof=output_adacof,
            output_baseline=True, output_frame_baseline=output_baseline, dim=args.dim,
            phase_net_checkpoint=getattr(args, "phasenet_checkpoint", "./src/phase_net/phase_net.pt")))
    return True


def dataset_plan(args, dataset_path, max_num=None, rng=random):
    """The (input a, input b, output paths) list of interpolate.py:100-158 for a folder of frames: triplet i uses
    frames i and i+2 and writes `<i+1 zero-padded to 4>.png` under <base_dir>/<img_output>/<dataset>/{fusion,...}."""
    name = os.path.basename(dataset_path)
    files = sorted(glob.glob("{}/*.png".format(dataset_path))) or sorted(glob.glob("{}/*.jpg".format(dataset_path)))
    num = len(files) - 2
    start, end = 0, num
    if max_num and max_num < num:
        start = rng.randint(0, num - max_num)
        end = start + max_num
    plan = []
    for i in range(start, end):
        fname = "{}.png".format(str(i + 1).zfill(4))
        out = {k: os.path.join(args.base_dir, args.img_output, name, k, fname)
               for k in ("fusion", "phasenet", "adacof", "baseline")}
        plan.append((files[i], files[i + 2], out))
    return plan


def interpolate_dataset(args, adacof_model, fusion_net, dataset_path="", max_num=None):
    if getattr(args, "vimeo_testset", False):
        return interpolate_vimeo_testset(args, adacof_model, fusion_net)
    done = 0
    for a, b, out in dataset_plan(args, dataset_path, max_num):
        if getattr(args, "fusion", True):
            for p in out.values():
                os.makedirs(os.path.dirname(p), exist_ok=True)
            done += bool(interpolate_fusion(args, adacof_model, fusion_net, a, b, out["fusion"], out["phasenet"],
                                            out["adacof"], out["baseline"]))
    return done


def interpolate_vimeo_testset(args, adacof_model, fusion_net, root=None):
    """interpolate.py:161-209: triplets listed in tri_testlist.txt, im1/im3 -> im2 (under <testset_root>/vimeo_interp_test,
    the same root the scoring side reads)."""
    if root is None:
        root = os.path.join(getattr(args, "testset_root", "Testset"), "vimeo_interp_test")
    with open(os.path.join(root, "tri_testlist.txt")) as f:
        triplets = [x.strip() for x in f.readlines() if x.strip()]
    done = 0
    for t in triplets:
        im1, im3 = os.path.join(root, "input", t, "im1.png"), os.path.join(root, "input", t, "im3.png")
        out = {k: os.path.join(args.base_dir, args.img_output, k, t, "im2.png")
               for k in ("fusion", "phasenet", "adacof", "baseline")}
        for p in out.values():
            os.makedirs(os.path.dirname(p), exist_ok=True)
        done += bool(interpolate_fusion(args, adacof_model, fusion_net, im1, im3, out["fusion"], out["phasenet"],
                                        out["adacof"], out["baseline"]))
    return done
