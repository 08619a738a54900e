"""Clip / dataset loops either side of the hot path: naming, ordering and idempotence on CPU with a
stand-in interpolator, and one real run over a tiny PNG directory on the GPU."""
import os
import types

import numpy as np
import pytest
import torch

from vfi_amd.evaluation import interpolate as ev
from vfi_amd.fusion_net import interpolate_video as iv


def test_output_numbering_matches_reference_loop():
    # reference interpolate_video.py:103,112,119 with index_from=1: originals at (idx-1)*2+1, interpolated one later
    assert iv.output_indices(0, 1) == (1, 2) and iv.output_indices(3, 1) == (7, 8)
    assert iv.frame_path("d", 7, 3) == os.path.join("d", "007.png")


def test_dataset_plan_names_and_window(tmp_path):
    d = tmp_path / "Clip9"
    d.mkdir()
    for i in range(6):
        (d / f"{i:03d}.png").write_bytes(b"x")
    args = types.SimpleNamespace(base_dir=str(tmp_path), img_output="out")
    plan = ev.dataset_plan(args, str(d))
    assert len(plan) == 4                                      # num = len - 2 (interpolate.py:109)
    a, b, out = plan[0]
    assert a.endswith("000.png") and b.endswith("002.png")     # frames i and i+2
    assert out["fusion"] == os.path.join(str(tmp_path), "out", "Clip9", "fusion", "0001.png")
    rng = types.SimpleNamespace(randint=lambda lo, hi: hi)
    assert len(ev.dataset_plan(args, str(d), max_num=2, rng=rng)) == 2


def test_interpolate_fusion_skips_existing(tmp_path, monkeypatch):
    calls = []
    monkeypatch.setattr(ev.fusion_interp, "interp", lambda ns: calls.append(ns.output_frame))
    args = types.SimpleNamespace(gpu_id=0, fusion_adacof_model="m", adacof_kernel_size=5, adacof_dilation=1,
                                 adacof_checkpoint=None, fusion_checkpoint=None, dim=64)
    out = tmp_path / "o.png"
    assert ev.interpolate_fusion(args, None, None, "a", "b", str(out), "p", "q", "r") is True
    out.write_bytes(b"x")
    assert ev.interpolate_fusion(args, None, None, "a", "b", str(out), "p", "q", "r") is False   # interpolate.py:63
    assert calls == [str(out)]


@pytest.mark.gpu
def test_interpolate_video_directory_on_gpu(tmp_path, device):
    from PIL import Image
    from oracle import pipeline_cpu, synth
    from vfi_amd.adacof.models import Model
    from vfi_amd.fusion_net.fusion_net import FusionNet
    from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    frames = []
    for i in range(4):
        f = synth.translating_pair(3, 64, 96, shift=(1.5 * i, -1.0 * i))[2]
        u8 = (f.transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)
        Image.fromarray(u8).save(src / f"{i:03d}.png")
        frames.append(u8)
    weights = pipeline_cpu.seeded_weights(0)
    adacof = Model(types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0))
    adacof.load(weights["adacof"]); adacof.eval()
    fusion = FusionNet().to(device); fusion.load_state_dict(weights["fusionnet"]); fusion.eval()
    runners = [FusionInterpolator(adacof, fusion, weights["phasenet"], device) for _ in range(2)]
    args = types.SimpleNamespace(gpu_id=0, input_video=str(src), output_video=str(dst), index_from=0, zpad=3)
    n = iv.interpolate_video(args, runners=runners)
    assert n == 3 and sorted(os.listdir(dst)) == [f"{i:03d}.png" for i in range(7)]
    for i in range(4):                                         # originals at even positions, bit exact
        assert np.array_equal(np.array(Image.open(dst / f"{2 * i:03d}.png")), frames[i])
    # an interpolated frame equals the runner's own output for that pair, quantised like save_image
    t = lambda u8: torch.from_numpy(u8).to(device).permute(2, 0, 1).float() / 255
    ref = runners[0](t(frames[1]), t(frames[2]))["final"][0]
    ref_u8 = ref.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).cpu().numpy()
    got = np.array(Image.open(dst / "003.png"))
    assert np.abs(got.astype(int) - ref_u8.astype(int)).max() <= 1


@pytest.mark.gpu
def test_evaluate_driver_on_a_tiny_testset(tmp_path, device):
    """vfi_amd.evaluation.evaluate.eval (reference src/evaluation/evaluate.py:219-278): models built once, dataset loop,
    device scoring, .npy cache, round-robin sharding of the test sets over ranks."""
    from PIL import Image
    from oracle import pipeline_cpu, synth
    from vfi_amd.evaluation import evaluate as evl
    root = tmp_path / "Testset"
    frames = {}
    for name, seed in (("ClipA", 3), ("ClipB", 4)):
        (root / name).mkdir(parents=True)
        for i in range(5):
            f = synth.translating_pair(seed, 72, 104, shift=(1.5 * i, -1.0 * i))[2]
            u8 = (f.transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)
            Image.fromarray(u8).save(root / name / f"{i:03d}.png")
            frames[(name, i)] = u8
    w = pipeline_cpu.seeded_weights(0)
    torch.save(w["fusionnet"], tmp_path / "fusion_net.pt")
    torch.save(w["phasenet"], tmp_path / "phase_net.pt")
    torch.save({"epoch": 0, "state_dict": w["adacof"]}, tmp_path / "ckpt.pth")
    argv = ["--fusion", "--phase", "--adacof", "--baseline", "--base_dir", str(tmp_path / "Eval"), "--test_sets", "ClipA", "ClipB",
            "--testset_root", str(root), "--dim", "64", "--max_num", "10", "--adacof_checkpoint", str(tmp_path / "ckpt.pth"),
            "--adacof_config", "", "--phasenet_checkpoint", str(tmp_path / "phase_net.pt"),
            "--fusion_checkpoint", str(tmp_path / "fusion_net.pt")]
    args = evl.parser.parse_args(argv)
    res = evl.eval(args, rank=1, world=2)                      # rank 1 of 2 owns ClipB only
    assert list(res) == ["ClipB"] and not (tmp_path / "Eval" / "result_ClipA.npy").exists()
    res = evl.eval(evl.parser.parse_args(argv), rank=0, world=1)
    assert list(res) == ["ClipA", "ClipB"]
    for name in ("ClipA", "ClipB"):
        r = res[name]
        assert r.shape == (3, 4, 7)                            # 3 triplets x (adacof, phase, fusion, baseline) x 7 measures
        assert np.isnan(r[:, :, 1]).all() and np.isfinite(np.delete(r, 1, axis=2)).all()
        for kind in ("fusion", "phasenet", "adacof", "baseline"):
            assert sorted(os.listdir(tmp_path / "Eval" / "interpolated" / name / kind)) == ["0001.png", "0002.png", "0003.png"]
        # PSNR / SSD columns re-derived on the host from the written PNGs (fusion = method index 2)
        for i in range(3):
            pred = np.array(Image.open(tmp_path / "Eval" / "interpolated" / name / "fusion" / f"{i + 1:04d}.png")).astype(np.float64) / 255
            tgt = frames[(name, i)].astype(np.float64) / 255          # the reference's target index (evaluate.py:151,158)
            crop = lambda a: a[(72 // 2 - 32):(72 // 2 + 32), (104 // 2 - 32):(104 // 2 + 32)]
            d = pred - crop(tgt)                                   # predictions are written at dim x dim already
            assert abs(r[i, 2, 2] - 10 * np.log10(1.0 / ((d ** 2).mean() + 1e-8))) <= 1e-3
            assert abs(r[i, 2, 3] - np.sqrt((d ** 2).sum())) <= 1e-3
    # --middle_frame_target scores against frame i+1 (the true middle frame) instead
    args_m = evl.parser.parse_args(argv + ["--middle_frame_target", "--test_sets", "ClipA"])
    mid = np.array(evl.evaluate_dataset(args_m, "ClipA"))
    assert mid.shape == (3, 4, 7) and not np.allclose(mid[:, 2, 2], res["ClipA"][:, 2, 2])
    # a set without predictions raises (the reference's IndexError) and leaves NO cache entry behind: an empty array saved
    # as result_<set>.npy would be taken for a result by every later run
    (root / "ClipC").mkdir()                                   # too few frames: nothing gets interpolated
    Image.fromarray(frames[("ClipA", 0)]).save(root / "ClipC" / "000.png")
    argv_c = [a if a not in ("ClipA", "ClipB") else "ClipC" for a in argv]
    argv_c = argv_c[:argv_c.index("--test_sets") + 2] + argv_c[argv_c.index("--test_sets") + 3:]      # one test set
    with pytest.raises(IndexError):
        evl.eval(evl.parser.parse_args(argv_c), rank=0, world=1)
    assert not (tmp_path / "Eval" / "result_ClipC.npy").exists()
    pred = np.array(Image.open(tmp_path / "Eval" / "interpolated" / "ClipA" / "fusion" / "0001.png")).astype(np.float64) / 255
    d = pred - (frames[("ClipA", 1)].astype(np.float64) / 255)[4:68, 20:84]
    assert abs(mid[0, 2, 2] - 10 * np.log10(1.0 / ((d ** 2).mean() + 1e-8))) <= 1e-3
    # second pass: everything is cached (PNGs skipped, arrays loaded)
    stamp = os.path.getmtime(tmp_path / "Eval" / "result_ClipA.npy")
    res2 = evl.eval(evl.parser.parse_args(argv), rank=0, world=1)
    assert os.path.getmtime(tmp_path / "Eval" / "result_ClipA.npy") == stamp and np.array_equal(res2["ClipA"], res["ClipA"], equal_nan=True)
