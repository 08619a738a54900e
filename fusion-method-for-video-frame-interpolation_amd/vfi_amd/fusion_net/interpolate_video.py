"""Clip interpolation over a frame directory -- counterpart of reference src/fusion_net/interpolate_video.py:50-123
(the I/O step either side of the hot path).

Same contract: frames `<index zero-padded to zpad>.png` in `args.input_video`, starting at `args.index_from`;
the output directory receives the originals at even positions and the interpolated frames at odd positions
(interpolate_video.py:103,112,119), written with torchvision.save_image's quantisation.

Execution differs from the reference's strictly serial loop (decode -> interp -> encode per frame, models and
pyramids rebuilt per frame):
  * PNG decode runs ahead on a thread pool (PIL releases the GIL), frames are uploaded from pinned memory;
  * frame pairs are independent, so `frames_in_flight` of them run concurrently on separate HIP streams, each with
    its own FusionInterpolator state (pyramid plan / workspace), sharing the weights;
  * PNG encode of finished frames runs on the pool while the GPU works on the next pairs;
  * with torch.distributed initialised (one process per GPU) every rank takes a contiguous block of pairs
    (vfi_amd.shard.shard_range) -- no collective on the data path.
Unlike the reference, frames are NOT centre-cropped to `dim` unless `args.dim` is given (interpolate_twoframe.py:109-113
crops to 512 by default); H and W must be multiples of 8 (FusionNet's three 2x poolings).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import shard
from .interpolate_twoframe import FusionInterpolator, build_models, crop_center


def frame_path(base, index, zpad):
    return os.path.join(base, str(index).zfill(zpad) + ".png")


def count_frames(base_dir):
    """interpolate_video.py:80: every regular file in the directory counts as a frame."""
    return len([n for n in os.listdir(base_dir) if os.path.isfile(os.path.join(base_dir, n))])


def output_indices(pair_idx, index_from):
    """(position of the first original, position of the interpolated frame) for pair `pair_idx` (0-based)."""
    return pair_idx * 2 + index_from, pair_idx * 2 + 1 + index_from


def _decode(path, dim=None):
    from PIL import Image
    img = np.array(Image.open(path))
    if img.ndim == 2:
        img = np.stack([img] * 3, -1)
    img = img[..., :3]
    if dim:
        img = crop_center(img, dim, dim)
    return torch.from_numpy(np.ascontiguousarray(img))          # (H,W,3) uint8


def _encode(arr_hwc_u8, path, event=None):
    from PIL import Image
    if event is not None:
        event.synchronize()                 # the D2H copy of this frame (enqueued on its compute stream) has landed
    if torch.is_tensor(arr_hwc_u8):
        arr_hwc_u8 = arr_hwc_u8.numpy()
    Image.fromarray(arr_hwc_u8).save(path, compress_level=1)    # same pixels as the default level, ~4x faster


def _quantise_to_host(t):
    """(3,H,W) float in [0,1] on device -> (H,W,3) uint8 pinned host tensor (torchvision.save_image quantisation),
    enqueued on the current stream; returns (host tensor, event to wait for)."""
    q = t.detach().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).contiguous()
    host = torch.empty(q.shape, dtype=torch.uint8, pin_memory=True)
    host.copy_(q, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return host, ev


def interpolate_video(args, runners=None, loaded_adacof_model=None, loaded_fusion_net=None, frames_in_flight=2,
                      io_threads=4, rank=0, world=1):
    """Interpolates every consecutive pair of `args.input_video` into `args.output_video`.
    Returns the number of interpolated frames written by this rank."""
    device = torch.device("cuda:{}".format(args.gpu_id))
    torch.cuda.set_device(device)
    if runners is None:
        if loaded_adacof_model is not None:
            args.loaded_adacof_model = loaded_adacof_model
        if loaded_fusion_net is not None:
            args.loaded_fusion_net = loaded_fusion_net
        adacof, fusion = build_models(args, device)
        pn = getattr(args, "phase_net_checkpoint", "./src/phase_net/phase_net.pt")
        state = torch.load(pn, map_location="cpu") if pn and os.path.exists(pn) else None
        runners = [FusionInterpolator(adacof, fusion, state, device) for _ in range(max(1, frames_in_flight))]
    base, out_dir = args.input_video, args.output_video
    index_from, zpad = getattr(args, "index_from", 0), getattr(args, "zpad", 3)
    dim = getattr(args, "dim", None)
    os.makedirs(out_dir, exist_ok=True)
    n_frames = count_frames(base)
    lo, hi = shard.shard_range(n_frames - 1, rank, world)
    streams = [torch.cuda.Stream(device=device) for _ in runners]
    pool = ThreadPoolExecutor(max_workers=io_threads)
    decoded = {}

    def want(i):                      # schedule decode of frame i (absolute pair-space index)
        if i not in decoded and lo <= i <= hi:
            decoded[i] = pool.submit(_decode, frame_path(base, i + index_from, zpad), dim)

    def upload(i):
        host = decoded[i].result()
        return host.pin_memory().to(device, non_blocking=True).permute(2, 0, 1).float().div_(255)

    for i in range(lo, min(hi, lo + 2 * len(runners)) + 1):
        want(i)
    writes, written = [], 0
    prev = None                       # (index, device tensor, upload event) of the pair's second frame
    for k, i in enumerate(range(lo, hi)):
        want(i + 2 * len(runners))
        want(i + 2 * len(runners) + 1)
        s = streams[k % len(runners)]
        with torch.cuda.stream(s):
            if prev and prev[0] == i:            # second frame of the previous pair, uploaded on another stream
                s.wait_event(prev[2])
                f0 = prev[1]
                f0.record_stream(s)
            else:
                f0 = upload(i)
            f1 = upload(i + 1)
            up_ev = torch.cuda.Event()
            up_ev.record(s)                      # the next pair waits for this upload only, not for this pair's compute
            out = runners[k % len(runners)](f0, f1)["final"][0]
            host, ev = _quantise_to_host(out)    # D2H on the same stream; a pool thread waits for it and encodes
        prev = (i + 1, f1, up_ev)
        a, b = output_indices(i, index_from)
        writes.append(pool.submit(_encode, decoded[i].result(), frame_path(out_dir, a, zpad)))
        writes.append(pool.submit(_encode, host, frame_path(out_dir, b, zpad), ev))
        written += 1
        decoded.pop(i - 1, None)
        while len(writes) > 8 * len(runners):    # bound the pinned buffers / queued encodes
            writes.pop(0).result()
    # last original frame (interpolate_video.py:116-119): written by the ONE rank that owns the last pair (ranks whose
    # shard is empty -- world > pairs -- also end at hi == n_frames - 1 and must not write it again)
    if hi == n_frames - 1 and n_frames > 0 and (hi > lo or (n_frames == 1 and rank == 0)):
        last = decoded[hi].result() if hi in decoded else _decode(frame_path(base, hi + index_from, zpad), dim)
        writes.append(pool.submit(_encode, last, frame_path(out_dir, (n_frames - 1) * 2 + index_from, zpad)))
    for w in writes:
        w.result()
    pool.shutdown()
    torch.cuda.synchronize(device)
    return written
