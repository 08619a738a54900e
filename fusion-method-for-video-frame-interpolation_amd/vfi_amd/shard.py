"""Frame-sharded multi-GPU execution: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in CPU tests).

The path has no cross-frame dependency (reference src/evaluation/interpolate.py:121-158 loops over
independent frame pairs), so the clip is partitioned into contiguous blocks of pair indices and the data
path needs NO collective.  Collectives used: one broadcast of all model parameters (one flat buffer) from
rank 0 at start-up, and one all_reduce of the (frames, seconds) counters at the end.
"""
import datetime
import os
import sys

import torch
import torch.distributed as dist


class ShardError(RuntimeError):
    """A collective step of the sharded run failed on this rank (rendezvous, weight broadcast, counter reduction).  The
    message names the rank, the step and the backend's own error, so that the launcher's log of the FIRST failing rank
    says what happened; callers exit non-zero on it (bench.py, evaluation) -- never retry, never re-exec."""


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def default_timeout_s():
    """Seconds a rank waits in the rendezvous / a collective before giving up (VFI_DIST_TIMEOUT_S, default 180): a rank
    that died must not leave the others hanging for the backend's half-hour default."""
    return float(os.environ.get("VFI_DIST_TIMEOUT_S", "180"))


def _guard(what, fn):
    rank = env_world()[0]
    try:
        return fn()
    except ShardError:
        raise
    except Exception as e:          # noqa: BLE001 -- whatever the backend raises (DistBackendError, RuntimeError, socket errors)
        msg = f"[vfi shard] rank {rank}: {what} failed: {type(e).__name__}: {e}"
        print(msg, file=sys.stderr, flush=True)
        raise ShardError(msg) from e


def init_distributed(backend=None, timeout_s=None):
    """Initialises the default process group from the torchrun environment (no-op for one process), with an explicit
    timeout; a rank that cannot join raises ShardError (rank and backend error in the message and on stderr)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("VFI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))     # (== local_rank on a real node)
        timeout = datetime.timedelta(seconds=timeout_s if timeout_s is not None else default_timeout_s())
        _guard(f"init_process_group(backend={backend!r}, world_size={world}, timeout={timeout.total_seconds():g} s)",
               lambda: dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout))
    return rank, local_rank, world


def shard_range(num_items, rank, world):
    """Contiguous block [lo, hi) of `num_items` for `rank`; blocks differ in size by at most one."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_module_states(modules, src=0):
    """One broadcast of every parameter and floating-point buffer of `modules` (flattened into a single
    buffer, so xGMI sees one large message instead of ~200 small ones).  Returns the number of floats."""
    tensors = []
    for m in modules:
        tensors += [p.data for p in m.parameters()]
        tensors += [b for b in m.buffers() if b.dtype.is_floating_point]
    if not tensors:
        return 0
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    if dist.is_initialized() and dist.get_world_size() > 1:
        what = f"broadcast of {flat.numel()} parameters from rank {src} (backend {dist.get_backend()})"
        if dist.get_backend() == "gloo" and flat.is_cuda:     # CPU rehearsal backend: stage through host memory
            host = flat.cpu()
            _guard(what, lambda: dist.broadcast(host, src=src))
            flat = host.to(flat.device)
        else:
            def bcast():
                dist.broadcast(flat, src=src)
                if flat.is_cuda:
                    torch.cuda.synchronize(flat.device)          # (an RCCL failure surfaces here, not at the enqueue)
            _guard(what, bcast)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    for m in modules:
        inv = getattr(m, "_invalidate", None)
        for sub in m.modules():
            inv = getattr(sub, "_invalidate", None)
            if inv:
                inv()
    return off


def reduce_counters(frames, seconds, device):
    """-> (total frames over all ranks, max seconds over ranks)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return frames, seconds
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    s = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    def reduce():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(s, op=dist.ReduceOp.MAX)
        return int(round(t.item())), s.item()
    return _guard("all_reduce of the (frames, seconds) counters", reduce)


def interpolate_clip(runner, frames, rank, world, output_baseline=False, sink=None):
    """Interpolates the middle frame of every consecutive pair (frames[i], frames[i+1]) assigned to this rank.
    `frames`: sequence of (3,H,W) device tensors (or a callable index -> tensor).  Returns {pair index: frame}."""
    n_pairs = len(frames) - 1
    lo, hi = shard_range(n_pairs, rank, world)
    out = {}
    for i in range(lo, hi):
        res = runner(frames[i], frames[i + 1], output_baseline=output_baseline)["final"]
        if sink is not None:
            sink(i, res)
        else:
            out[i] = res
    return out
