"""N>1 path on CPU: world_size-2 gloo process groups exercising the shard partition, the single-buffer
parameter broadcast and the counter reduction of vfi_amd.shard (no GPU, no HIP calls)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from vfi_amd import shard


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            blocks = [shard.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = shard.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    from vfi_amd.fusion_net.fusion_net import FusionNet
    torch.manual_seed(100 + rank)                     # ranks start with DIFFERENT weights
    net = FusionNet()
    before = net.encoder_layers[0].weight.detach().clone()
    n = shard.broadcast_module_states([net], src=0)
    after = net.encoder_layers[0].weight.detach().clone()

    class FakeRunner:                                 # the data path needs no collective: any per-pair function shards
        def __call__(self, a, b, output_baseline=False):
            return {"final": (a + b) / 2}
    frames = [torch.full((3, 2, 2), float(i)) for i in range(8)]
    mine = shard.interpolate_clip(FakeRunner(), frames, rank, world)
    total, tmax = shard.reduce_counters(len(mine), 1.0 + rank, torch.device("cpu"))
    ret[rank] = dict(n=n, changed=not torch.equal(before, after), w0=after.flatten()[:4].tolist(),
                     pairs=sorted(mine), vals=[float(v.mean()) for _, v in sorted(mine.items())], total=total, tmax=tmax)
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_broadcast_shard_and_reduce():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert a["n"] == b["n"] == 629350                 # FusionNet parameter count
    assert a["w0"] == b["w0"] and b["changed"] and not a["changed"]
    assert a["pairs"] == [0, 1, 2, 3] and b["pairs"] == [4, 5, 6]
    assert a["vals"] == [0.5, 1.5, 2.5, 3.5] and b["vals"] == [4.5, 5.5, 6.5]
    assert a["total"] == b["total"] == 7 and a["tmax"] == b["tmax"] == 2.0


def test_evaluate_needs_an_explicit_base_dir_under_several_ranks(monkeypatch):
    """vfi_amd.evaluation.evaluate under torchrun: the default --base_dir embeds each process's start time, so the ranks
    would write to different folders -- refused before anything touches a GPU; rank r runs on GPU LOCAL_RANK."""
    import pytest
    from vfi_amd.evaluation import evaluate as evl
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="base_dir"):
        evl.eval(evl.parser.parse_args(["--fusion"]))
