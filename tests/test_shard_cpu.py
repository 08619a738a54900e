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


def _rank_with_a_dead_peer(rank, world, port, q):
    """Rank 1 dies before the rendezvous (a crashed process, a GPU that did not come up); rank 0 must give up after the
    explicit timeout with a ShardError that names itself and the step, not hang for the backend's default half hour."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if rank == 1:
        os._exit(7)
    try:
        shard.init_distributed(backend="gloo", timeout_s=8)
        q.put("joined?")
    except shard.ShardError as e:
        q.put(str(e))
        q.close()
        q.join_thread()        # (the message must have left this process before it exits)
        os._exit(3)            # what bench.py does: non-zero, message already on stderr


def test_a_failing_rank_makes_the_others_exit_with_a_named_error():
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_with_a_dead_peer, args=(r, 2, port, q)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode is not None for p in procs), "a rank is still waiting"
    assert procs[1].exitcode == 7 and procs[0].exitcode == 3
    msg = q.get(timeout=5)
    assert msg.startswith("[vfi shard] rank 0: init_process_group(backend='gloo', world_size=2, timeout=8 s) failed"), msg
    assert time.time() - t0 < 90


def test_bench_parent_relays_the_first_failing_ranks_stderr(tmp_path):
    """bench.spawn_ranks' post-mortem: from the launcher's per-rank logs, the stderr of the rank that failed first."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    run = tmp_path / "none_abc" / "attempt_0"
    for r, (text, delay) in enumerate([("[vfi shard] rank 0: broadcast failed: peer gone\n", 0.2), ("all fine\n", 0.0),
                                       ("Traceback (most recent call last):\n  boom on rank 2\n", 0.0)]):
        d = run / str(r)
        d.mkdir(parents=True)
        (d / "stderr.log").write_text(text)
        os.utime(d / "stderr.log", (1000.0 + r, 1000.0 + (10 if r == 0 else r)))      # rank 2's log stopped first
    rank, tail = bench.first_failing_rank_log(str(tmp_path))
    assert rank == 2 and "boom on rank 2" in tail


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
