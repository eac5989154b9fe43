"""N > 1 path on the CPU: two gloo ranks split a batch of utterances with no data-path collective and
reassemble the results in order.  The per-shard work is a stand-in (no GPU here)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from sopro_amd.sharding import run_sharded, shard_indices, unshard


def test_shard_indices_partition_every_item_once():
    for n in (0, 1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert seen == list(range(n))
            sizes = [len(shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert unshard([[10, 12], [11]], 3) == [10, 11, 12]
    with pytest.raises(ValueError):
        unshard([[1], [2]], 3)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = [torch.full((3,), float(i)) for i in range(9)]
        seen = []

        def fn(xs):
            seen.extend(int(x[0]) for x in xs)
            return [x * 2 + rank * 0 for x in xs]  # a rank-independent function of the utterance

        out = run_sharded(items, fn)
        t = torch.tensor([float(len(seen))])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the only collective the bench uses: a MAX over ranks
        q.put((rank, seen, [float(o[0]) for o in out], float(t)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_run_reassembles_in_order():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7]
    for _rank, _seen, out, mx in res:
        assert out == [2.0 * i for i in range(9)]
        assert mx == 5.0
