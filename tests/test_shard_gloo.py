"""N>1 path on CPU: two gloo ranks shard a corpus with no overlap and aggregate counters."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from lepton_amd import shard

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [100, 900, 500, 400, 300, 50, 800]
    mine = shard.shard_indices(len(sizes), world, rank, sizes)
    seeds = shard.weak_seeds(3, rank, 1000)
    dist.barrier()
    agg = shard.aggregate({"jpeg_bytes": sum(sizes[i] for i in mine), "images": len(mine), "step_s_max": 1.0 + rank})
    q.put((rank, mine, seeds, agg))
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    (r0, m0, s0, a0), (r1, m1, s1, a1) = res
    assert sorted(m0 + m1) == list(range(7)) and not set(m0) & set(m1)
    assert not set(s0) & set(s1)
    assert a0 == a1
    assert a0["jpeg_bytes"] == 3050 and a0["images"] == 7 and a0["step_s_max"] == 2.0
    load0 = sum([100, 900, 500, 400, 300, 50, 800][i] for i in m0)
    assert abs(load0 - 1525) <= 400   # greedy balance


def test_round_robin_without_sizes():
    sys.path.insert(0, ROOT)
    from lepton_amd import shard

    assert shard.shard_indices(10, 4, 1) == [1, 5, 9]
    assert sum(len(shard.shard_indices(10, 4, r)) for r in range(4)) == 10
