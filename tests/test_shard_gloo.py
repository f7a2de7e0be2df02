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


def _skew_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import hashlib

    import torch.distributed as dist

    import oracle_binding as ob
    from lepton_amd import corpus, shard
    from lepton_amd.codec import JpegImage

    dist.init_process_group("gloo", rank=rank, world_size=world)
    # BASELINE.json configs[3] in miniature: a mixed corpus (sizes a factor 30 apart), every rank derives the same list and
    # takes its share by JPEG bytes -- no data-path collective, only the barrier and the counter all-reduce
    shapes = [(96, 64), (640, 480), (160, 120), (320, 240), (512, 384), (64, 64), (800, 600), (256, 192), (128, 96), (400, 300), (720, 480), (200, 152)]
    jpgs = [corpus.synth_jpeg(w, h, 7000 + i, skew=2.0 if i % 3 == 0 else 0.0) for i, (w, h) in enumerate(shapes)]
    sizes = [len(j) for j in jpgs]
    mine = shard.shard_indices(len(jpgs), world, rank, sizes)
    done = {}
    for i in mine:   # the rank's work: the hot path on its own images (the CPU oracle stands in for the GPU here)
        img = JpegImage(jpgs[i])
        segs = img.plan()
        streams, _ = ob.oracle_encode(img.desc, segs)
        done[i] = hashlib.md5(img.write_lep(streams)).hexdigest()
    dist.barrier()
    agg = shard.aggregate({"jpeg_bytes": sum(sizes[i] for i in mine), "images": len(mine), "elapsed_max": 1.0 + 0.5 * rank})
    q.put((rank, mine, done, agg, sizes))
    dist.destroy_process_group()


def test_two_ranks_shard_a_size_skewed_corpus():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_skew_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    (_, m0, d0, a0, sizes), (_, m1, d1, a1, sizes1) = res
    assert sizes == sizes1                                        # same corpus on both ranks
    assert sorted(m0 + m1) == list(range(len(sizes))) and not set(m0) & set(m1)
    assert set(d0) == set(m0) and set(d1) == set(m1)              # every image coded exactly once
    assert a0 == a1 and a0["jpeg_bytes"] == sum(sizes) and a0["images"] == len(sizes) and a0["elapsed_max"] == 1.5
    loads = [sum(sizes[i] for i in m) for m in (m0, m1)]
    assert max(loads) / (sum(loads) / 2) < 1.15, loads           # by-bytes balance despite a 30x spread of file sizes
    assert max(sizes) / min(sizes) > 20
