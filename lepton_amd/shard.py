"""Image-level sharding across the GPUs of a node: images are independent, so each rank
codes its own images with NO data-path collective; torch.distributed (RCCL over xGMI on GPUs, gloo in
the CPU tests) is used only for the start/stop barriers and one all_reduce of throughput counters."""
import os


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_indices(n_items, world, rank, sizes=None):
    """Greedy least-loaded assignment of items (sorted by size, largest first) to ranks; deterministic.
    Without sizes: round robin."""
    if sizes is None:
        return list(range(rank, n_items, world))
    order = sorted(range(n_items), key=lambda i: (-sizes[i], i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += sizes[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


def weak_seeds(per_rank, rank, seed0):
    """Weak scaling: every rank codes `per_rank` distinct images; seeds never collide across ranks."""
    return [seed0 + rank * per_rank + i for i in range(per_rank)]


def aggregate(local, backend_device=None):
    """Sum a dict of numeric counters over all ranks, max over '*_max' keys; returns floats on every rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: float(v) for k, v in local.items()}
    keys = sorted(local)
    dev = backend_device or "cpu"
    sums = torch.tensor([float(local[k]) for k in keys if not k.endswith("_max")], dtype=torch.float64, device=dev)
    maxs = torch.tensor([float(local[k]) for k in keys if k.endswith("_max")] or [0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(maxs, op=dist.ReduceOp.MAX)
    out, si, mi = {}, 0, 0
    for k in keys:
        if k.endswith("_max"):
            out[k] = float(maxs[mi]); mi += 1
        else:
            out[k] = float(sums[si]); si += 1
    return out
