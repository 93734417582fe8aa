"""Rank-level plumbing of the sharded workload (one process per GPU, launched by torch.distributed.run).

The path shards by independent units -- pairs of flow fields (BASELINE config 4) or output row bands of
one huge field (config 5) -- so there is no data-path collective.  The only exchange is the one-off RCCL
broadcast of a shared source field; its 128-byte unique id travels over the launcher's process group.
Nothing here touches the GPU: the functions are exercised with gloo on CPU (tests/test_multirank.py).
"""
import numpy as np


def shard(n_items, rank, world):
    """Contiguous block of item indices owned by `rank` (blocks differ by at most one item)."""
    if world < 1 or not 0 <= rank < world or n_items < 0:
        raise ValueError("bad shard request: n_items={}, rank={}, world={}".format(n_items, rank, world))
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def row_band(height, rank, world, align=8):
    """Output rows [start, stop) of rank's band when one field is split over `world` GPUs; band edges are
    multiples of `align` rows (the compose kernel's tile height) except the last."""
    tiles = -(-height // align)
    r = shard(tiles, rank, world)
    return min(r.start * align, height), min(r.stop * align, height)


def local_device(local_rank, n_devices):
    if n_devices <= 0:
        raise RuntimeError("no HIP device visible")
    return local_rank % n_devices


def broadcast_bytes(dist, payload, src=0):
    """Broadcast a fixed-size uint8 array (e.g. the RCCL unique id) from `src` over the process group."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(payload, np.uint8).copy())
    dist.broadcast(t, src)
    return t.numpy()


def max_over_ranks(dist, values):
    """Element-wise maximum of a list of floats over all ranks (timings are reported as the slowest rank's)."""
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]
