"""Multi-GPU sharding of a parameter batch.

The path shards by independent parameter points: row i of the table goes to rank
i mod world (INTERLEAVED, because per-point cost correlates with s and theta; a
contiguous split of a sorted table would load one GPU with all the expensive
points).  There is no exchange during compute; the only collective is the gather
of the per-rank output tables to rank 0 (RCCL over xGMI when the tensors are on
GPUs; the same code runs on gloo for the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n, rank, world):
    """Global row indices owned by `rank`."""
    return np.arange(rank, n, world)


def shard_sizes(n, world):
    return [len(range(r, n, world)) for r in range(world)]


def gather_table(local_out, n, rank, world, dst=0, force=False):
    """Gather the [n_local, 8] per-rank tables to `dst` and undo the interleave.
    Returns the [n, 8] table on dst, None elsewhere.  force=True runs the collective in a world of one as well (the
    RCCL gather of device tensors exercised on a 1-GPU box: tests/test_gpu_boundary.py)."""
    if world == 1 and not force:
        return local_out
    sizes = shard_sizes(n, world)
    nmax = max(sizes)
    pad = torch.empty((nmax, 8), dtype=local_out.dtype, device=local_out.device)
    pad[: local_out.shape[0]] = local_out
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=dst)
        table = torch.empty((n, 8), dtype=local_out.dtype, device=local_out.device)
        for r in range(world):
            table[r::world] = bufs[r][: sizes[r]]
        return table
    dist.gather(pad, None, dst=dst)
    return None
