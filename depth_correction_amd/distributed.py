"""Multi-GPU sharding of the path: sequences are independent units (neighbourhoods never cross
sequences, train.py:166-175; batch_loss concatenates per-sequence pointwise losses, loss.py:205-213), so every
rank owns whole sequences and the only exchange per iteration is ONE all-reduce (RCCL over xGMI on the GPUs,
gloo in the CPU tests) of the packed vector [sum of pointwise losses, masked-point count, dL/dw ...].
The message is a few dozen bytes: latency-bound, no bucketing or overlap to tune.
"""
from __future__ import annotations

import torch

__all__ = ['shard_sequences', 'all_reduce_sum', 'world_info']


def world_info(group=None):
    """(rank, world size) of the default / given process group, (0, 1) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_sequences(n_sequences, rank, world):
    """Indices of the sequences rank ``rank`` owns: round robin, so sequence q lives on GPU q mod world."""
    assert 0 <= rank < world
    return list(range(rank, n_sequences, world))


def all_reduce_sum(vec, group=None):
    """In-place sum of ``vec`` over the ranks (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec
