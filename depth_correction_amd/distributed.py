"""Multi-GPU sharding of the path: sequences are independent units (neighbourhoods never cross
sequences, train.py:166-175; batch_loss concatenates per-sequence pointwise losses, loss.py:205-213), so every
rank owns whole sequences and the only exchange per iteration is ONE all-reduce (RCCL over xGMI on the GPUs,
gloo in the CPU tests) of the packed vector [sum of pointwise losses, masked-point count, dL/dw ...].
The message is a few dozen bytes: latency-bound, no bucketing or overlap to tune.
"""
from __future__ import annotations

import torch

__all__ = ['shard_sequences', 'all_reduce_sum', 'world_info', 'GradReducer', 'gather_objects']


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
    """In-place sum of ``vec`` over the ranks (no-op for a single process; with DC_FORCE_DIST=1 in the environment a
    one-rank group still issues the collective, which is how the RCCL path is exercised on a one-GPU box)."""
    import os
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get('DC_FORCE_DIST') == '1'):
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec


def gather_objects(obj, group=None, device=None):
    """List with every rank's ``obj`` (rank order) on every rank; ``[obj]`` for a single process.  Checkpoint-time
    only (pose corrections of the sequences other ranks own): small pickled tensors, not on the iteration path.
    Under RCCL the pickled bytes are staged on torch's CURRENT device, so ``device`` (this rank's GPU) is made current
    for the call -- a launcher that passes cfg.device = 'cuda:<local rank>' without torch.cuda.set_device would otherwise
    stage every rank's buffer on cuda:0."""
    import torch.distributed as dist
    rank, world = world_info(group)
    if world == 1:
        return [obj]
    out = [None] * world
    dev = torch.device(device) if device is not None else None
    if dev is not None and dev.type == 'cuda':
        with torch.cuda.device(dev):
            dist.all_gather_object(out, obj, group=group)
    else:
        dist.all_gather_object(out, obj, group=group)
    return out


class GradReducer(object):
    """The ONE collective of a training iteration when sequences are sharded over ranks.

    Every rank evaluates the sequences it owns and back-propagates its WEIGHTED local loss ``weight * loss`` (the sum
    of the pointwise losses of its sequences for the min-eigenvalue / trace loss, loss.py:205-213; the sum of its
    sequences' ICP losses for icp_loss, loss.py:373-403).  ``reduce`` then packs

        [ weight * loss,  weight,  grads of the SHARED parameters ... ]

    (shared = the model's ``w`` / ``exponent`` and, for PoseCorrection.common, the one 6-vector all sequences share,
    eval.py:48-53) into one fp64 vector, all-reduces it (RCCL over xGMI; gloo in the CPU tests; a few dozen bytes, so
    latency-bound), and leaves on every rank: the global mean loss, ``grad / total weight`` in every shared
    parameter's ``.grad`` -- identical on all ranks, so identical optimiser steps keep the replicas in lock step --
    and the LOCAL parameters' gradients (per-sequence / per-pose corrections, which only their owner holds and
    optimises) scaled by the same ``1 / total weight``."""

    def __init__(self, shared_params, local_params=(), group=None):
        self.shared = [p for p in shared_params if p is not None]
        self.local = [p for p in local_params if p is not None and all(p is not q for q in self.shared)]
        self.group = group
        self._buf = None

    def _buffer(self, device):
        n = 2 + sum(p.numel() for p in self.shared)
        if self._buf is None or self._buf.device != device or self._buf.numel() != n:
            self._buf = torch.zeros((n,), dtype=torch.float64, device=device)
        return self._buf

    def reduce(self, weighted_loss, weight, with_grads=True):
        """weighted_loss: 0-dim tensor = weight * local mean loss (already back-propagated when ``with_grads``);
        weight: float.  Returns (global mean loss as a detached fp64 0-dim tensor, total weight tensor)."""
        device = weighted_loss.device if isinstance(weighted_loss, torch.Tensor) else (
            self.shared[0].device if self.shared else torch.device('cpu'))
        buf = self._buffer(device)
        buf.zero_()
        buf[0] = weighted_loss.detach() if isinstance(weighted_loss, torch.Tensor) else float(weighted_loss)
        buf[1] = float(weight)
        if with_grads:
            at = 2
            for p in self.shared:
                if p.grad is not None:
                    buf[at:at + p.numel()] = p.grad.detach().reshape(-1).to(device=device, dtype=torch.float64)
                at += p.numel()
        all_reduce_sum(buf, self.group)
        total = buf[1]
        if with_grads:
            at = 2
            for p in self.shared:
                g = (buf[at:at + p.numel()] / total).reshape(p.shape).to(device=p.device, dtype=p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                at += p.numel()
            for p in self.local:
                if p.grad is not None:
                    p.grad.div_(total.to(device=p.device, dtype=p.dtype))
        return (buf[0] / total).clone(), total.clone()
