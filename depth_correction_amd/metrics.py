"""Map-accuracy metric of the reference (metrics.py:55-125, used by scripts/model_poses_learning:142-146): the
one-directional chamfer distance = mean distance from every point of ``x`` to its nearest neighbour in ``y``.
The nearest-neighbour search is the GPU grid search (dc_knn_build with k = 1, fp64 distances, bit-exact ordering)
instead of pytorch3d's ``knn_points``; batches are lists / a leading dimension of clouds."""
from __future__ import annotations

import torch

from . import ops

__all__ = ['chamfer_distance']


def _as_batch(x):
    if isinstance(x, (list, tuple)):
        return list(x)
    assert isinstance(x, torch.Tensor)
    return [x] if x.dim() == 2 else list(x)


def chamfer_distance(x, y, x_lengths=None, y_lengths=None, apply_point_reduction=True, batch_reduction='mean',
                     point_reduction='mean'):
    """Distances from the points of ``x`` to the cloud ``y`` ([P,3] / [N,P,3] tensors or lists of [P_i,3])."""
    xs, ys = _as_batch(x), _as_batch(y)
    if len(xs) != len(ys):
        raise ValueError('y does not have the correct shape.')
    per_cloud = []
    for b, (a, c) in enumerate(zip(xs, ys)):
        if x_lengths is not None:
            a = a[:int(x_lengths[b])]
        if y_lengths is not None:
            c = c[:int(y_lengths[b])]
        dist, _ = ops.knn(c.detach().contiguous(), 1, query=a.detach().to(c.dtype).contiguous())
        per_cloud.append(dist[:, 0].to(a.dtype))
    if not apply_point_reduction:
        return per_cloud[0] if isinstance(x, torch.Tensor) and x.dim() == 2 else per_cloud
    red = torch.stack([d.sum() / (len(d) if point_reduction == 'mean' else 1) for d in per_cloud])
    if batch_reduction is None:
        return red
    return red.sum() / (len(red) if batch_reduction == 'mean' else 1)
