"""Neighbourhood search with the reference's signature (nearest_neighbors.py:22-80), on the GPU.

Same contract as the cKDTree-based original: fp64 distance ordering, ``k`` nearest / ``k`` within ``r`` /
all within ``r``; missing neighbours are -1 (distance inf); results come back on the input device as
int64 indices and float64 distances.
"""
from __future__ import annotations

import torch

from . import ops

__all__ = ['ball_angle_to_distance', 'nearest_neighbors']


def ball_angle_to_distance(angle, radius=1.0):
    """Chord length subtending ``angle`` on a sphere (nearest_neighbors.py:13-19)."""
    assert isinstance(angle, torch.Tensor)
    dist = torch.sqrt(2. * (1. - torch.cos(torch.clamp(angle, 0., torch.pi))))
    return radius * dist if (isinstance(radius, float) or radius != 1.0) else dist


def nearest_neighbors(points, query, k=None, r=None, n_jobs=-1):
    """(dist, ind) of the neighbours of ``query`` rows among ``points`` rows; ``dist`` is None for pure radius search.

    ``n_jobs`` is accepted for signature compatibility (the GPU kernel has no thread knob)."""
    assert isinstance(points, torch.Tensor) and isinstance(query, torch.Tensor)
    assert k or r
    if not points.is_cuda:
        raise RuntimeError('nearest_neighbors: tensors must be on the GPU (depth_correction_amd has no CPU path)')
    pts = points.detach().reshape(-1, points.shape[-1]).contiguous()
    qry = query.detach().reshape(-1, points.shape[-1]).contiguous()
    if pts.shape[1] != 3:
        raise ValueError('nearest_neighbors: the HIP builder works on 3-D points')
    if not pts.dtype.is_floating_point:
        pts, qry = pts.double(), qry.double()
    same = qry.data_ptr() == pts.data_ptr() and qry.shape == pts.shape
    if k:
        dist, ind, ind64 = ops.knn(pts, int(k), r=r, query=None if same else qry.to(pts.dtype), want_index64=True)
        return dist, _as_reference_index(ind, ind64)
    return None, _as_reference_index(ops.radius_neighbors(pts, float(r), query=None if same else qry.to(pts.dtype)))


def _as_reference_index(ind32, ind64=None):
    """int64 copy of the builder's int32 table (the reference's index dtype, nearest_neighbors.py:78) that REMEMBERS the int32
    table it came from: the kernels behind DepthCloud take int32, and converting back cost a pass per consumer."""
    from .autograd import NeighborhoodGraph
    ind = ind32.long() if ind64 is None else ind64        # (the k-NN kernels write both tables)
    ind._dc_graph = NeighborhoodGraph(ind, nbr=ind32)
    return ind
