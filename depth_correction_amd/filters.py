"""Cloud filters with the reference's signatures (filters.py:24-309): boolean masks or sliced clouds.

Eigenvalue / eigenvalue-ratio / valid-neighbour bounds run as HIP mask kernels (dc_mask_bounds, dc_valid_count)
when the cloud lives on the GPU; ``within_bounds`` on arbitrary tensors is the reference's comparison in torch.
Depth and voxel-grid pre-filters feed the path's inputs: ``filter_depth`` is a tensor comparison,
``filter_grid`` keeps the reference's host algorithm (dict over voxel keys with a seeded shuffle) because which
point survives per voxel is defined by numpy's generator.
"""
from __future__ import annotations

import numpy as np
import torch
from numpy.lib.recfunctions import structured_to_unstructured

from . import ops
from .autograd import NeighborhoodGraph
from .depth_cloud import DepthCloud

__all__ = ['filter_depth', 'filter_eigenvalue', 'filter_eigenvalue_ratio', 'filter_eigenvalue_ratios',
           'filter_eigenvalues', 'filter_grid', 'filter_shadow_points', 'filter_valid_neighbors', 'shadow_points_mask',
           'within_bounds']

default_rng = np.random.default_rng(135)


def _report(keep, lo, name, hi):
    print('%.3f = %i / %i points kept (%.3g <= %s <= %.3g).'
          % (keep.double().mean(), keep.sum(), keep.numel(), lo if lo is not None else float('nan'), name,
             hi if hi is not None else float('nan')))


def within_bounds(x, min=None, max=None, bounds=None, log_variable=None):
    """Mask of min <= x <= max (inclusive); None / infinite bounds are inactive; NaN fails active bounds."""
    x = x if isinstance(x, torch.Tensor) else torch.tensor(x)
    if bounds:
        assert min is None and max is None
        min, max = bounds
    keep = torch.ones((x.numel(),), dtype=torch.bool, device=x.device)
    if min is not None and min > -float('inf'):
        keep = keep & (x.flatten() >= min)
    if max is not None and max < float('inf'):
        keep = keep & (x.flatten() <= max)
    if log_variable is not None:
        _report(keep, min, log_variable, max)
    return keep


def _gpu_bounds(num, num_index, den, den_index, lo, hi):
    mask = torch.ones((num.shape[0],), dtype=torch.bool, device=num.device)
    return ops.mask_bounds(mask, num.detach().contiguous(), num_index, None if den is None else den.detach().contiguous(),
                           den_index, lo, hi)


def _finish(cloud, keep, only_mask):
    return keep if only_mask else cloud[keep]


def filter_depth(cloud, min=None, max=None, only_mask=False, log=False):
    assert isinstance(cloud, (DepthCloud, np.ndarray))
    if isinstance(cloud, DepthCloud):
        depth = cloud.depth
    else:
        x = structured_to_unstructured(cloud[['x', 'y', 'z']]) if cloud.dtype.names else cloud
        if cloud.dtype.names and 'vp_x' in cloud.dtype.names:
            x = x - structured_to_unstructured(cloud[['vp_x', 'vp_y', 'vp_z']])
        depth = torch.as_tensor(np.linalg.norm(x, axis=1))
    keep = within_bounds(depth, min=min, max=max, log_variable='depth' if log else None)
    if only_mask:
        return keep
    return cloud[keep] if isinstance(cloud, DepthCloud) else cloud[keep.numpy()]


def filter_grid(cloud, grid_res, only_mask=False, keep='random', preserve_order=False, log=False, rng=default_rng):
    """One point per voxel of edge ``grid_res``.  The survivor of a voxel is the LAST point of the (optionally
    shuffled or reversed) sequence falling into it, exactly as the reference's dict construction yields."""
    assert isinstance(cloud, (DepthCloud, np.ndarray, torch.Tensor))
    assert isinstance(grid_res, float) and grid_res > 0.0 and keep in ('first', 'random', 'last')
    pts = cloud.get_points() if isinstance(cloud, DepthCloud) else cloud
    if isinstance(pts, np.ndarray) and torch.cuda.is_available():
        # host arrays (the datasets' structured clouds, preproc.filtered_cloud): filter on the GPU, index on the host
        xyz = structured_to_unstructured(pts[['x', 'y', 'z']]) if pts.dtype.names else pts
        if xyz.ndim == 2 and xyz.shape[1] == 3 and xyz.dtype in (np.float32, np.float64):
            pts = torch.as_tensor(np.ascontiguousarray(xyz), device=torch.device('cuda', torch.cuda.current_device()))
    if isinstance(pts, torch.Tensor) and pts.is_cuda and pts.dim() == 2 and pts.shape[1] == 3:
        # GPU path (dc_voxel_filter): only the processing sequence is made on the host, because numpy's generator
        # defines which point of a voxel survives a 'random' filter
        n = pts.shape[0]
        seq = None
        if keep == 'first':
            seq = torch.arange(n - 1, -1, -1, dtype=torch.int32, device=pts.device)
        elif keep == 'random':
            perm = np.arange(n)
            rng.shuffle(perm)                       # same permutation as shuffling the reference's index list
            seq = torch.as_tensor(perm.astype(np.int32), device=pts.device)
        ind = ops.voxel_filter(pts.detach().contiguous(), grid_res, seq, preserve_order)
        if ind is not None:
            if log:
                print('%.3f = %i / %i points kept (grid res. %.3f m).' % (len(ind) / max(n, 1), len(ind), n, grid_res))
            if only_mask:
                return ind.tolist()
            return cloud[ind.cpu().numpy()] if isinstance(cloud, np.ndarray) else cloud[ind]
        if keep == 'random':
            raise RuntimeError('voxel range too large for the GPU key and the generator was already advanced')
    # host path: CPU-only machines preparing datasets, or a voxel range beyond the GPU's 3 x 21-bit key (the
    # reference's own dict construction; data preparation, not part of the hot path)
    if isinstance(cloud, DepthCloud):
        x = cloud.get_points().detach().cpu().numpy()
    elif isinstance(cloud, np.ndarray):
        x = structured_to_unstructured(cloud[['x', 'y', 'z']]) if cloud.dtype.names else cloud
    else:
        x = cloud.detach().cpu().numpy()
    voxels = np.floor(x / grid_res).astype(int)
    order = np.arange(len(voxels))
    if keep == 'first':
        order = order[::-1]
    elif keep == 'random':
        order = list(range(len(voxels)))
        rng.shuffle(order)
        order = np.asarray(order, dtype=np.int64)
    survivor = {}
    for i, key in zip(order.tolist(), map(tuple, voxels[order].tolist())):
        survivor[key] = i
    ind = sorted(survivor.values()) if preserve_order else list(survivor.values())
    if log:
        print('%.3f = %i / %i points kept (grid res. %.3f m).' % (len(ind) / len(voxels), len(ind), len(voxels), grid_res))
    return ind if only_mask else cloud[ind]


def filter_valid_neighbors(cloud, min=None, only_mask=False, log=False):
    assert isinstance(cloud, DepthCloud) and cloud.neighbors is not None
    if cloud.neighbors.is_cuda:
        cnt = ops.valid_count(cloud.graph().nbr)
    else:
        cnt = cloud.valid_neighbor_mask().sum(dim=-1)
    keep = within_bounds(cnt, min=min, log_variable='valid neighbors' if log else None)
    return _finish(cloud, keep, only_mask)


def filter_eigenvalue(cloud, eigenvalue=0, min=None, max=None, only_mask=False, log=False):
    with torch.no_grad():
        ev = cloud.eigvals
        if ev.is_cuda:
            keep = _gpu_bounds(ev, eigenvalue, None, 0, min, max)
            if log:
                _report(keep, min, 'eigenvalue %i' % eigenvalue, max)
        else:
            keep = within_bounds(ev[:, eigenvalue], min=min, max=max, log_variable='eigenvalue %i' % eigenvalue if log else None)
    return _finish(cloud, keep, only_mask)


def filter_eigenvalue_ratio(cloud, eigenvalues=(0, 1), min=None, max=None, only_mask=False, log=False):
    assert cloud.eigvals is not None and len(eigenvalues) == 2 and all(0 <= i <= 2 for i in eigenvalues)
    i, j = eigenvalues
    with torch.no_grad():
        ev = cloud.eigvals
        if ev.is_cuda:
            keep = _gpu_bounds(ev, i, ev, j, min, max)
            if log:
                _report(keep, min, 'eigenvalue %i / eigenvalue %i' % tuple(eigenvalues), max)
        else:
            keep = within_bounds(ev[:, i] / ev[:, j], min=min, max=max,
                                 log_variable='eigenvalue %i / eigenvalue %i' % tuple(eigenvalues) if log else None)
    return _finish(cloud, keep, only_mask)


def _all_of(cloud, bounds, one, what, only_mask, log):
    mask = None
    for b in (bounds or []):
        m = one(b)
        mask = m if mask is None else mask & m
    if mask is None:
        mask = torch.ones((cloud.size(),), dtype=torch.bool, device=cloud.device())
    if log:
        print('%.3f = %i / %i points kept (%s within bounds).' % (mask.double().mean(), mask.sum(), mask.numel(), what))
    return _finish(cloud, mask, only_mask)


def filter_eigenvalues(cloud: DepthCloud, bounds: list, only_mask: bool = False, log: bool = False):
    return _all_of(cloud, bounds, lambda b: filter_eigenvalue(cloud, int(b[0]), min=b[1], max=b[2], only_mask=True, log=log),
                   'eigenvalues', only_mask, log)


def filter_eigenvalue_ratios(cloud: DepthCloud, bounds: list, only_mask: bool = False, log: bool = False):
    return _all_of(cloud, bounds, lambda b: filter_eigenvalue_ratio(cloud, (int(b[0]), int(b[1])), min=b[2], max=b[3],
                                                                   only_mask=True, log=log),
                   'eigenvalue ratios', only_mask, log)


def _shadow_bounds(angle_bounds):
    lo = 0.0 if (angle_bounds[0] is None or not (angle_bounds[0] >= 0.0)) else float(angle_bounds[0])
    hi = torch.pi if (angle_bounds[1] is None or not (angle_bounds[1] <= torch.pi)) else float(angle_bounds[1])
    # the reference holds the bounds in a float32 tensor (torch.as_tensor of Python floats) and fills the angles of
    # missing neighbours with its mean
    lo, hi = float(np.float32(lo)), float(np.float32(hi))
    fill = float((np.float32(lo) + np.float32(hi)) / np.float32(2.0))
    return lo, hi, fill


def shadow_points_mask(cloud: DepthCloud, neighborhood_angle: float, angle_bounds: list, log: bool = False):
    """``update_dir_neighbors(angle=neighborhood_angle)`` + ``filter_shadow_points(only_mask=True)`` in one walk over the
    direction grid (dc_shadow_filter): the table of direction neighbours, which the reference's callers drop right after the
    filter (preproc.py:44-47: ``cloud[mask]`` keeps the per-point fields only), is never written.  Device clouds only."""
    from .nearest_neighbors import ball_angle_to_distance
    lo, hi, _ = _shadow_bounds(angle_bounds)
    r = float(ball_angle_to_distance(torch.as_tensor(neighborhood_angle)))
    x = cloud.get_points()
    with torch.no_grad():
        mask = ops.shadow_filter(x.detach().contiguous(), cloud.vps.detach().to(x.dtype).contiguous(),
                                 cloud.dirs.detach().contiguous(), r, lo, hi)
    if log:
        print('%.3f = %i / %i points kept (shadow points removed).' % (mask.double().mean(), mask.sum(), mask.numel()))
    return mask


def filter_shadow_points(cloud: DepthCloud, angle_bounds: list, only_mask: bool = False, log: bool = False):
    """Scan-shadow filter (filters.py:257-309): keep a point when the angles between the ray back to its viewpoint
    and the vectors to its direction-neighbours all lie within ``angle_bounds``.  Needs ``update_dir_neighbors``
    (radius search on the unit directions, on the GPU).  ``only_mask=True`` returns the mask (the reference returns
    the flag itself there, an obvious slip)."""
    assert cloud.vps is not None and cloud.dir_neighbors is not None
    lo, hi, fill = _shadow_bounds(angle_bounds)
    x = cloud.get_points()
    if x.is_cuda:
        # one kernel over the direction-neighbour rows (dc_shadow_mask): no [N, K, 3] tensors
        with torch.no_grad():
            mask = ops.shadow_mask(x.detach().contiguous(), cloud.vps.detach().to(x.dtype).contiguous(),
                                   NeighborhoodGraph.of(cloud.dir_neighbors).nbr, lo, hi, fill)
    else:
        # host tensors (data preparation on a CPU-only machine): the reference's tensor expressions
        to_vp = (cloud.vps.expand_as(x) - x).unsqueeze(dim=1)
        to_nb = x[cloud.dir_neighbors] - x.unsqueeze(dim=1)
        ang = torch.acos(torch.nn.functional.cosine_similarity(to_vp, to_nb, dim=-1))
        ang = torch.where(cloud.dir_neighbor_weights != 1.0, torch.full_like(ang, fill), ang)
        mask = (ang.amin(dim=-1) >= lo) & (ang.amax(dim=-1) <= hi)
    if log:
        print('%.3f = %i / %i points kept (shadow points removed).' % (mask.double().mean(), mask.sum(), mask.numel()))
    return mask if only_mask else cloud[mask]
