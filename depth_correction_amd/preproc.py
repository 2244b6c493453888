"""Per-iteration glue with the reference's signatures (preproc.py:25-243): filtered / local feature clouds, the
global cloud, neighbourhood establishment, per-iteration features and the global mask -- for ball neighbourhoods
(NeighborhoodType.plane needs RANSAC plane segmentation from PCL / open3d and is out of scope).
"""
from __future__ import annotations

import numpy as np
import torch

from .config import Config, NeighborhoodType
from .depth_cloud import DepthCloud
from .filters import (filter_depth, filter_eigenvalue_ratios, filter_eigenvalues, filter_grid, filter_shadow_points, shadow_points_mask,
                      filter_valid_neighbors, within_bounds)
from .transform import xyz_axis_angle_to_matrix

__all__ = ['compute_neighborhood_features', 'establish_neighborhoods', 'filtered_cloud', 'global_cloud',
           'global_cloud_mask', 'local_feature_cloud', 'offset_cloud']


def _ball_only(cfg):
    if cfg.nn_type != NeighborhoodType.ball:
        raise NotImplementedError('only ball neighbourhoods are implemented (plane neighbourhoods need PCL / open3d)')


def filtered_cloud(cloud, cfg: Config):
    """Depth + voxel-grid pre-filters (preproc.py:25-32)."""
    if (cfg.min_depth is not None and cfg.min_depth > 0.0) or (cfg.max_depth is not None and cfg.max_depth < float('inf')):
        cloud = filter_depth(cloud, min=cfg.min_depth, max=cfg.max_depth, log=cfg.log_filters)
    if cfg.grid_res > 0.0:
        cloud = filter_grid(cloud, grid_res=cfg.grid_res, keep='random', log=cfg.log_filters,
                            rng=np.random.default_rng(cfg.random_seed))
    return cloud


def _and_mask(cloud, mask):
    if cloud.mask is None:
        cloud.mask = torch.ones((len(cloud),), dtype=torch.bool, device=cloud.device())
    cloud.mask = cloud.mask & mask


def local_feature_cloud(cloud, cfg: Config):
    """Scan -> DepthCloud with neighbours, features and the planarity mask (preproc.py:35-64)."""
    if isinstance(cloud, torch.Tensor):
        # raw rows [N, >=3] already on the device (what the node holds after the upload of a message)
        if cloud.is_cuda and cloud.dim() == 2 and cloud.dtype in (torch.float32, torch.float64) and cfg.shadow_angle_bounds:
            return _with_features(_prefiltered_cloud(cloud.detach(), cfg), cfg, points_current=True)
        return _with_features(DepthCloud.from_points(cloud[:, :3], dtype=cfg.numpy_float_type(), device=cfg.device), cfg)
    if isinstance(cloud, np.ndarray):
        make = DepthCloud.from_structured_array if cloud.dtype.names else DepthCloud.from_points
        cloud = make(cloud, dtype=cfg.numpy_float_type(), device=cfg.device)
    assert isinstance(cloud, DepthCloud)
    if cfg.shadow_angle_bounds:                      # preproc.py:44-47
        if cloud.dirs.is_cuda:
            # the direction-neighbour table would be dropped by cloud[mask] right away: the mask comes from one grid walk
            cloud = cloud[shadow_points_mask(cloud, cfg.shadow_neighborhood_angle, cfg.shadow_angle_bounds, log=cfg.log_filters)]
        else:
            cloud.update_dir_neighbors(angle=cfg.shadow_neighborhood_angle)
            cloud = filter_shadow_points(cloud, cfg.shadow_angle_bounds, log=cfg.log_filters)
    return _with_features(cloud, cfg)


_chords = {}


def _chord(angle):
    from .nearest_neighbors import ball_angle_to_distance
    if angle not in _chords:
        _chords[angle] = float(ball_angle_to_distance(torch.as_tensor(angle)))
    return _chords[angle]


def _prefiltered_cloud(raw, cfg: Config):
    """from_points + shadow filter + cloud[mask] of raw device rows in one native call (dc_scan_prefilter): the first statements of
    local_feature_cloud (preproc.py:36-47) without the host between their launches; the same kernels, the same cloud."""
    from . import ops
    from .filters import _shadow_bounds
    lo, hi, _ = _shadow_bounds(cfg.shadow_angle_bounds)
    # (the reference evaluates the chord in float32 tensor arithmetic, nearest_neighbors.py:13-19: kept, once per configuration value)
    r = _chord(float(cfg.shadow_neighborhood_angle))
    dtype = cfg.torch_float_type()
    n = raw.shape[0]
    vps, dirs, depth, points = ops.scan_prefilter(raw.contiguous(), None, dtype, r, lo, hi)
    if cfg.log_filters:
        print('%.3f = %i / %i points kept (shadow points removed).' % (len(dirs) / max(n, 1), len(dirs), n))
    return DepthCloud(vps, dirs, depth, points=points)


def _with_features(cloud, cfg: Config, points_current=False):
    """Neighbourhoods, features and the planarity mask (preproc.py:48-63).  ``points_current``: cloud.points already are
    vps + depth * dirs of its fields (dc_scan_prefilter wrote them): update_all's first statement is skipped."""
    if points_current and cloud.points is not None:
        cloud.update_neighbors(k=cfg.nn_k, r=cfg.nn_r, _weights=False)
        cloud.update_features(scale=None)
    else:
        cloud.update_all(k=cfg.nn_k, r=cfg.nn_r)
    if cloud.eigvals.is_cuda and not cfg.log_filters and (cfg.eigenvalue_bounds or cfg.eigenvalue_ratio_bounds):
        # the same masks without the ones / and passes between them: every bound in one kernel (dc_mask_bounds_multi), or one kernel per
        # bound ANDed in place (dc_mask_bounds) beyond eight
        from . import ops
        bounds = [(int(e), None, lo, hi) for e, lo, hi in (cfg.eigenvalue_bounds or [])] + \
                 [(int(i), int(j), lo, hi) for i, j, lo, hi in (cfg.eigenvalue_ratio_bounds or [])]
        if len(bounds) <= 8:
            # every bound in one pass over the eigenvalues (dc_mask_bounds_multi), written or ANDed into the mask
            cloud.mask = ops.mask_bounds_all(cloud.eigvals.detach().contiguous(), bounds,
                                             mask=None if cloud.mask is None else cloud.mask.clone())
            return cloud
        mask = torch.ones((len(cloud),), dtype=torch.bool, device=cloud.device()) if cloud.mask is None else cloud.mask.clone()
        ev = cloud.eigvals.detach().contiguous()
        for i, j, lo, hi in bounds:
            ops.mask_bounds(mask, ev, i, None if j is None else ev, 0 if j is None else j, lo, hi)
        cloud.mask = mask
        return cloud
    if cfg.eigenvalue_bounds:
        _and_mask(cloud, filter_eigenvalues(cloud, cfg.eigenvalue_bounds, only_mask=True, log=cfg.log_filters))
    if cfg.eigenvalue_ratio_bounds:
        _and_mask(cloud, filter_eigenvalue_ratios(cloud, cfg.eigenvalue_ratio_bounds, only_mask=True, log=cfg.log_filters))
    return cloud


def offset_cloud(clouds, model):
    corrected = [model(c) if model is not None else c for c in clouds]
    return DepthCloud.concatenate(corrected, fields=DepthCloud.source_fields + ['eigvals'])


def global_cloud(clouds=None, model=None, poses=None, pose_corrections=None, dataset=None):
    """Corrected scans moved to the world frame and concatenated (preproc.py:80-119)."""
    if dataset is not None:
        assert clouds is None and poses is None
        clouds, poses = zip(*dataset)
        clouds = [DepthCloud.from_structured_array(c, dtype=np.float64) for c in clouds]
        poses = torch.as_tensor(np.array(poses))
    assert clouds is not None and poses is not None
    if pose_corrections is not None:
        if pose_corrections.shape[-1] == 6:
            pose_corrections = xyz_axis_angle_to_matrix(pose_corrections)
        poses = poses @ pose_corrections
    moved = []
    for cloud, pose in zip(clouds, poses):
        if model is not None:
            cloud = model(cloud)
        moved.append(cloud.transform(pose))
    return DepthCloud.concatenate(moved, dependent=True)


def global_cloud_mask(cloud: DepthCloud, mask, cfg: Config):
    """AND of the global-cloud filters into ``mask`` (modified in place like the reference, preproc.py:122-164)."""
    if mask is None:
        mask = torch.ones((len(cloud),), dtype=torch.bool, device=cloud.device())
    else:
        print('%.3f = %i / %i points kept (previous filters).' % (mask.double().mean(), mask.sum(), mask.numel()))
    log = cfg.log_filters
    if cfg.min_valid_neighbors:
        mask &= filter_valid_neighbors(cloud, min=cfg.min_valid_neighbors, only_mask=True, log=log)
    if cfg.eigenvalue_bounds:
        mask &= filter_eigenvalues(cloud, bounds=cfg.eigenvalue_bounds, only_mask=True, log=log)
    if cfg.eigenvalue_ratio_bounds:
        mask &= filter_eigenvalue_ratios(cloud, bounds=cfg.eigenvalue_ratio_bounds, only_mask=True, log=log)
    if cfg.dir_dispersion_bounds:
        mask &= within_bounds(cloud.dir_dispersion(), bounds=cfg.dir_dispersion_bounds,
                              log_variable='dir dispersion' if log else None)
    if cfg.vp_dispersion_bounds:
        mask &= within_bounds(cloud.vp_dispersion(), bounds=cfg.vp_dispersion_bounds,
                              log_variable='vp dispersion' if log else None)
    if cfg.vp_dispersion_to_depth2_bounds:
        mask &= within_bounds(cloud.vp_dispersion_to_depth2(), bounds=cfg.vp_dispersion_to_depth2_bounds,
                              log_variable='vp dispersion to depth2' if log else None)
    return mask


def establish_neighborhoods(dataset=None, clouds=None, poses=None, cloud=None, cfg: Config = None):
    """(neighbors, weights) of the global cloud, found once before the optimisation (preproc.py:168-185)."""
    _ball_only(cfg)
    if cloud is None:
        cloud = global_cloud(clouds=clouds, poses=poses, dataset=dataset)
    cloud.update_all(k=cfg.nn_k, r=cfg.nn_r, scale=cfg.nn_scale, keep_neighbors=False)
    return cloud.neighbors, cloud.weights


def compute_neighborhood_features(dataset=None, clouds=None, poses=None, model=None, pose_corrections=None, cloud=None,
                                  neighborhoods=None, cfg: Config = None):
    """Features of the (re-corrected) global cloud on fixed neighbourhoods (preproc.py:195-217)."""
    _ball_only(cfg)
    if neighborhoods is None:
        neighborhoods = establish_neighborhoods(dataset=dataset, cloud=cloud, cfg=cfg)
    if cloud is None:
        cloud = global_cloud(clouds=clouds, model=model, poses=poses, pose_corrections=pose_corrections, dataset=dataset)
    cloud.neighbors, cloud.weights = neighborhoods
    cloud.update_all(scale=cfg.nn_scale, keep_neighbors=True)
    return cloud
