"""Small tensor helpers of the reference's utils.py that sit on the hot path: ``covs`` (utils.py:109-149),
``trace`` (:152-154) and the ``timing`` decorator (:54-64).

``covs`` is the generic (any axes, any device) form and is plain torch; DepthCloud never calls it for
neighbourhood covariances -- those come from the fused HIP kernel (dc_features_fwd) without materialising
the ``[N,K,3,3]`` products.
"""
from __future__ import annotations

import functools
import time

import torch

__all__ = ['covs', 'trace', 'timing']


def timing(f):
    @functools.wraps(f)
    def timed(*args, **kwargs):
        t0 = time.time()
        try:
            return f(*args, **kwargs)
        finally:
            print('%s %.6f s' % (f.__name__, time.time() - t0))
    return timed


def covs(x, obs_axis=-2, var_axis=-1, center=True, correction=True, weights=None):
    """Batched (weighted) covariance of samples along ``obs_axis`` over variables along ``var_axis``.

    Normalisation as the reference: divide by ``sum(weights) - 1`` clamped to 1e-6 (``correction``)."""
    assert isinstance(x, torch.Tensor) and obs_axis != var_axis
    assert weights is None or isinstance(weights, torch.Tensor)
    total = weights.sum(dim=obs_axis, keepdim=True) if weights is not None else x.shape[obs_axis]
    if center:
        mean = ((weights * x).sum(dim=obs_axis, keepdim=True) / total) if weights is not None \
            else x.mean(dim=obs_axis, keepdim=True)
        x = x - mean
    other = var_axis + 1 if var_axis >= 0 else var_axis - 1
    outer = x.unsqueeze(var_axis) * x.unsqueeze(other)
    if weights is not None:
        outer = weights.unsqueeze(var_axis) * outer
    if obs_axis < var_axis and obs_axis < 0:
        obs_axis -= 1
    elif obs_axis > var_axis and obs_axis > 0:
        obs_axis += 1
    outer = outer.sum(dim=obs_axis)
    if correction:
        total = total - 1
    if isinstance(total, torch.Tensor) and total.dtype.is_floating_point:
        total = total.clamp(1e-6, None)
    return outer / total


def trace(x, dim1=-2, dim2=-1):
    return x.diagonal(dim1=dim1, dim2=dim2).sum(dim=-1)
