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
    """Batched (weighted) covariance matrices: samples along ``obs_axis``, variables along ``var_axis``
    (signature and normalisation of utils.py:109-149: divide by ``sum(weights) - 1``, clamped to 1e-6 when it is a
    float tensor).  The observation axis disappears and the variable axis becomes two adjacent axes.

    Axes of mixed sign (e.g. obs_axis=-1, var_axis=0) follow the definition here; the reference's axis shift sums over
    the wrong axis in that one combination, which none of its callers use.

    Formed as one batched matrix product  Xc^T diag(w) Xc  on [..., K, V] views instead of a materialised
    ``[..., K, V, V]`` outer-product tensor."""
    assert isinstance(x, torch.Tensor) and (weights is None or isinstance(weights, torch.Tensor))
    nd = x.dim()
    obs, var = obs_axis % nd, var_axis % nd
    assert obs != var
    xs = x.movedim((obs, var), (-2, -1))                                  # [..., K, V]
    if weights is not None:
        ws = torch.broadcast_to(weights, x.shape).movedim((obs, var), (-2, -1))[..., :1]      # [..., K, 1]
        norm = ws.sum(dim=-2, keepdim=True)                               # [..., 1, 1]
    else:
        ws, norm = None, xs.shape[-2]
    if center:
        xs = xs - ((ws * xs).sum(dim=-2, keepdim=True) / norm if ws is not None else xs.mean(dim=-2, keepdim=True))
    left = xs if ws is None else ws * xs
    cov = left.transpose(-1, -2) @ xs                                     # [..., V, V]
    if correction:
        norm = norm - 1
    if isinstance(norm, torch.Tensor) and norm.dtype.is_floating_point:
        norm = norm.clamp(min=1e-6)
    cov = cov / norm
    at = var - (1 if obs < var else 0)                                    # where the variable axis sits without obs
    return cov.movedim((-2, -1), (at, at + 1))


def trace(x, dim1=-2, dim2=-1):
    return x.diagonal(dim1=dim1, dim2=dim2).sum(dim=-1)
