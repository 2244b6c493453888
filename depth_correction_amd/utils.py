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

__all__ = ['covs', 'trace', 'timing', 'rotation_angle', 'translation_norm', 'transform_inv', 'delta_transform', 'hashable']


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


# ---- pose helpers the caller scripts report with (scripts/train_demo:10; utils.py:174-205), numpy 4x4 matrices --------------
def rotation_angle(T):
    """Angle of the rotation part, from its trace (clipped into arccos' domain)."""
    import numpy as np
    c = (np.trace(np.asarray(T)[:-1, :-1]) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0)))


def translation_norm(T):
    import numpy as np
    return float(np.linalg.norm(np.asarray(T)[:-1, -1]))


def transform_inv(T):
    """Inverse of a rigid transform [R t; 0 1] as [R^T  -R^T t; 0 1]."""
    import numpy as np
    T = np.asarray(T)
    out = np.eye(T.shape[0])
    R = T[:-1, :-1]
    out[:-1, :-1] = R.T
    out[:-1, -1:] = -R.T @ T[:-1, -1:]
    return out


def delta_transform(T_0, T_1):
    """D with T_1 = T_0 D (a linear solve, as the reference does it: no orthonormality assumed)."""
    import numpy as np
    return np.linalg.solve(T_0, T_1)


def hashable(obj):
    """Nested lists / dicts / slices / arrays as tuples, so that they can seed a generator (utils.py:67-76)."""
    import numpy as np
    if isinstance(obj, (list, tuple)):
        return tuple(hashable(o) for o in obj)
    if isinstance(obj, dict):
        return hashable(sorted(obj.items()))
    if isinstance(obj, slice):
        return obj.start, obj.stop, obj.step
    if isinstance(obj, np.ndarray):
        return hashable(obj.tolist())
    return obj
