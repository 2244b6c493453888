"""CPU oracle: a plain torch/numpy restatement of depth_correction's map-consistency hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product package (``depth_correction_amd``)
never imports anything under ``oracle/`` and fails loudly when its HIP library is missing.

Every function restates the reference algorithm *as written* (same operation order, same
materialised intermediates, autograd for the backward) and cites the reference file:line it follows
(paths relative to the reference's ``src/depth_correction/``).  Parity is pinned by
``tests/golden/*.npz`` generated from the live reference by ``oracle/gen_golden.py`` (the generator
asserts restatement == reference before writing) and re-checked by ``tests/test_oracle_golden.py``.

Third-party arithmetic the reference delegates to and that is not part of its source tree:
  * scipy.spatial.cKDTree (reference pins scipy==1.7.2, python_requirements.txt:12) -- called here the
    same way (``knn_ckdtree``), and cross-checked by an independent brute-force fp64 restatement
    (``knn_bruteforce``) that defines the ordering contract (ascending fp64 squared distance,
    s = ((dx*dx + dy*dy) + dz*dz), no FMA contraction; exact ties are outside the contract).
  * LAPACK syevd through torch.linalg.eigh (reference pins torch==1.10.0) -- called the same way.
  * pytorch3d.transforms.axis_angle_to_matrix (``@stable``, not installed, source absent) -- the
    published algorithm (axis-angle -> quaternion with the small-angle series -> rotation matrix) is
    restated in ``axis_angle_to_matrix``; parity of that single function is UNPINNED (no fixture of
    the reference exists for it), everything downstream of it is pinned with it substituted.
"""
from __future__ import annotations

import numpy as np
import torch

__all__ = [
    'axis_angle_to_matrix', 'closed_form_backward', 'covs', 'dispersion', 'eval_sequence', 'features', 'point_to_point',
    'shadow_mask',
    'global_mask', 'knn_bruteforce', 'knn_ckdtree', 'local_mask', 'model_bias', 'model_apply', 'pointwise_loss',
    'point_to_plane', 'points_from', 'radius_bruteforce', 'radius_ckdtree', 'reduce_loss', 'trace',
    'transform_cloud', 'within_bounds', 'xyz_axis_angle_to_matrix', 'nn1_correspondences',
]


# ----------------------------------------------------------------------------------------------
# K4 / K4r: neighbourhood builder.  nearest_neighbors.py:22-80, depth_cloud.py:210-215
# ----------------------------------------------------------------------------------------------
def knn_ckdtree(points, k, r=None, query=None):
    """k-NN exactly as nearest_neighbors.py:46-49: cKDTree (fp64, leafsize 16) self query.

    Returns (dist f64 [N,k], ind i64 [N,k]); missing neighbours (k within r) -> ind = -1, dist = inf.
    """
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float64)
    q = p if query is None else np.asarray(query, dtype=np.float64)
    index = cKDTree(p)
    dist, ind = index.query(q, k, workers=-1, **({'distance_upper_bound': r} if r else {}))
    ind = ind.astype(np.int64)
    ind[ind == index.n] = -1
    return dist, ind


def radius_ckdtree(points, r):
    """Radius search as nearest_neighbors.py:50-51,69-73: ragged lists padded with -1 to [N,Kmax]."""
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float64)
    index = cKDTree(p)
    lists = index.query_ball_point(p, r, workers=-1)
    n = max(len(x) for x in lists)
    ind = np.full((len(lists), n), -1, dtype=np.int64)
    for i, x in enumerate(lists):
        ind[i, :len(x)] = x
    return ind


def _sqdist_f64(p, q):
    # cKDTree's sqeuclidean_distance_double for n=3: s = 0; s += d*d per axis, in order, no FMA.
    d = p[None, :, :] - q[:, None, :]
    s = d[..., 0] * d[..., 0]
    s = s + d[..., 1] * d[..., 1]
    s = s + d[..., 2] * d[..., 2]
    return s


def knn_bruteforce(points, k, r=None, chunk=2048):
    """Independent fp64 brute force stating the ordering contract of K4 (ascending (d2, index))."""
    p = np.asarray(points, dtype=np.float64)
    n = p.shape[0]
    kk = min(k, n)
    ind = np.full((n, k), -1, dtype=np.int64)
    dist = np.full((n, k), np.inf, dtype=np.float64)
    for s in range(0, n, chunk):
        d2 = _sqdist_f64(p, p[s:s + chunk])
        order = np.lexsort((np.broadcast_to(np.arange(n), d2.shape), d2), axis=-1)[:, :kk]
        dsel = np.take_along_axis(d2, order, axis=-1)
        isel = order.astype(np.int64)
        if r:
            # cKDTree.query keeps strictly d < distance_upper_bound (compared on squared values).
            bad = ~(dsel < r * r)
            isel = np.where(bad, -1, isel)
            dsel = np.where(bad, np.inf, dsel)
        ind[s:s + chunk, :kk] = isel
        dist[s:s + chunk, :kk] = np.sqrt(dsel)
    return dist, ind


def radius_bruteforce(points, r, chunk=2048):
    """Independent fp64 brute force of K4r: all j with d2 <= r*r, ascending index, -1 padded."""
    p = np.asarray(points, dtype=np.float64)
    n = p.shape[0]
    lists = []
    for s in range(0, n, chunk):
        d2 = _sqdist_f64(p, p[s:s + chunk])
        for row in d2:
            lists.append(np.nonzero(row <= r * r)[0])
    kmax = max(len(x) for x in lists)
    ind = np.full((n, kmax), -1, dtype=np.int64)
    for i, x in enumerate(lists):
        ind[i, :len(x)] = x
    return ind


def nn1_correspondences(points1, points2, ratio):
    """K17 as train.py:186-193: 1-NN of scan 1 in scan 2, inliers = dist <= quantile(dist, ratio)."""
    from scipy.spatial import cKDTree
    p1 = np.asarray(points1, dtype=np.float64)
    p2 = np.asarray(points2, dtype=np.float64)
    dists, ids = cKDTree(p2).query(p1, k=1)
    th = np.quantile(dists[~np.isnan(dists)], ratio)
    mask1 = dists <= th
    return mask1, ids[mask1], th


# ----------------------------------------------------------------------------------------------
# K1-K3: model, rigid transform, points.  model.py:243-261, depth_cloud.py:122-152
# ----------------------------------------------------------------------------------------------
def model_bias(inc, w, exponent):
    """model.py:243-248: bias = pow(inc, exponent) @ w.T  (inc [N,1], w/exponent [1,P])."""
    x = torch.pow(inc, exponent)
    return torch.matmul(x, w.t()).view((-1, 1))


def model_apply(depth, inc, mask, w, exponent, kind='ScaledPolynomial'):
    """Masked depth correction: ScaledPolynomial model.py:250-261, Polynomial :188-199, Linear :113-146 (w = [[w0, w1,
    b]]), InvCos :289-313 and ScaledInvCos :316-349 (w = [[p0]]); ``exponent`` is used by the polynomials only."""
    assert kind in ('ScaledPolynomial', 'Polynomial', 'Linear', 'InvCos', 'ScaledInvCos')

    def corrected(d, g):
        if kind == 'Linear':
            return w[0, 0] * d + w[0, 1] * g + w[0, 2]
        if kind == 'InvCos':
            return d - w[0, 0] / torch.cos(g)
        if kind == 'ScaledInvCos':
            return d * (1. - w[0, 0] / torch.cos(g).abs())
        bias = model_bias(g, w, exponent)
        return d * (1. - bias) if kind == 'ScaledPolynomial' else d - bias
    if mask is None:
        return corrected(depth, inc)
    out = depth.clone()
    out[mask] = corrected(depth[mask], inc[mask])
    return out


def transform_cloud(vps, dirs, T, normals=None):
    """depth_cloud.py:135-152: vps' = vps R^T + t^T, dirs' = dirs R^T, normals' = normals R^T."""
    T = T.to(dtype=vps.dtype)
    R = T[:3, :3]
    t = T[:3, 3:]
    out = [torch.matmul(vps, R.t()) + t.t(), torch.matmul(dirs, R.t())]
    if normals is not None:
        out.append(torch.matmul(normals, R.t()))
    return tuple(out)


def points_from(vps, dirs, depth):
    """depth_cloud.py:122-124."""
    return vps + depth * dirs


def axis_angle_to_matrix(axis_angle):
    """Published pytorch3d algorithm (transforms/rotation_conversions.py, @stable):
    axis_angle_to_quaternion (small-angle series below 1e-6) followed by quaternion_to_matrix.
    UNPINNED (pytorch3d absent); see module docstring."""
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    s_over = torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half) / safe)
    q = torch.cat([torch.cos(half), axis_angle * s_over], dim=-1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
    ), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def xyz_axis_angle_to_matrix(xyz_axis_angle):
    """transform.py:68-78."""
    shp = xyz_axis_angle.shape[:-1]
    mat = torch.zeros(shp + (4, 4), dtype=xyz_axis_angle.dtype)
    mat[..., :3, :3] = axis_angle_to_matrix(xyz_axis_angle[..., 3:])
    mat[..., :3, 3] = xyz_axis_angle[..., :3]
    mat[..., 3, 3] = 1.
    return mat


# ----------------------------------------------------------------------------------------------
# K5-K12: gather, mean, weights, covariance, eigh, normals, incidence angle.
# depth_cloud.py:291-295,303-304,356-369,376-424 ; utils.py:109-154
# ----------------------------------------------------------------------------------------------
def covs(x, weights=None):
    """utils.py:109-149 with obs_axis=-2, var_axis=-1, center=True, correction=True.
    Materialises the [N,K,3,3] outer products exactly as the reference does."""
    if weights is not None:
        w = weights.sum(dim=-2, keepdim=True)
        xm = (weights * x).sum(dim=-2, keepdim=True) / w
    else:
        w = x.shape[-2]
        xm = x.mean(dim=-2, keepdim=True)
    xc = x - xm
    xx = xc.unsqueeze(-1) * xc.unsqueeze(-2)
    if weights is not None:
        xx = weights.unsqueeze(-1) * xx
    xx = xx.sum(dim=-3)
    w = w - 1
    if isinstance(w, torch.Tensor) and w.dtype.is_floating_point:
        w = w.clamp(1e-6, None)
    return xx / w


def trace(x):
    """utils.py:152-154."""
    return x.diagonal(dim1=-2, dim2=-1).sum(dim=-1)


def features(points, neighbors, dirs, weights=None, scale=None):
    """update_features (depth_cloud.py:426-433) on given neighbourhoods.

    points [N,3], neighbors i64 [N,K] (-1 = missing, gathers the last point like torch indexing
    does, depth_cloud.py:303-304), dirs [N,3].  Returns dict(mean, weights, cov, eigvals, eigvecs,
    normals, inc_angles, distances).
    """
    valid = (neighbors >= 0)
    if weights is None:
        weights = valid.to(points.dtype)[..., None]              # depth_cloud.py:213
    nbr = points[neighbors]                                      # :303-304
    distances = torch.linalg.norm(points.unsqueeze(1) - nbr, dim=-1)   # :200-204 (dead work, kept)
    w = weights.sum(dim=(-2, -1))[..., None]
    mean = (weights * nbr).sum(dim=-2) / w                       # :291-295
    weights = valid.to(points.dtype)[..., None]                  # :358
    if scale is not None:
        dist = (points - mean).norm(dim=1, keepdim=True)
        weights = weights * torch.exp(-(dist / scale) ** 2)[..., None]   # :360-363 (broadcast over K)
    cov = covs(nbr, weights=weights)                             # :366-369
    eigvals, eigvecs = torch.linalg.eigh(cov)                    # :376-396
    normals = eigvecs[..., 0]                                    # :414
    cos = (dirs * normals).sum(dim=-1)
    normals = -torch.sign(cos)[..., None] * normals              # :401-407
    inc = torch.arccos((dirs * normals).sum(dim=-1).abs()).unsqueeze(-1)   # :423
    return dict(mean=mean, weights=weights, cov=cov, eigvals=eigvals, eigvecs=eigvecs,
                normals=normals, inc_angles=inc, distances=distances, neighbor_points=nbr)


# ----------------------------------------------------------------------------------------------
# K13: filters / masks.  filters.py:85-113,184-254 ; preproc.py:53-62,122-164 ; depth_cloud.py:314-326
# ----------------------------------------------------------------------------------------------
def within_bounds(x, lo=None, hi=None):
    """filters.py:85-113: inclusive bounds; None / +-inf = unbounded; NaN compares False."""
    keep = torch.ones((x.numel(),), dtype=torch.bool)
    if lo is not None and lo > -float('inf'):
        keep = keep & (x.flatten() >= lo)
    if hi is not None and hi < float('inf'):
        keep = keep & (x.flatten() <= hi)
    return keep


def _eig_masks(eigvals, eigenvalue_bounds, eigenvalue_ratio_bounds):
    mask = torch.ones((eigvals.shape[0],), dtype=torch.bool)
    for e, lo, hi in (eigenvalue_bounds or []):                  # filters.py:196-221
        mask = mask & within_bounds(eigvals[:, int(e)], lo, hi)
    for i, j, lo, hi in (eigenvalue_ratio_bounds or []):         # filters.py:224-254
        mask = mask & within_bounds(eigvals[:, int(i)] / eigvals[:, int(j)], lo, hi)
    return mask


def local_mask(eigvals, eigenvalue_bounds=None, eigenvalue_ratio_bounds=None):
    """Mask part of local_feature_cloud, preproc.py:53-62."""
    return _eig_masks(eigvals, eigenvalue_bounds, eigenvalue_ratio_bounds)


def dispersion(vec, neighbors, weights):
    """depth_cloud.py:314-326: trace of the weighted covariance of vec[neighbors]."""
    return trace(covs(vec[neighbors], weights=weights))


def global_mask(mask, neighbors, eigvals, vps=None, dirs=None, weights=None, min_valid_neighbors=None,
                eigenvalue_bounds=None, eigenvalue_ratio_bounds=None, dir_dispersion_bounds=None,
                vp_dispersion_bounds=None):
    """preproc.py:122-164."""
    n = eigvals.shape[0]
    mask = torch.ones((n,), dtype=torch.bool) if mask is None else mask.clone()
    if min_valid_neighbors:
        mask &= within_bounds((neighbors >= 0).sum(dim=-1), lo=min_valid_neighbors)   # filters.py:184-193
    mask &= _eig_masks(eigvals, eigenvalue_bounds, eigenvalue_ratio_bounds)
    if dir_dispersion_bounds:
        mask &= within_bounds(dispersion(dirs, neighbors, weights), *dir_dispersion_bounds)
    if vp_dispersion_bounds:
        mask &= within_bounds(dispersion(vps, neighbors, weights), *vp_dispersion_bounds)
    return mask


# ----------------------------------------------------------------------------------------------
# K14-K16: losses.  loss.py:125-150,216-370,406-488
# ----------------------------------------------------------------------------------------------
def pointwise_loss(eigvals=None, cov=None, kind='min_eigval_loss', mask=None, offset=None, sqrt=False,
                   normalization=False, inlier_ratio=1.0, inlier_max_loss=None, inlier_loss_mult=1.0):
    """loss.py:250-289 / :330-363 including the quantile-inlier branch (:256-277): of the masked points only those
    whose raw loss is <= mult * quantile(raw loss, inlier_ratio) (and <= inlier_max_loss) contribute."""
    if kind == 'min_eigval_loss':
        if mask is not None:
            eigvals = eigvals[mask]
        loss = eigvals[:, 0]
        if normalization:
            loss = loss / eigvals.sum(dim=-1).clamp(min=1e-6)
    else:
        if mask is not None:
            cov = cov[mask]
        loss = trace(cov)
    if inlier_ratio < 1.0:                                            # loss.py:256-267
        q = torch.quantile(loss, inlier_ratio, dim=0)
        if inlier_loss_mult != 1.0:
            q = inlier_loss_mult * q
        inlier_max_loss = q if inlier_max_loss is None else torch.min(torch.as_tensor(inlier_max_loss, dtype=q.dtype), q)
    if inlier_max_loss is not None:                                   # :269-277
        loss = loss[loss <= inlier_max_loss]
    if offset is not None:
        loss = loss - offset
    loss = torch.relu(loss)
    if sqrt:
        loss = torch.sqrt(loss)
    return loss


def reduce_loss(x, reduction='mean'):
    """loss.py:125-150 (no weights, no nan skipping)."""
    if reduction == 'mean':
        return x.mean()
    if reduction == 'sum':
        return x.sum()
    return x


def point_to_plane(points, normals, masks):
    """loss.py:406-488 with precomputed correspondences (the train.py:178-210 call pattern).

    points / normals: lists of [N_s,3] per scan; masks[i] = (mask1 bool[N_i], idx2 i64[M_i])."""
    total = 0.0
    n_pairs = len(points) - 1
    for i in range(n_pairs):
        p1 = torch.as_tensor(points[i], dtype=torch.float)       # :436-437 (cast to fp32)
        p2 = torch.as_tensor(points[i + 1], dtype=torch.float)
        mask1, mask2 = masks[i]
        p1i, p2i = p1[mask1], p2[mask2]
        n1 = normals[i][mask1]
        k = torch.multiply(n1, p2i - p1i).sum(dim=-1, keepdims=True)
        d12 = torch.linalg.norm(p2i - (p2i - k * n1), dim=-1).mean()
        n2 = normals[i + 1][mask2]
        k = torch.multiply(n2, p1i - p2i).sum(dim=-1, keepdims=True)
        d21 = torch.linalg.norm(p1i - (p1i - k * n2), dim=-1).mean()
        total = total + 0.5 * (d12 + d21)
    return torch.as_tensor(total / n_pairs)


def point_to_point(points, masks):
    """loss.py:491-565 with precomputed correspondences: mean |x2 - x1| per pair (points cast to fp32, :524-525),
    averaged over the consecutive pairs."""
    total = 0.0
    n_pairs = len(points) - 1
    for i in range(n_pairs):
        p1 = torch.as_tensor(points[i], dtype=torch.float)
        p2 = torch.as_tensor(points[i + 1], dtype=torch.float)
        mask1, mask2 = masks[i]
        total = total + torch.linalg.norm(p2[mask2] - p1[mask1], dim=1).mean()       # :552-553
    return torch.as_tensor(total / n_pairs)


def shadow_mask(points, vps, dir_neighbors, angle_bounds):
    """filters.py:257-309: keep a point when the angles between (viewpoint - x) and (neighbour - x) over its direction
    neighbours (depth_cloud.py:217-224; -1 = missing) all lie within the bounds.  The bounds live in a float32 tensor
    (:277-278) and missing neighbours take their mean (:293-294)."""
    lo = 0.0 if (angle_bounds[0] is None or not (angle_bounds[0] >= 0.0)) else angle_bounds[0]
    hi = torch.pi if (angle_bounds[1] is None or not (angle_bounds[1] <= torch.pi)) else angle_bounds[1]
    bounds = torch.as_tensor([lo, hi])
    x = torch.as_tensor(points)
    o = torch.as_tensor(vps)
    ox = o.unsqueeze(dim=1) - x.unsqueeze(dim=1)
    nx = x[dir_neighbors] - x.unsqueeze(dim=1)
    a = torch.acos(torch.nn.functional.cosine_similarity(ox, nx, dim=-1))
    a[torch.as_tensor(dir_neighbors) < 0] = bounds.mean()
    return (a.amin(dim=-1) >= bounds[0]) & (a.amax(dim=-1) <= bounds[1]), a


# ----------------------------------------------------------------------------------------------
# One training iteration for one sequence, as eval.py:85-112 -> preproc.py:80-119,195-217 -> loss
# ----------------------------------------------------------------------------------------------
def eval_sequence(scans, poses, w, exponent, neighbors, mask, kind='min_eigval_loss', model='ScaledPolynomial',
                  normalization=True, sqrt=False, pose_deltas=None, reduction='none', weights=None, inlier_ratio=1.0,
                  inlier_max_loss=None, inlier_loss_mult=1.0):
    """scans: list of dict(vps, dirs, depth [n,1], inc [n,1], mask bool[n] or None) in the local frame.

    Returns (pointwise masked loss, dict of global-cloud features).  All torch ops, so
    ``loss.sum().backward()`` reproduces the reference's autograd backward (K18)."""
    if pose_deltas is not None:
        poses = torch.matmul(poses, xyz_axis_angle_to_matrix(pose_deltas))      # eval.py:68-82
    vps_g, dirs_g, depth_g = [], [], []
    for s, T in zip(scans, poses):
        d = model_apply(s['depth'], s['inc'], s.get('mask'), w, exponent, model)   # model.py:76-78
        v, r = transform_cloud(s['vps'], s['dirs'], T)                            # preproc.py:116
        vps_g.append(v), dirs_g.append(r), depth_g.append(d)
    vps_g, dirs_g, depth_g = torch.cat(vps_g), torch.cat(dirs_g), torch.cat(depth_g)   # :118
    x = points_from(vps_g, dirs_g, depth_g)
    f = features(x, neighbors, dirs_g, weights=weights)
    f['points'], f['dirs'], f['depth'] = x, dirs_g, depth_g
    loss = pointwise_loss(eigvals=f['eigvals'], cov=f['cov'], kind=kind, mask=mask, sqrt=sqrt,
                          normalization=normalization, inlier_ratio=inlier_ratio, inlier_max_loss=inlier_max_loss,
                          inlier_loss_mult=inlier_loss_mult)
    return reduce_loss(loss, reduction), f


# ----------------------------------------------------------------------------------------------
# K18: closed-form backward -- restated independently of autograd, used to pin the
# formula the HIP backward implements.  numpy fp64.
# ----------------------------------------------------------------------------------------------
def closed_form_backward(points, neighbors, mask, kind='min_eigval_loss', normalization=True, sqrt=False,
                         reduction='mean', offset=None):
    """Returns dict(loss, grad_points [N,3]) for L = reduce(pointwise loss over mask)."""
    x = np.asarray(points, dtype=np.float64)
    nb = np.asarray(neighbors)
    n, k = nb.shape
    wgt = (nb >= 0).astype(np.float64)
    xn = x[nb]                                   # -1 wraps to the last point, weight 0
    W = wgt.sum(1)
    D = np.maximum(W - 1.0, 1e-6)
    m = (wgt[..., None] * xn).sum(1) / W[:, None]
    d = xn - m[:, None, :]
    C = np.einsum('nk,nki,nkj->nij', wgt, d, d) / D[:, None, None]
    lam, V = np.linalg.eigh(C)
    v0 = V[:, :, 0]
    T = lam.sum(1)
    msk = np.ones(n, bool) if mask is None else np.asarray(mask, bool)
    if kind == 'min_eigval_loss':
        ell = lam[:, 0] / np.maximum(T, 1e-6) if normalization else lam[:, 0].copy()
    else:
        ell = T.copy()
    if offset is not None:
        off = np.zeros(n)
        off[msk] = offset
        ell = ell - off
    pos = ell > 0
    ell = np.maximum(ell, 0.0)
    M = float(msk.sum()) if reduction == 'mean' else 1.0
    a = msk * pos / M
    if sqrt:
        a = a * np.where(ell > 0, 0.5 / np.sqrt(np.where(ell > 0, ell, 1.0)), 0.0)
        ell = np.sqrt(ell)
    eye = np.eye(3)[None]
    vv = v0[:, :, None] * v0[:, None, :]
    if kind == 'min_eigval_loss':
        if normalization:
            Tc = np.maximum(T, 1e-6)
            G = a[:, None, None] * (vv / Tc[:, None, None]
                                    - ((T > 1e-6) * lam[:, 0] / Tc ** 2)[:, None, None] * eye)
        else:
            G = a[:, None, None] * vv
    else:
        G = a[:, None, None] * eye
    contrib = 2.0 * (wgt / D[:, None])[..., None] * np.einsum('nij,nkj->nki', G, d)
    g = np.zeros_like(x)
    np.add.at(g, nb.reshape(-1), contrib.reshape(-1, 3))
    loss = (ell * msk).sum() / M
    return dict(loss=loss, grad_points=g, pointwise=ell, eigvals=lam, mean=m, v0=v0)


def euler_matrix(ai, aj, ak):
    """tf.transformations.euler_matrix for its default axes 'sxyz' (published definition: static-frame rotations about x, y, z
    in that order, R = Rz(ak) Ry(aj) Rx(ai)); tf is a ROS package that is absent here, so the fixture generator substitutes
    this restatement into the reference (dataset.py:25,801) and records that in the fixture's meta."""
    import numpy as np
    Rx = np.array([[1, 0, 0], [0, np.cos(ai), -np.sin(ai)], [0, np.sin(ai), np.cos(ai)]])
    Ry = np.array([[np.cos(aj), 0, np.sin(aj)], [0, 1, 0], [-np.sin(aj), 0, np.cos(aj)]])
    Rz = np.array([[np.cos(ak), -np.sin(ak), 0], [np.sin(ak), np.cos(ak), 0], [0, 0, 1]])
    M = np.eye(4)
    M[:3, :3] = Rz @ Ry @ Rx
    return M
