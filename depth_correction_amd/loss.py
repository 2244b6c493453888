"""Map-consistency losses with the reference's signatures (loss.py:125-579).

Two execution paths produce the same numbers:
  * fused (eval.eval_loss_clouds, plan.SequencePlan): covariance -> eigen -> pointwise loss -> backward record in
    the HIP kernels, never materialising per-point features;
  * un-fused (this module on a DepthCloud whose features came from ``update_all``): the pointwise post-processing
    of loss.py:250-289 on ``cloud.eigvals`` / ``cloud.cov`` (small ``[M]`` tensors), differentiable down to the
    points through dc_features_bwd.  All of the reference's options are available here (mask, offset, sqrt,
    normalisation, quantile inliers, reductions).
``icp_loss`` with precomputed correspondences runs as one kernel per scan pair (dc_p2plane_sequence for
point-to-plane, dc_p2point_sequence for point-to-point distances) including its backward to the model weights and
poses; ``point_to_point_dist`` as a metric on GPU clouds (scripts/model_poses_learning:142-146) uses the same kernel.
"""
from __future__ import annotations

import warnings
from enum import Enum

import torch

from . import ops
from .depth_cloud import DepthCloud
from .plan import PlanRegistry
from .utils import trace

__all__ = ['batch_loss', 'create_loss', 'icp_loss', 'loss_by_name', 'min_eigval_loss', 'point_to_plane_dist',
           'point_to_point_dist', 'reduce', 'Reduction', 'trace_loss', 'icp_correspondences']


class Reduction(Enum):
    NONE = 'none'
    MEAN = 'mean'
    SUM = 'sum'


def reduce(x, reduction=Reduction.MEAN, weights=None, only_finite=False, skip_nans=False):
    assert reduction in Reduction
    keep = x.isfinite() if only_finite else (~x.isnan() if skip_nans else None)
    if keep is not None:
        weights = weights[keep] if weights is not None else None
        x = x[keep]
    if reduction == Reduction.MEAN:
        return x.mean() if weights is None else (weights * x).sum() / weights.sum()
    if reduction == Reduction.SUM:
        return x.sum() if weights is None else (weights * x).sum()
    return x


def batch_loss(loss_fun, clouds, masks=None, offsets=None, reduction=Reduction.MEAN, only_finite=False,
               skip_nans=False, **kwargs):
    """Pointwise losses of several sequences, concatenated, then reduced once (loss.py:181-213)."""
    assert callable(loss_fun) and isinstance(clouds, (list, tuple))
    masks = len(clouds) * [None] if masks is None else masks
    offsets = len(clouds) * [None] if offsets is None else offsets
    assert len(masks) == len(clouds) == len(offsets)
    parts, loss_clouds = [], []
    for cloud, mask, offset in zip(clouds, masks, offsets):
        part, loss_cloud = loss_fun(cloud, mask=mask, offset=offset, reduction=Reduction.NONE, **kwargs)
        parts.append(part)
        loss_clouds.append(loss_cloud)
    return reduce(torch.cat(parts), reduction=reduction, only_finite=only_finite, skip_nans=skip_nans), loss_clouds


def _pointwise(raw_fun, cloud, mask, offset, sqrt, reduction, inlier_max_loss, inlier_ratio, inlier_loss_mult,
               only_finite, skip_nans):
    """Shared tail of min_eigval_loss / trace_loss (loss.py:244-294, 324-370)."""
    assert offset is None or isinstance(offset, (DepthCloud, torch.Tensor))
    if mask is not None:
        print('Using %.3f valid entries from input cloud.' % mask.float().mean())
        cloud, mask = cloud[mask], None
    loss = raw_fun(cloud)
    if inlier_ratio < 1.0:
        assert offset is None
        q = torch.quantile(loss, inlier_ratio, dim=0)
        if inlier_loss_mult != 1.0:
            q = inlier_loss_mult * q
        inlier_max_loss = q if inlier_max_loss is None else torch.min(torch.as_tensor(inlier_max_loss, dtype=q.dtype,
                                                                                       device=q.device), q)
    if inlier_max_loss is not None:
        assert offset is None
        mask = loss <= inlier_max_loss
        print('Using %i (%.3g) inliers with loss <= %.3g.' % (mask.sum().item(), mask.float().mean().item(),
                                                              float(torch.as_tensor(inlier_max_loss).detach())))
        cloud, loss = cloud[mask], loss[mask]
    if offset is not None:
        loss = loss - (offset.loss if isinstance(offset, DepthCloud) else offset)
    loss = torch.relu(loss)
    if sqrt:
        loss = torch.sqrt(loss)
    cloud = cloud.copy()
    cloud.loss = loss
    return reduce(loss, reduction=reduction, only_finite=only_finite, skip_nans=skip_nans), cloud


def min_eigval_loss(cloud, mask=None, offset=None, sqrt=False, normalization=False, reduction=Reduction.MEAN,
                    inlier_max_loss=None, inlier_ratio=1.0, inlier_loss_mult=1.0, only_finite=False, skip_nans=False,
                    **kwargs):
    """Smallest neighbourhood eigenvalue (optionally over the total variance) averaged over the masked points."""
    if isinstance(cloud, (list, tuple)):
        return batch_loss(min_eigval_loss, cloud, masks=mask, offsets=offset, sqrt=sqrt, normalization=normalization,
                          reduction=reduction, inlier_max_loss=inlier_max_loss, inlier_ratio=inlier_ratio,
                          inlier_loss_mult=inlier_loss_mult, only_finite=only_finite, skip_nans=skip_nans)
    assert isinstance(cloud, DepthCloud) and cloud.eigvals is not None

    def raw(c):
        lam = c.eigvals
        return lam[:, 0] / lam.sum(dim=-1).clamp(min=1e-6) if normalization else lam[:, 0]
    return _pointwise(raw, cloud, mask, offset, sqrt, reduction, inlier_max_loss, inlier_ratio, inlier_loss_mult,
                      only_finite, skip_nans)


def trace_loss(cloud, mask=None, offset=None, sqrt=None, reduction=Reduction.MEAN, inlier_max_loss=None,
               inlier_ratio=1.0, inlier_loss_mult=1.0, only_finite=False, skip_nans=False, **kwargs):
    """Trace of the neighbourhood covariance averaged over the masked points."""
    if isinstance(cloud, (list, tuple)):
        return batch_loss(trace_loss, cloud, masks=mask, offsets=offset, sqrt=sqrt, reduction=reduction,
                          inlier_max_loss=inlier_max_loss, inlier_ratio=inlier_ratio, inlier_loss_mult=inlier_loss_mult,
                          only_finite=only_finite, skip_nans=skip_nans)
    assert isinstance(cloud, DepthCloud) and cloud.cov is not None
    return _pointwise(lambda c: trace(c.cov), cloud, mask, offset, sqrt, reduction, inlier_max_loss, inlier_ratio,
                      inlier_loss_mult, only_finite, skip_nans)


# ---- ICP-style losses ---------------------------------------------------------------------------------------------
def icp_correspondences(points1, points2, ratio):
    """(mask1 bool [N1], idx2 int64 [M]): 1-NN of scan 1 in scan 2 on the GPU, inliers = dist <= quantile(dist, ratio)
    (train.py:186-193, loss.py:440-452)."""
    dist, idx = ops.knn(points2.detach().contiguous(), 1, query=points1.detach().to(points2.dtype).contiguous())
    dist, idx = dist[:, 0].contiguous(), idx[:, 0].contiguous()
    mask1, idx2, _ = ops.nn1_corr(dist, idx, ratio)                  # quantile select + compaction on the device (dc_nn1_corr)
    return mask1, idx2.long(), dist


def _pair_points(cloud):
    pts = cloud.to_points() if cloud.points is None else cloud.points
    assert not torch.all(torch.isnan(pts))
    return torch.as_tensor(pts, dtype=torch.float)


def _pair_matches(points1, points2, masks, i, ratio):
    if masks is not None:
        m1, m2 = masks[i]
        return torch.as_tensor(m1, device=points1.device), torch.as_tensor(m2, device=points1.device), torch.tensor(-1.0)
    mask1, idx2, dist = icp_correspondences(points1, points2, ratio)
    return mask1, idx2, dist[mask1].mean()


def point_to_plane_dist(clouds: list, icp_inlier_ratio=0.5, masks=None, differentiable=True, verbose=False, **kwargs):
    """Mean symmetric point-to-plane distance over consecutive scan pairs (loss.py:406-488), un-fused form."""
    assert 0.0 <= icp_inlier_ratio <= 1.0
    assert masks is None or len(clouds) == len(masks) + 1
    total, n_pairs = 0.0, len(clouds) - 1
    for i in range(n_pairs):
        c1, c2 = clouds[i], clouds[i + 1]
        assert c1.normals is not None, 'Cloud must have normals computed to estimate point to plane distance'
        p1, p2 = _pair_points(c1), _pair_points(c2)
        mask1, mask2, inl_err = _pair_matches(p1, p2, masks, i, icp_inlier_ratio)
        a, b = p1[mask1], p2[mask2]
        assert len(a) > 0, 'Point clouds do not intersect. Try to sample lidar scans more frequently'
        n1, n2 = c1.normals[mask1], c2.normals[mask2]
        d12 = torch.linalg.norm((n1 * (b - a)).sum(dim=-1, keepdims=True) * n1, dim=-1).mean()
        d21 = torch.linalg.norm((n2 * (a - b)).sum(dim=-1, keepdims=True) * n2, dim=-1).mean()
        total = total + 0.5 * (d12 + d21)
        if inl_err > 0.3:
            warnings.warn('ICP inliers error is too big: %.3f (> 0.3) [m] for pairs (%i, %i)' % (inl_err, i, i + 1))
        if verbose:
            print('Mean point to plane distance: %.3f [m] for scans: (%i, %i), inliers error: %.6f'
                  % (float(total), i, i + 1, float(inl_err)))
    return torch.as_tensor(total / n_pairs)


def _as_cloud(c):
    return c if isinstance(c, DepthCloud) else DepthCloud.from_points(torch.as_tensor(c))


def point_to_point_dist(clouds: list, icp_inlier_ratio=0.5, masks=None, differentiable=True, verbose=False, **kwargs):
    """Mean point-to-point distance over consecutive scan pairs (loss.py:491-565).

    On GPU clouds that carry no autograd graph (the map-accuracy metric of scripts/model_poses_learning:142-146) the
    distances of all pairs come from one dc_p2point_sequence call; correspondences not given are found by the GPU 1-NN
    builder.  Clouds inside an autograd graph take the reference's tensor expressions (icp_loss routes training through
    the fused kernel with its hand-derived backward instead)."""
    assert 0.0 <= icp_inlier_ratio <= 1.0
    assert masks is None or len(clouds) == len(masks) + 1
    n_pairs = len(clouds) - 1
    pts = [_pair_points(c) if isinstance(c, DepthCloud) else torch.as_tensor(c, dtype=torch.float) for c in clouds]
    if n_pairs > 0 and all(p.is_cuda and not p.requires_grad for p in pts):
        pairs, errs = [], []
        for i in range(n_pairs):
            mask1, mask2, inl_err = _pair_matches(pts[i], pts[i + 1], masks, i, icp_inlier_ratio)
            ia = (torch.nonzero(mask1).reshape(-1) if mask1.dtype == torch.bool else mask1).to(torch.int32).contiguous()
            ib = mask2.to(torch.int32).contiguous()
            assert len(ia) > 0 and len(ib) > 0, 'Point clouds do not intersect. Try to sample lidar scans more frequently'
            pairs.append((i, i + 1, ia, ib))
            errs.append(inl_err)
        # the points as they are (already corrected and posed by the caller): unit "directions" x, depth 1, identity poses
        scans = [(ops.PointSet(None, p.contiguous(), torch.ones((len(p),), dtype=p.dtype, device=p.device)), None) for p in pts]
        seq = ops.IcpSequence(scans, pairs, with_model=False, plane=False)
        eye = torch.eye(4, dtype=torch.float64, device=pts[0].device)[:3].reshape(1, 12).repeat(len(pts), 1).contiguous()
        out = seq.eval(eye)
        for i, inl_err in enumerate(errs):
            if inl_err > 0.3:
                warnings.warn('ICP inliers error is too big: %.3f (> 0.3) [m] for pairs (%i, %i)' % (inl_err, i, i + 1))
        return out[0].to(torch.float32)
    total = 0.0
    for i in range(n_pairs):
        p1, p2 = pts[i], pts[i + 1]
        mask1, mask2, inl_err = _pair_matches(p1, p2, masks, i, icp_inlier_ratio)
        a, b = p1[mask1], p2[mask2]
        assert len(a) > 0 and len(b) > 0, 'Point clouds do not intersect. Try to sample lidar scans more frequently'
        total = total + torch.linalg.norm(b - a, dim=1).mean()
        if inl_err > 0.3:
            warnings.warn('ICP inliers error is too big: %.3f (> 0.3) [m] for pairs (%i, %i)' % (inl_err, i, i + 1))
        if verbose:
            print('Mean point to point distance: %.3f [m] for scans: (%i, %i), inliers error: %.6f'
                  % (float(total), i, i + 1, float(inl_err)))
    return torch.as_tensor(total / n_pairs)


class _P2PlaneSequence(torch.autograd.Function):
    """icp_loss of one sequence (all consecutive pairs) through dc_p2plane_sequence: forward and backward come out
    of the same host call."""

    @staticmethod
    def forward(ctx, w, exponent, poses, plan, kind):
        P12 = poses.detach().to(torch.float64)[:, :3, :].reshape(-1, 12).contiguous()
        wv = None if w is None else w.detach().reshape(-1).to(torch.float64).contiguous()
        ev = None if exponent is None else exponent.detach().reshape(-1).to(torch.float64).contiguous()
        out = plan.eval(P12, kind, wv, ev)
        ctx.save_for_backward(out)
        ctx.meta = (None if w is None else (w.shape, w.dtype), poses.dtype, poses.shape[0],
                    exponent.shape if isinstance(exponent, torch.Tensor) else None)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        wmeta, pdt, ns, eshape = ctx.meta
        nt = (out.numel() - 1 - 12 * ns) // 2
        gw = ge = gT = None
        if wmeta is not None and ctx.needs_input_grad[0]:
            gw = (g * out[1:1 + nt]).reshape(wmeta[0]).to(wmeta[1])
        if eshape is not None and ctx.needs_input_grad[1]:
            ge = (g * out[1 + nt:1 + 2 * nt]).reshape(eshape)
        if ctx.needs_input_grad[2]:
            gT = torch.zeros((ns, 4, 4), dtype=torch.float64, device=out.device)
            gT[:, :3, :] = (g * out[1 + 2 * nt:]).reshape(ns, 3, 4)
            gT = gT.to(pdt)
        return gw, ge, gT, None, None


_icp_plans = PlanRegistry()


def _icp_sequence_plan(seq_clouds, seq_masks, with_model, plane):
    fields = [t for c in seq_clouds for t in (c.vps, c.dirs, c.depth, c.inc_angles, c.normals if plane else None, c.mask)]
    fields += [m for pair in seq_masks for m in pair]

    def build():
        scans = []
        for c in seq_clouds:
            n = len(c)
            cont = lambda t: t.detach().expand(n, t.shape[-1]).contiguous() if t.dim() == 2 else t.detach().contiguous()
            vps = None if not bool(c.vps.any()) else cont(c.vps)
            inc = cont(c.inc_angles) if (with_model and c.inc_angles is not None) else None
            scans.append((ops.PointSet(vps, cont(c.dirs), cont(c.depth), inc, c.mask if with_model else None),
                          cont(c.normals.to(c.dirs.dtype)) if plane else None))
        dev = scans[0][0].device
        pairs = []
        for i, (m1, m2) in enumerate(seq_masks):
            m1 = torch.as_tensor(m1, device=dev)
            ia = (torch.nonzero(m1).reshape(-1) if m1.dtype == torch.bool else m1).to(torch.int32).contiguous()
            ib = torch.as_tensor(m2, device=dev).to(torch.int32).contiguous()
            assert len(ia) > 0, 'Point clouds do not intersect. Try to sample lidar scans more frequently'
            pairs.append((i, i + 1, ia, ib))
        return ops.IcpSequence(scans, pairs, with_model=with_model, plane=plane)
    return _icp_plans.get(fields, (bool(with_model), bool(plane)), build)


def _fused_icp_sequence(seq_clouds, seq_poses, model, seq_masks, plane=True):
    """ICP loss of one sequence through dc_p2plane_sequence / dc_p2point_sequence (clouds in the sensor frame + poses + model)."""
    kind = getattr(model, 'kernel_kind', None) if model is not None else None
    w, e = model.kernel_params() if kind else (None, None)
    plan = _icp_sequence_plan(seq_clouds, seq_masks, bool(kind), plane)
    poses = seq_poses if isinstance(seq_poses, torch.Tensor) else torch.stack(list(seq_poses))
    return _P2PlaneSequence.apply(w, e, poses, plan, kind)


class _MovedClouds(object):
    """The corrected, posed scans of a sequence (what the reference's icp_loss concatenates into its loss cloud,
    loss.py:381-386,396-398), produced only when a caller looks at the returned cloud."""

    def __init__(self, seq, poses, model, loss):
        self._args, self._cloud, self.loss = (seq, poses, model), None, loss

    def _materialize(self):
        if self._cloud is None:
            seq, poses, model = self._args
            with torch.no_grad():
                moved = [model(c) for c in seq] if model is not None else list(seq)
                if poses is not None:
                    moved = [c.transform(p) for c, p in zip(moved, poses)]
                self._cloud = DepthCloud.concatenate(moved)
                self._cloud.loss = self.loss
        return self._cloud

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self._materialize(), name)

    def __len__(self):
        return sum(len(c) for c in self._args[0])


def icp_loss(clouds, poses=None, model=None, masks=None, **kwargs):
    """ICP-like loss over lists of sequences of scans (loss.py:373-403)."""
    p2plane = kwargs['icp_point_to_plane']
    fused = (masks is not None and poses is not None and clouds and clouds[0][0].dirs.is_cuda
             and (not p2plane or all(c.normals is not None for seq in clouds for c in seq))
             and (model is None or getattr(model, 'kernel_kind', None)))
    loss, loss_cloud = 0., []
    for i, seq in enumerate(clouds):
        if fused:
            loss_seq = _fused_icp_sequence(seq, poses[i], model, masks[i], plane=bool(p2plane))
            loss = loss + loss_seq
            loss_cloud.append(_MovedClouds(seq, poses[i], model, loss))
            continue
        moved = [model(c) for c in seq] if model is not None else seq
        if poses is not None:
            moved = [c.transform(p) for c, p in zip(moved, poses[i])]
        fun = point_to_plane_dist if p2plane else point_to_point_dist
        loss_seq = fun(moved, masks=None if masks is None else masks[i], **kwargs)
        loss = loss + loss_seq
        cloud = DepthCloud.concatenate(moved)
        cloud.loss = loss
        loss_cloud.append(cloud)
    return loss / len(clouds), loss_cloud


def loss_by_name(name):
    assert name in ('min_eigval_loss', 'trace_loss', 'icp_loss')
    return globals()[name]


def create_loss(cfg):
    loss = loss_by_name(cfg.loss)

    def loss_fun(*args, **kwargs):
        return loss(*args, **kwargs, **cfg.loss_kwargs)
    loss_fun.name = cfg.loss
    loss_fun.kwargs = cfg.loss_kwargs
    return loss_fun
