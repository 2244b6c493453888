"""Loss evaluation over sequences with the reference's signatures (eval.py:31-112).

``eval_loss_clouds`` is the body of one training iteration.  When the configuration is the one the fused kernels
cover (ball neighbourhoods on the GPU, min-eigenvalue or trace loss -- with or without quantile inliers -- without
offsets / distance weights, any of the reference's models or none) every sequence is evaluated by its cached
``SequencePlan`` -- one to three kernel launches -- and the returned loss carries the hand-derived backward to
``model.w`` / ``model.exponent`` / the pose corrections.  Any other configuration goes through the un-fused DepthCloud operators with identical results.
"""
from __future__ import annotations

import torch

from .config import Config, NeighborhoodType, PoseCorrection
from .depth_cloud import DepthCloud
from .plan import PlanRegistry, SequencePlan, consistency_loss
from .preproc import (compute_neighborhood_features, global_cloud, global_cloud_mask, local_feature_cloud,
                      offset_cloud)
from .transform import corrected_poses, xyz_axis_angle_to_matrix

__all__ = ['create_corrected_poses', 'eval_loss_clouds', 'initialize_pose_corrections', 'fused_supported',
           'PlanCloud', 'LazyFeatureCloud']


def initialize_pose_corrections(datasets, cfg: Config):
    """Zero 6-vector corrections per cfg.pose_correction (eval.py:31-65); lengths of ``datasets`` are used."""
    kwargs = dict(dtype=cfg.torch_float_type(), device=cfg.device, requires_grad=True)
    deltas = []
    for ds in datasets:
        if cfg.pose_correction == PoseCorrection.common:
            delta = deltas[0] if deltas else torch.zeros((1, 6), **kwargs)
        elif cfg.pose_correction == PoseCorrection.sequence:
            delta = torch.zeros((1, 6), **kwargs)
        elif cfg.pose_correction == PoseCorrection.pose:
            delta = torch.zeros((len(ds), 6), **kwargs)
        else:
            delta = None
        deltas.append(delta)
    return deltas


def create_corrected_poses(poses, pose_deltas, cfg: Config):
    """T_s = T0_s * [Exp(axis-angle) | xyz] (eval.py:68-82)."""
    if cfg.pose_correction == PoseCorrection.none:
        return poses
    assert len(poses) == len(pose_deltas)
    if cfg.pose_correction == PoseCorrection.common:
        assert all(d is pose_deltas[0] for d in pose_deltas[1:])
    return [corrected_poses(p, d) for p, d in zip(poses, pose_deltas)]


def fused_supported(clouds, model, cfg: Config):
    kw = cfg.loss_kwargs
    return (getattr(cfg, 'fused', True) and cfg.nn_type == NeighborhoodType.ball
            and cfg.loss in ('min_eigval_loss', 'trace_loss') and not cfg.loss_offset and not cfg.nn_scale
            # NaN-dropping reductions run inside the fused kernels (round 4); together with quantile gating: un-fused operators
            and not ((kw.get('only_finite') or kw.get('skip_nans'))
                     and (kw.get('inlier_ratio', 1.0) < 1.0 or kw.get('inlier_max_loss') is not None))
            and (model is None or getattr(model, 'kernel_kind', None) is not None)
            and clouds[0][0].dirs.is_cuda and all(c.inc_angles is not None for seq in clouds for c in seq))


class PlanCloud(object):
    """Lazy view of a fused evaluation: the DepthCloud fields callers may inspect (points, eigvals, loss, mask) are
    produced on first access by one more forward with the per-point outputs switched on."""

    def __init__(self, plan, w, exponent, poses, count=None, inliers=None):
        self._plan, self._args, self._out = plan, (w, exponent, poses), None
        self.count = plan.count if count is None else count      # pointwise terms behind the sequence's share of the mean loss
        self._inliers = inliers                  # gated evaluation (loss.py:256-277): the centre rows that were kept

    def _materialize(self):
        if self._out is None:
            p = self._plan
            w, e, poses = self._args
            out = p.forward(w, e, poses, want_pointwise=True, want_eigvals=True)
            loss = p.unpermute(out['pointwise'])
            mask = None if p.mask_full is None else p.unpermute(p.mask_full)
            if self._inliers is not None:        # gated evaluation: the loss cloud is the inliers (loss.py:269-277)
                inl = self._inliers
                if p.centre_idx is not None:     # compact centre rows -> all points
                    inl = torch.zeros((p.n,), dtype=torch.bool, device=inl.device).index_put_((p.centre_idx.long(),), inl)
                mask = p.unpermute(inl)
            self._out = dict(points=p.points(), eigvals=p.unpermute(out['eigvals']), loss=loss, mask=mask)
        return self._out

    def __getattr__(self, name):
        if name in ('points', 'eigvals', 'loss', 'mask'):
            return self._materialize()[name]
        raise AttributeError(name)

    def __len__(self):
        return self._plan.n


_plans = PlanRegistry()


class LazyFeatureCloud(object):
    """The global feature cloud of a sequence (global_cloud -> compute_neighborhood_features, eval.py:90-98), computed on
    first access.  The reference builds it every iteration even when the ICP loss never looks at it (eval.py:100-104);
    callers that do (callbacks logging the map) get the same cloud, everyone else pays nothing."""

    def __init__(self, seq_clouds, model, poses, nn, cfg):
        self._args, self._cloud = (seq_clouds, model, poses, nn, cfg), None

    def _materialize(self):
        if self._cloud is None:
            seq_clouds, model, poses, nn, cfg = self._args
            with torch.no_grad():
                g = global_cloud(clouds=seq_clouds, model=model, poses=poses.detach())
                self._cloud = compute_neighborhood_features(cloud=g, neighborhoods=nn, cfg=cfg)
        return self._cloud

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self._materialize(), name)

    def __len__(self):
        return sum(len(c) for c in self._args[0])


def _plan_for(seq_clouds, poses, nn, mask, model, cfg):
    """SequencePlan of (local clouds, neighbourhoods, mask), all constant over the optimisation: kept in a registry keyed
    by the identity and version of every tensor it was built from (plan.PlanRegistry)."""
    neighbors = nn[0]
    kw = cfg.loss_kwargs
    nan_policy = 'only_finite' if kw.get('only_finite') else ('skip_nans' if kw.get('skip_nans') else None)     # loss.py:125-137
    # ball neighbourhoods (a ragged table) and no pose corrections: rows of similar length share wavefronts (SequencePlan.degree_group)
    by_degree = bool(getattr(cfg, 'nn_r', None)) and not getattr(cfg, 'nn_k', None) and str(getattr(cfg, 'pose_correction', 'none')).endswith('none')
    ball = bool(getattr(cfg, 'nn_r', None)) and not getattr(cfg, 'nn_k', None)       # rows of different lengths: SequencePlan.heavy_first
    flags = (cfg.loss, bool(kw.get('normalization', False)), bool(kw.get('sqrt', False)),
             getattr(model, 'kernel_kind', None) if model is not None else None, nan_policy, by_degree, ball)
    tensors = [t for c in seq_clouds for t in (c.vps, c.dirs, c.depth, c.inc_angles, c.mask)] + [neighbors, mask]

    def build():
        return SequencePlan(seq_clouds, poses.detach(), neighbors, None if mask is None else mask.to(neighbors.device),
                            model_kind=flags[3] or 'ScaledPolynomial', loss=cfg.loss,
                            normalization=flags[1] and cfg.loss == 'min_eigval_loss', sqrt=flags[2], nan_policy=nan_policy,
                            degree_group=by_degree, heavy_first=ball)
    return _plans.get(tensors, flags, build)


def eval_loss_clouds(clouds, poses, pose_deltas, masks, ns, model, loss_fun, cfg: Config):
    """Loss of all sequences for the current model / pose corrections (eval.py:85-112).

    Returns (loss, loss_clouds, updated poses, feature clouds) like the reference."""
    poses_upd = create_corrected_poses(poses, pose_deltas, cfg)

    if cfg.loss == 'icp_loss':
        if clouds[0][0].normals is None:
            clouds = [[local_feature_cloud(c, cfg) for c in seq] for seq in clouds]
        loss, loss_cloud = loss_fun(clouds, poses_upd, model, masks=masks)
        feat = [LazyFeatureCloud(c, model, p, nn, cfg) for c, p, nn in zip(clouds, poses_upd, ns)] if ns else None
        return loss, loss_cloud, poses_upd, feat

    if fused_supported(clouds, model, cfg):
        # masks are established once (train.py:212-215); when absent they come from one un-fused evaluation
        if not masks or masks[0] is None:
            masks = []
            for c, p, nn in zip(clouds, poses_upd, ns):
                g = compute_neighborhood_features(cloud=global_cloud(clouds=c, model=model, poses=p.detach()),
                                                  neighborhoods=nn, cfg=cfg)
                masks.append(global_cloud_mask(g, g.mask, cfg))
        use_model = model is not None and getattr(model, 'kernel_kind', None) is not None
        total, count, views = 0.0, 0.0, []
        kw = cfg.loss_kwargs
        gating = dict(inlier_ratio=kw.get('inlier_ratio', 1.0), inlier_max_loss=kw.get('inlier_max_loss'),
                      inlier_loss_mult=kw.get('inlier_loss_mult', 1.0))
        for c, p, nn, m in zip(clouds, poses_upd, ns, masks):
            plan = _plan_for(c, p, nn, m, model, cfg)
            w, e = model.kernel_params() if use_model else (None, None)
            s, cnt = consistency_loss(plan, w, e, p, **gating)          # cnt: masked points, or inliers (device scalar)
            total, count = total + s, count + cnt
            views.append(PlanCloud(plan, w, e, p, count=cnt,
                                   inliers=getattr(plan, 'inlier_rows', None) if (isinstance(cnt, torch.Tensor) and not plan.nan_policy) else None))
        if isinstance(count, torch.Tensor):
            loss = total / count                                          # 0 / 0 = nan, like the mean of no inliers
        else:
            loss = total / count if count > 0 else total * float('nan')   # mean over all masked points (loss.py:211)
        return loss, views, poses_upd, views

    offsets = [offset_cloud(c, model) for c in clouds] if cfg.loss_offset else None
    global_clouds = [global_cloud(clouds=c, model=model, poses=p) for c, p in zip(clouds, poses_upd)]
    feat_clouds = [compute_neighborhood_features(cloud=cloud, neighborhoods=nn, cfg=cfg)
                   for cloud, nn in zip(global_clouds, ns)]
    if (not masks or masks[0] is None) and isinstance(feat_clouds[0], DepthCloud):
        masks = [global_cloud_mask(cloud, cloud.mask if hasattr(cloud, 'mask') else None, cfg) for cloud in feat_clouds]
    loss, loss_cloud = loss_fun(feat_clouds, mask=masks, offset=offsets)
    return loss, loss_cloud, poses_upd, feat_clouds
