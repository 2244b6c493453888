#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the LIVE reference (runs only where /root/reference exists).

Test infrastructure.  The reference is imported read-only from /root/reference/src with import
stubs for ROS / open3d / pytorch3d / tensorboard (none of them touches hot-path arithmetic, SURVEY
8c).  The single arithmetic stub is pytorch3d.transforms.axis_angle_to_matrix, replaced by the
oracle's restatement of the published algorithm (recorded in each fixture's ``meta``).

Before a fixture is written, the oracle restatement (oracle/dc_oracle.py) is run on the same inputs
and asserted equal to the reference's outputs, so a committed fixture certifies both.

Usage:  python oracle/gen_golden.py [names]    (writes tests/golden/{c0_plane,room_k10,icp_pairs,knn,grid,shadow,models,inliers,io,online}.npz, about 12 MB)
"""
import os
import sys
import tempfile
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import dc_oracle as O                                                     # noqa: E402
from depth_correction_amd.dataset import PlaneDataset, RoomBoxDataset, KittiLikeDataset, add_depth_noise  # noqa: E402

for m in ['pytorch3d', 'pytorch3d.io', 'pytorch3d.structures', 'pytorch3d.ops', 'pytorch3d.ops.knn',
          'pytorch3d.transforms', 'pytorch3d.renderer', 'open3d', 'ros_numpy', 'rospy', 'sensor_msgs',
          'sensor_msgs.msg', 'geometry_msgs', 'geometry_msgs.msg', 'nav_msgs', 'nav_msgs.msg', 'std_msgs',
          'std_msgs.msg', 'tf', 'tf.transformations', 'tf2_msgs', 'tf2_msgs.msg', 'torch.utils.tensorboard']:
    sys.modules[m] = MagicMock()
sys.modules['pytorch3d.transforms'].axis_angle_to_matrix = O.axis_angle_to_matrix
sys.modules['rospy'].is_shutdown.return_value = False
np.object = object                                   # removed numpy alias used at nearest_neighbors.py:69
sys.path.insert(0, '/root/reference/src')

import depth_correction.config as RC                                      # noqa: E402
RC.cmd_out = lambda *a, **k: ('unknown', '')          # config.py:160 shells out to git
from depth_correction.config import Config, PoseCorrection                # noqa: E402
from depth_correction.depth_cloud import DepthCloud                       # noqa: E402
from depth_correction.dataset import PlaneDataset as RefPlaneDataset      # noqa: E402
from depth_correction.eval import eval_loss_clouds                        # noqa: E402
from depth_correction.loss import create_loss, icp_loss, point_to_plane_dist, point_to_point_dist   # noqa: E402
from depth_correction.model import InvCos, Linear, Polynomial, ScaledInvCos, ScaledPolynomial   # noqa: E402
from depth_correction.nearest_neighbors import nearest_neighbors          # noqa: E402
from depth_correction.preproc import (establish_neighborhoods, global_cloud, global_cloud_mask,
                                      local_feature_cloud)                # noqa: E402
from depth_correction.transform import xyz_axis_angle_to_matrix          # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
META = ('reference: ctu-vras/depth_correction @ /root/reference; torch %s; scipy %s; numpy %s; '
        'pytorch3d.axis_angle_to_matrix substituted by oracle restatement (unpinned)'
        % (torch.__version__, __import__('scipy').__version__, np.__version__))


def npy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def compact(x):
    """Indices as int32 (values < 2^31), everything else untouched."""
    x = npy(x)
    return x.astype(np.int32) if x.dtype == np.int64 else x


def close(a, b, rtol=1e-9, atol=1e-12, what=''):
    a, b = npy(a), npy(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ok = np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
    assert ok, '%s: max abs diff %.3g' % (what, np.nanmax(np.abs(a - b)))


def xyz_of(cloud):
    return np.stack([cloud[f] for f in 'xyz'], axis=1).astype(np.float64)


def base_cfg(**kw):
    cfg = Config()
    cfg.log_dir = tempfile.mkdtemp()
    cfg.min_depth, cfg.max_depth, cfg.grid_res = 0.0, float('inf'), 0.0
    cfg.nn_r = None
    cfg.log_filters = False
    cfg.from_dict(kw)
    return cfg


def oracle_scans(ref_clouds):
    return [dict(vps=c.vps.detach(), dirs=c.dirs.detach(), depth=c.depth.detach(), inc=c.inc_angles.detach(),
                 mask=None if c.mask is None else c.mask) for c in ref_clouds]


def run_sequence(name, scans_xyz, poses_np, cfg, w0, exponent, model_cls, variants, pose_deltas0=None,
                 out=None, prefix='', grad_points_for=2):
    """Drive the reference exactly like train.py:94-215 (set-up) and eval.py:85-112 (iteration).

    With ``out``/``prefix`` given, only the variant results are added to an existing fixture dict
    (same scans, poses and neighbourhoods as the base run)."""
    base = out is None
    if base:
        out = dict(meta=np.array(META), cfg_nn_k=cfg.nn_k, cfg_min_valid_neighbors=cfg.min_valid_neighbors,
                   eigenvalue_ratio_bounds=np.array(cfg.eigenvalue_ratio_bounds, dtype=np.float64),
                   vp_dispersion_bounds=np.array(cfg.vp_dispersion_bounds, dtype=np.float64),
                   poses=poses_np, n_scans=len(scans_xyz))
    out[prefix + 'w'], out[prefix + 'exponent'] = np.array(w0), np.array(exponent)
    out[prefix + 'model'] = np.array(model_cls.__name__)
    clouds = [local_feature_cloud(xyz.copy(), cfg) for xyz in scans_xyz]             # train.py:94-104
    poses = torch.as_tensor(poses_np)
    for s, (xyz, c) in enumerate(zip(scans_xyz, clouds)):
        if not base:
            break
        out['scan%d_xyz' % s] = xyz_of(xyz)
        for f in ('dirs', 'depth', 'neighbors', 'eigvals', 'inc_angles', 'mask') + (('mean', 'cov') if s == 0 else ()):
            out['scan%d_%s' % (s, f)] = compact(getattr(c, f))
        out['scan%d_absdot' % s] = npy((c.dirs * c.normals).sum(-1).abs())
        # restatement check: local features
        dist_o, ind_o = O.knn_ckdtree(npy(c.points), cfg.nn_k)
        assert np.array_equal(ind_o, npy(c.neighbors)), 'local knn'
        f = O.features(c.points, c.neighbors, c.dirs)
        for k in ('mean', 'cov', 'eigvals', 'inc_angles'):
            close(f[k], getattr(c, k), what='local ' + k)
        m = O.local_mask(c.eigvals, cfg.eigenvalue_bounds, cfg.eigenvalue_ratio_bounds)
        assert torch.equal(m, c.mask), 'local mask'

    g0 = global_cloud(clouds=clouds, poses=poses)                                   # train.py:166
    ns = establish_neighborhoods(cloud=g0, cfg=cfg)                                 # train.py:172
    mask = global_cloud_mask(g0, g0.mask if hasattr(g0, 'mask') else None, cfg)     # train.py:212
    if base:
        out['g_neighbors'] = compact(ns[0])
        assert torch.equal(ns[1], (ns[0] >= 0).double()[..., None])                 # weights are implied
        out['g_mask'] = npy(mask)
        out['g0_points'], out['g0_eigvals'] = npy(g0.points), npy(g0.eigvals)
        out['g0_vp_dispersion'] = npy(g0.vp_dispersion())
        out['g0_dir_dispersion'] = npy(g0.dir_dispersion())
    # restatement check: global neighbourhoods and mask
    _, ind_o = O.knn_ckdtree(npy(g0.points), cfg.nn_k)
    assert np.array_equal(ind_o, npy(ns[0])), 'global knn'
    _, ind_b = O.knn_bruteforce(npy(g0.points), cfg.nn_k)
    assert np.array_equal(ind_b, npy(ns[0])), 'global knn vs brute force'
    lm = torch.cat([c.mask for c in clouds])
    m_o = O.global_mask(lm, ns[0], g0.eigvals, vps=g0.vps, dirs=g0.dirs, weights=ns[1],
                        min_valid_neighbors=cfg.min_valid_neighbors, eigenvalue_bounds=cfg.eigenvalue_bounds,
                        eigenvalue_ratio_bounds=cfg.eigenvalue_ratio_bounds,
                        dir_dispersion_bounds=cfg.dir_dispersion_bounds, vp_dispersion_bounds=cfg.vp_dispersion_bounds)
    assert torch.equal(m_o, mask), 'global mask'
    print('%s: N=%d, masked=%d' % (name, len(g0), int(mask.sum())))

    for tag, loss_name, loss_kwargs in variants:
        cfg.loss = loss_name
        cfg.loss_kwargs.update(loss_kwargs)
        cfg.pose_correction = PoseCorrection.pose if pose_deltas0 is not None else PoseCorrection.none
        loss_fun = create_loss(cfg)
        model = model_cls(w=list(w0), exponent=list(exponent))
        pd = None
        if pose_deltas0 is not None:
            pd = torch.tensor(pose_deltas0, dtype=torch.float64, requires_grad=True)
        loss, loss_clouds, poses_upd, feat = eval_loss_clouds([clouds], [poses], [pd], [mask], [ns], model,
                                                              loss_fun, cfg)   # eval.py:85
        fc = feat[0]
        fc.points.retain_grad()
        loss.backward()
        tag = prefix + tag
        out['%s_loss' % tag] = npy(loss)
        out['%s_pointwise' % tag] = npy(loss_clouds[0].loss)
        out['%s_grad_w' % tag] = npy(model.w.grad)
        if grad_points_for > 0:
            out['%s_grad_points' % tag] = npy(fc.points.grad)
            grad_points_for -= 1
        if pd is not None:
            out['%s_grad_pose_deltas' % tag] = npy(pd.grad)
            out[prefix + 'pose_deltas'] = np.array(pose_deltas0)
            out[prefix + 'poses_upd'] = npy(poses_upd[0])
        if tag == prefix + variants[0][0]:
            for f in ('points', 'eigvals') + (('mean', 'cov', 'inc_angles') if base else ()):
                out[prefix + 'g_' + f] = npy(getattr(fc, f))
            if base:
                out['g_absdot'] = npy((fc.dirs * fc.normals).sum(-1).abs())

        # restatement check: the whole iteration through the oracle, autograd backward
        w = torch.tensor([list(w0)], dtype=torch.float64, requires_grad=True)
        e = torch.tensor([list(exponent)], dtype=torch.float64)
        pdo = None if pd is None else torch.tensor(pose_deltas0, dtype=torch.float64, requires_grad=True)
        inl = {k: cfg.loss_kwargs[k] for k in ('inlier_ratio', 'inlier_max_loss', 'inlier_loss_mult') if k in cfg.loss_kwargs}
        gated = inl.get('inlier_ratio', 1.0) < 1.0 or inl.get('inlier_max_loss') is not None
        if gated:
            out['%s_n_inliers' % tag] = np.array(len(loss_clouds[0]))
        lo, fo = O.eval_sequence(oracle_scans(clouds), poses, w, e, ns[0], mask, kind=loss_name,
                                 model=model_cls.__name__, normalization=cfg.loss_kwargs['normalization'],
                                 sqrt=cfg.loss_kwargs['sqrt'], pose_deltas=pdo, reduction='mean', **inl)
        fo['points'].retain_grad()
        lo.backward()
        close(lo, loss, what=tag + ' loss')
        close(w.grad, model.w.grad, rtol=1e-8, what=tag + ' grad_w')
        close(fo['points'].grad, fc.points.grad, rtol=1e-8, atol=1e-14, what=tag + ' grad_points')
        close(fo['eigvals'], fc.eigvals, what=tag + ' eigvals')
        if pd is not None:
            close(pdo.grad, pd.grad, rtol=1e-8, atol=1e-14, what=tag + ' grad_pose')
        if gated:
            print('  %-22s loss=%.9g grad_w=%s inliers=%d' % (tag, loss.item(), npy(model.w.grad).ravel(), len(loss_clouds[0])))
            continue
        # closed form vs the reference's autograd
        cf = O.closed_form_backward(npy(fc.points), npy(ns[0]), npy(mask), kind=loss_name,
                                    normalization=cfg.loss_kwargs['normalization'], sqrt=cfg.loss_kwargs['sqrt'])
        close(cf['loss'], loss, what=tag + ' closed-form loss')
        close(cf['grad_points'], fc.points.grad, rtol=1e-6, atol=1e-13, what=tag + ' closed-form grad')
        print('  %-22s loss=%.9g grad_w=%s' % (tag, loss.item(), npy(model.w.grad).ravel()))

    if name:
        np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    return out


def gen_c0_plane():
    """BASELINE config 0: PlaneDataset(10 000, 2 poses), nn_k=4, fp64, min_eigval_loss."""
    ref_ds = RefPlaneDataset(n_pts=10_000, n_poses=2)
    our_ds = PlaneDataset(n_pts=10_000, n_poses=2)
    scans, poses = [], []
    rng = np.random.default_rng(7)
    for (rc, rp), (oc, op) in zip(ref_ds, our_ds):
        close(xyz_of(rc), xyz_of(oc), what='PlaneDataset cloud')        # generator restatement check
        close(rp, op, what='PlaneDataset pose')
        scans.append(add_depth_noise(oc, 0.01, rng))
        poses.append(op)
    cfg = base_cfg(nn_k=4, min_valid_neighbors=4)
    variants = [('mineig_norm', 'min_eigval_loss', dict(normalization=True, sqrt=False)),
                ('mineig_raw', 'min_eigval_loss', dict(normalization=False, sqrt=False))]
    run_sequence('c0_plane', scans, np.stack(poses), cfg, (-0.002, 0.001), (2.0, 4.0), ScaledPolynomial, variants)


def gen_room():
    """Room-box scans (configs 1-3 shape, reduced to 4 x 5000 points), nn_k=10, all loss variants,
    then the same with per-pose corrections (model_poses_learning pattern)."""
    ds = RoomBoxDataset(n_pts=2000, n_poses=4)
    scans = [c for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    cfg = base_cfg(nn_k=10, min_valid_neighbors=5, vp_dispersion_bounds=[])
    variants = [('mineig_norm', 'min_eigval_loss', dict(normalization=True, sqrt=False)),
                ('mineig_raw', 'min_eigval_loss', dict(normalization=False, sqrt=False)),
                ('mineig_norm_sqrt', 'min_eigval_loss', dict(normalization=True, sqrt=True)),
                ('mineig_raw_sqrt', 'min_eigval_loss', dict(normalization=False, sqrt=True)),
                ('trace', 'trace_loss', dict(sqrt=False)),
                ('trace_sqrt', 'trace_loss', dict(sqrt=True))]
    out = run_sequence(None, scans, poses, cfg, (1e-3, 2e-3), (2.0, 4.0), ScaledPolynomial, variants)

    rng = np.random.default_rng(11)
    pd = np.concatenate([0.02 * rng.normal(size=(4, 3)), 0.01 * rng.normal(size=(4, 3))], axis=1)
    pd[0] = 0.0                                                   # zero delta: the small-angle branch
    cfg = base_cfg(nn_k=10, min_valid_neighbors=5, vp_dispersion_bounds=[])
    variants = [('mineig_norm', 'min_eigval_loss', dict(normalization=True, sqrt=False)),
                ('trace', 'trace_loss', dict(sqrt=False))]
    run_sequence(None, scans, poses, cfg, (1e-3, 2e-3), (2.0, 4.0), ScaledPolynomial, variants,
                 pose_deltas0=pd, out=out, prefix='poses_', grad_points_for=1)
    cfg = base_cfg(nn_k=10, min_valid_neighbors=5, vp_dispersion_bounds=[])
    variants = [('mineig_norm', 'min_eigval_loss', dict(normalization=True, sqrt=False))]
    run_sequence('room_k10', scans, poses, cfg, (2e-3, -1e-3), (1.0, 3.0), Polynomial, variants,
                 out=out, prefix='poly_', grad_points_for=0)


def gen_inliers():
    """Quantile-inlier gating of the pointwise loss (loss.py:256-277) on the room_k10 inputs (same scans, poses, neighbourhoods
    and mask as tests/golden/room_k10.npz): only the results are stored."""
    ds = RoomBoxDataset(n_pts=2000, n_poses=4)
    scans = [c for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    variants = [('norm_r07', 'min_eigval_loss', dict(normalization=True, sqrt=False, inlier_ratio=0.7, inlier_loss_mult=1.0)),
                ('raw_sqrt_r09_m08', 'min_eigval_loss', dict(normalization=False, sqrt=True, inlier_ratio=0.9, inlier_loss_mult=0.8)),
                ('trace_r05', 'trace_loss', dict(sqrt=False, inlier_ratio=0.5, inlier_loss_mult=1.0))]
    out = dict(meta=np.array(META))
    cfg = base_cfg(nn_k=10, min_valid_neighbors=5, vp_dispersion_bounds=[])
    full = run_sequence(None, scans, poses, cfg, (1e-3, 2e-3), (2.0, 4.0), ScaledPolynomial, variants, out=dict(out),
                        prefix='inl_', grad_points_for=0)
    keep = {k: v for k, v in full.items() if k == 'meta' or (k.startswith('inl_') and not k.startswith('inl_g_')
                                                              and not k.endswith('_pointwise'))}
    for tag, _, kw in variants:
        keep['inl_%s_ratio' % tag] = np.array(kw['inlier_ratio'])
        keep['inl_%s_mult' % tag] = np.array(kw['inlier_loss_mult'])
    np.savez_compressed(os.path.join(GOLD, 'inliers.npz'), **keep)


def gen_models():
    """Linear / InvCos / ScaledInvCos (model.py:113-146, 289-349) through one training iteration on the room_k10 inputs
    (same scans, poses, neighbourhoods and mask as gen_room): loss, gradients of every model parameter, corrected depth
    of scan 0.  Outputs only -- the inputs are those of room_k10.npz."""
    ds = RoomBoxDataset(n_pts=2000, n_poses=4)
    scans = [c for c, _ in ds]
    poses = torch.as_tensor(np.stack([p for _, p in ds]))
    cfg = base_cfg(nn_k=10, min_valid_neighbors=5, vp_dispersion_bounds=[])
    cfg.loss, cfg.pose_correction = 'min_eigval_loss', PoseCorrection.none
    cfg.loss_kwargs.update(normalization=True, sqrt=False)
    clouds = [local_feature_cloud(xyz.copy(), cfg) for xyz in scans]
    g0 = global_cloud(clouds=clouds, poses=poses)
    ns = establish_neighborhoods(cloud=g0, cfg=cfg)
    mask = global_cloud_mask(g0, g0.mask if hasattr(g0, 'mask') else None, cfg)
    loss_fun = create_loss(cfg)
    out = dict(meta=np.array(META), inputs=np.array('room_k10.npz'))
    for name, model, params in (('Linear', Linear(w0=0.995, w1=2e-3, b=1e-3), ('w0', 'w1', 'b')),
                                ('InvCos', InvCos(p0=2e-3), ('p0',)),
                                ('ScaledInvCos', ScaledInvCos(p0=1e-3), ('p0',))):
        loss, loss_clouds, _, feat = eval_loss_clouds([clouds], [poses], [None], [mask], [ns], model, loss_fun, cfg)
        loss.backward()
        w0 = np.array([[float(getattr(model, p).detach()) for p in params]])
        grad = np.array([[float(getattr(model, p).grad) for p in params]])
        out[name + '_w'], out[name + '_loss'], out[name + '_grad_w'] = w0, npy(loss), grad
        out[name + '_depth0'] = npy(model(clouds[0]).depth)
        out[name + '_pointwise'] = npy(loss_clouds[0].loss)
        # restatement check
        w = torch.tensor(w0, dtype=torch.float64, requires_grad=True)
        lo, _ = O.eval_sequence(oracle_scans(clouds), poses, w, torch.zeros_like(w), ns[0], mask, model=name, reduction='mean')
        lo.backward()
        close(lo, loss, what=name + ' loss')
        close(w.grad, grad, rtol=1e-6, atol=1e-12, what=name + ' grad')
        print('models: %-13s loss=%.9g grad=%s' % (name, loss.item(), grad.ravel()))
    np.savez_compressed(os.path.join(GOLD, 'models.npz'), **out)


def gen_icp():
    """Config 4 shape reduced: KITTI-like ring scans, point-to-plane with precomputed correspondences
    (train.py:178-210), gradient w.r.t. model weights and per-pose corrections."""
    ds = KittiLikeDataset(n_poses=3, n_rings=16, n_azimuth=512)
    cfg = base_cfg(nn_k=8, min_valid_neighbors=5, vp_dispersion_bounds=[])
    clouds = [local_feature_cloud(c, cfg) for c, _ in ds]
    poses = torch.as_tensor(np.stack([p for _, p in ds]))
    ratio = 0.3
    masks = []
    for j in range(len(clouds) - 1):
        p1 = npy(clouds[j].transform(poses[j]).to_points())
        p2 = npy(clouds[j + 1].transform(poses[j + 1]).to_points())
        m1, m2, _ = O.nn1_correspondences(p1, p2, ratio)
        masks.append((m1, m2))
    rng = np.random.default_rng(5)
    pd0 = np.concatenate([0.02 * rng.normal(size=(3, 3)), 0.005 * rng.normal(size=(3, 3))], axis=1)
    model = ScaledPolynomial(w=[1e-3, 2e-3], exponent=[2.0, 4.0])
    pd = torch.tensor(pd0, dtype=torch.float64, requires_grad=True)
    poses_upd = torch.matmul(poses, xyz_axis_angle_to_matrix(pd))
    tc = [model(c).transform(p) for c, p in zip(clouds, poses_upd)]                 # loss.py:381-386
    loss = point_to_plane_dist(tc, icp_inlier_ratio=ratio, masks=masks)
    loss.backward()
    out = dict(meta=np.array(META), poses=npy(poses), pose_deltas=pd0, w=npy(model.w), exponent=npy(model.exponent),
               loss=npy(loss), grad_w=npy(model.w.grad), grad_pose_deltas=npy(pd.grad), n_scans=len(clouds),
               ratio=ratio)
    for s, c in enumerate(clouds):
        for f in ('vps', 'dirs', 'depth', 'inc_angles', 'mask', 'normals'):
            out['scan%d_%s' % (s, f)] = npy(getattr(c, f))
    for j, (m1, m2) in enumerate(masks):
        out['pair%d_mask1' % j], out['pair%d_idx2' % j] = m1, m2
    # restatement check
    w = torch.tensor(npy(model.w), requires_grad=True)
    pdo = torch.tensor(pd0, dtype=torch.float64, requires_grad=True)
    pu = torch.matmul(poses, O.xyz_axis_angle_to_matrix(pdo))
    pts, nrm = [], []
    for c, T in zip(clouds, pu):
        d = O.model_apply(c.depth, c.inc_angles, c.mask, w, model.exponent.detach())
        v, r, n = O.transform_cloud(c.vps, c.dirs, T, normals=c.normals.detach())
        pts.append(O.points_from(v, r, d)), nrm.append(n)
    lo = O.point_to_plane(pts, nrm, masks)
    lo.backward()
    close(lo, loss, what='icp loss')
    close(w.grad, model.w.grad, rtol=1e-6, what='icp grad_w')
    close(pdo.grad, pd.grad, rtol=1e-6, atol=1e-12, what='icp grad_pose')
    print('icp_pairs: loss=%.9g grad_w=%s' % (loss.item(), npy(model.w.grad).ravel()))

    # point-to-point distances on the same scans and correspondences (loss.py:491-565): through icp_loss
    # (icp_point_to_plane=False, :373-403) for the training form, and point_to_point_dist itself for the metric form
    model2 = ScaledPolynomial(w=[1e-3, 2e-3], exponent=[2.0, 4.0])
    pd2 = torch.tensor(pd0, dtype=torch.float64, requires_grad=True)
    poses_upd2 = torch.matmul(poses, xyz_axis_angle_to_matrix(pd2))
    loss2, _ = icp_loss([clouds], [poses_upd2], model2, masks=[masks], icp_point_to_plane=False, icp_inlier_ratio=ratio)
    loss2.backward()
    metric = point_to_point_dist([c.transform(p) for c, p in zip(clouds, poses)], icp_inlier_ratio=ratio, masks=masks)
    metric_nn = point_to_point_dist([c.transform(p) for c, p in zip(clouds, poses)], icp_inlier_ratio=ratio,
                                    differentiable=False)
    out.update(p2p_loss=npy(loss2), p2p_grad_w=npy(model2.w.grad), p2p_grad_pose_deltas=npy(pd2.grad),
               p2p_metric=npy(metric), p2p_metric_nn=npy(metric_nn))
    w2 = torch.tensor(npy(model2.w), requires_grad=True)
    pdo2 = torch.tensor(pd0, dtype=torch.float64, requires_grad=True)
    pu2 = torch.matmul(poses, O.xyz_axis_angle_to_matrix(pdo2))
    pts2 = []
    for c, T in zip(clouds, pu2):
        d = O.model_apply(c.depth, c.inc_angles, c.mask, w2, model2.exponent.detach())
        v, r = O.transform_cloud(c.vps, c.dirs, T)
        pts2.append(O.points_from(v, r, d))
    lo2 = O.point_to_point(pts2, masks)
    lo2.backward()
    close(lo2, loss2, what='p2p loss')
    close(w2.grad, model2.w.grad, rtol=1e-6, what='p2p grad_w')
    close(pdo2.grad, pd2.grad, rtol=1e-6, atol=1e-12, what='p2p grad_pose')
    print('icp_pairs (point to point): loss=%.9g metric=%.9g metric_nn=%.9g grad_w=%s'
          % (loss2.item(), metric.item(), metric_nn.item(), npy(model2.w.grad).ravel()))
    np.savez_compressed(os.path.join(GOLD, 'icp_pairs.npz'), **out)


def gen_shadow():
    """Scan-shadow filter (filters.py:257-309) on direction neighbourhoods (depth_cloud.py:217-224) and the depth
    pre-filter (filters.py:116-141) of a ring scan that sees two walls and the ground behind their edges, as the online
    node applies them (preproc.py:44-47; scripts/depth_correction:31-58)."""
    from depth_correction.filters import filter_depth, filter_shadow_points
    ds = KittiLikeDataset(n_poses=1, n_rings=32, n_azimuth=1024)
    cloud, _ = ds[0]
    out = dict(meta=np.array(META))
    for tag, dtype in (('f64', np.float64), ('f32', np.float32)):
        dc = DepthCloud.from_structured_array(cloud, dtype=dtype)
        # a range step every 40 beams (an occluding pole in front of the background): shadow points at its edges
        depth = dc.depth.clone()
        az = torch.arange(len(depth)) % 1024
        depth[(az % 40) < 3] *= 0.6
        dc = DepthCloud(dc.vps, dc.dirs, depth)
        keep = filter_depth(dc, min=1.0, max=25.0, only_mask=True)
        dc = dc[keep]
        dc.update_points()
        angle = 0.017453 * 1.5                                       # config.py:204 shadow_neighborhood_angle, widened
        dc.update_dir_neighbors(angle=angle)
        bounds = [float(np.radians(5.0)), float('inf')]              # config.py:205
        dc.loss = torch.arange(len(dc), dtype=dc.depth.dtype)         # a sliced field that reveals which points survive
        kept = filter_shadow_points(dc, list(bounds), log=False)
        mask = torch.zeros((len(dc),), dtype=torch.bool)
        mask[kept.loss.long()] = True
        mo, ang = O.shadow_mask(dc.points, dc.vps, dc.dir_neighbors, list(bounds))
        assert torch.equal(mo, mask), 'shadow mask restatement differs'
        ang[dc.dir_neighbors < 0] = float('nan')
        out.update({tag + '_vps': npy(dc.vps), tag + '_dirs': npy(dc.dirs), tag + '_depth': npy(dc.depth),
                    tag + '_depth_keep': npy(keep), tag + '_dir_neighbors': compact(dc.dir_neighbors),
                    tag + '_mask': npy(mask), tag + '_angle_min': npy(torch.nan_to_num(ang, nan=10.0).amin(dim=-1)),
                    tag + '_angle_max': npy(torch.nan_to_num(ang, nan=-10.0).amax(dim=-1))})
        print('shadow %s: %d rays, %d after depth filter, K_dir=%d, %d kept by the shadow filter'
              % (tag, len(keep), len(dc), dc.dir_neighbors.shape[1], int(mask.sum())))
    out['angle'], out['bounds_deg'] = angle, 5.0
    np.savez_compressed(os.path.join(GOLD, 'shadow.npz'), **out)


def gen_grid():
    """filters.filter_grid (filters.py:24-82) for the three keep modes and a seeded generator."""
    from depth_correction.filters import filter_grid
    rng = np.random.default_rng(21)
    pts = np.concatenate([rng.uniform(-3, 3, size=(4000, 3)) * [1, 1, 0.1], rng.normal(size=(1000, 3))])
    pts = pts.astype(np.float32).astype(np.float64)
    out = dict(meta=np.array(META), points=pts, grid_res=0.25)
    for keep in ('first', 'last', 'random'):
        for po in (False, True):
            ind = filter_grid(pts, 0.25, only_mask=True, keep=keep, preserve_order=po, rng=np.random.default_rng(135))
            out['%s_%d' % (keep, po)] = np.asarray(ind, dtype=np.int32)
    print('grid: %d points -> %d voxels' % (len(pts), len(out['last_0'])))
    np.savez_compressed(os.path.join(GOLD, 'grid.npz'), **out)


def gen_knn():
    """nearest_neighbors() itself: k, k within r, r only (nearest_neighbors.py:22-80)."""
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-2, 2, size=(2000, 3)) * [1, 1, 0.02],
                          rng.uniform(-1, 1, size=(1000, 3))]).astype(np.float32).astype(np.float64)
    p = torch.as_tensor(pts)
    out = dict(meta=np.array(META), points=pts)
    d, i = nearest_neighbors(p, p, k=10)
    out['k10_dist'], out['k10_ind'] = npy(d), npy(i)
    do, io = O.knn_bruteforce(pts, 10)
    assert np.array_equal(io, npy(i))
    close(do, d, what='knn dist')
    d, i = nearest_neighbors(p, p, k=8, r=0.15)
    out['k8_r015_dist'], out['k8_r015_ind'] = npy(d), npy(i)
    do, io = O.knn_bruteforce(pts, 8, r=0.15)
    assert np.array_equal(io, npy(i)), 'k within r'
    close(np.where(np.isinf(do), -1, do), np.where(np.isinf(npy(d)), -1, npy(d)), what='knn r dist')
    _, i = nearest_neighbors(p, p, r=0.12)
    out['r012_ind'] = npy(i)
    assert np.array_equal(O.radius_bruteforce(pts, 0.12), npy(i)), 'radius'
    assert np.array_equal(O.radius_ckdtree(pts, 0.12), npy(i)), 'radius ckdtree'
    print('knn: k10 / k8+r / r ok, Kmax(r=0.12)=%d' % npy(i).shape[1])
    np.savez_compressed(os.path.join(GOLD, 'knn.npz'), **out)


def gen_io():
    """Scan and pose files read by the LIVE reference readers: KITTI-360 ``.bin`` with the ego-box crop
    (datasets/kitti360.py:96-109), ASL-laser point CSV / ``.npz`` (asl_laser.py:33-45), pose CSV read and written
    (asl_laser.py:48-66), FEE-corridor structured ``.npz`` (fee_corridor.py:35-45).  The fixture holds the FILES (as bytes)
    and what the reference's readers and DepthCloud.from_structured_array / from_points + filter_depth make of them."""
    for m in ['kitti360scripts', 'kitti360scripts.helpers', 'kitti360scripts.helpers.annotation', 'kitti360scripts.helpers.labels',
              'kitti360scripts.helpers.ply', 'kitti360scripts.devkits', 'kitti360scripts.devkits.commons',
              'kitti360scripts.devkits.commons.loadCalibration', 'matplotlib', 'matplotlib.cm', 'matplotlib.pyplot']:
        sys.modules.setdefault(m, MagicMock())
    from depth_correction.datasets import asl_laser, fee_corridor, kitti360
    from depth_correction.filters import filter_depth
    from numpy.lib.recfunctions import structured_to_unstructured, unstructured_to_structured
    rng = np.random.default_rng(5)
    tmp = tempfile.mkdtemp()
    out = dict(meta=np.array(META))
    rd = lambda path: np.frombuffer(open(path, 'rb').read(), dtype=np.uint8).copy()

    # ---- KITTI-360 .bin: rays around the car, a fifth of them inside the 1 m ego box
    n = 3000
    xyz = rng.normal(size=(n, 3)) * [12.0, 9.0, 1.5]
    xyz[::5] = rng.uniform(-0.99, 0.99, size=(len(xyz[::5]), 3))
    xyz[7] = [1.0, 0.5, 0.2]                                   # exactly on the box: |x| <= d is dropped
    raw = np.concatenate([xyz, rng.uniform(0, 1, size=(n, 1))], axis=1).astype(np.float32)
    os.makedirs(os.path.join(tmp, 'velo'))
    path = os.path.join(tmp, 'velo', '%010d.bin' % 3)
    raw.tofile(path)
    ds = object.__new__(kitti360.Dataset)                      # the reader alone: no sequence folder to open
    ds.cloud_dir = os.path.join(tmp, 'velo')
    cloud = ds.local_cloud(3)
    assert cloud.dtype.names == ('x', 'y', 'z', 'i') and 0 < len(cloud) < n
    out['kitti_bin'] = rd(path)
    out['kitti_xyzi'] = structured_to_unstructured(cloud)
    for tag, dtype in (('f32', np.float32), ('f64', np.float64)):
        dc = DepthCloud.from_structured_array(cloud, dtype=dtype)
        keep = filter_depth(dc, min=2.0, max=25.0, only_mask=True)
        out['kitti_%s_dirs' % tag], out['kitti_%s_depth' % tag], out['kitti_%s_keep' % tag] = npy(dc.dirs), npy(dc.depth), npy(keep)
        assert bool((dc.vps == 0).all())

    # ---- ASL-laser point CSV (header; id, x, y, z, intensity) and its .npz twin
    pts = rng.normal(size=(500, 3)) * [6.0, 5.0, 2.0]
    path = os.path.join(tmp, 'PointCloud7.csv')
    with open(path, 'w') as f:
        f.write('timestamp,x,y,z,intensity\n')
        for i, q in enumerate(pts):
            f.write('%d,%.9g,%.9g,%.9g,%.4f\n' % (1000 + i, q[0], q[1], q[2], rng.uniform()))
    got = asl_laser.read_points(path)
    assert got.shape == (500, 3)
    out['asl_csv'], out['asl_csv_points'] = rd(path), got
    dc = DepthCloud.from_points(torch.as_tensor(got), dtype=torch.float64)      # (with a numpy array and a numpy dtype the reference raises)
    out['asl_f64_dirs'], out['asl_f64_depth'] = npy(dc.dirs), npy(dc.depth)
    path = os.path.join(tmp, 'cloud7.npz')
    np.savez(path, got)
    assert np.array_equal(asl_laser.read_points_npz(path), got)
    out['asl_npz'] = rd(path)

    # ---- pose CSV: written by the reference, read back by it
    ids = [3, 4, 7]
    poses = []
    for k in ids:
        T = np.eye(4)
        T[:3, :3] = npy(O.axis_angle_to_matrix(torch.tensor([[0.1 * k, -0.05 * k, 0.2]], dtype=torch.float64)))[0]
        T[:3, 3] = [1.5 * k, -0.25 * k, 0.1]
        poses.append(T)
    path = os.path.join(tmp, 'poses.csv')
    asl_laser.write_poses(ids, poses, path, ts=[10.5, 11.5, 12.5])
    rids, rposes = asl_laser.read_poses(path)
    assert rids == ids
    out['poses_csv'], out['poses_ids'], out['poses_T'] = rd(path), np.array(rids), np.stack(rposes)
    close(np.stack(rposes), np.stack(poses), rtol=0, atol=1e-9, what='pose csv round trip')

    # ---- FEE corridor: structured array with viewpoints under the key 'cloud', string pose ids
    m = 400
    vp = rng.normal(size=(m, 3)) * 0.05 + [0.2, -0.1, 0.4]
    xyz = vp + rng.normal(size=(m, 3)) * [5.0, 4.0, 1.0]
    arr = unstructured_to_structured(np.concatenate([xyz, vp], axis=1).astype(np.float32),
                                     names=['x', 'y', 'z', 'vp_x', 'vp_y', 'vp_z'])
    path = os.path.join(tmp, 'scan.npz')
    np.savez(path, cloud=arr)
    got = fee_corridor.read_points_npz(path)
    assert got.dtype.names == arr.dtype.names
    out['fee_npz'] = rd(path)
    dc = DepthCloud.from_structured_array(got, dtype=np.float64)
    out['fee_f64_vps'], out['fee_f64_dirs'], out['fee_f64_depth'] = npy(dc.vps), npy(dc.dirs), npy(dc.depth)
    path = os.path.join(tmp, 'fee_poses.csv')
    with open(path, 'w') as f:
        f.write('poseId, timestamp, ' + ', '.join('T%d%d' % (r, c) for r in range(4) for c in range(4)) + '\n')
        for k, T in zip(ids, poses):
            f.write('%s, %.9f, %s\n' % ('1669300%03d' % k, 0.5 * k, ', '.join('%.9f' % x for x in T.flatten())))
    fids, fposes = fee_corridor.read_poses(path)
    out['fee_poses_csv'], out['fee_poses_ids'], out['fee_poses_T'] = rd(path), np.array([int(i) for i in fids]), np.asarray(fposes)
    print('io: kitti %d -> %d rows, asl csv %d, fee npz %d, %d poses' % (n, len(cloud), len(pts), m, len(ids)))
    np.savez_compressed(os.path.join(GOLD, 'io.npz'), **out)


def gen_helpers():
    """Host helpers the caller scripts import (scripts/train_demo:4,10, model_poses_learning_icp:16): utils.delta_transform /
    rotation_angle / translation_norm / transform_inv (utils.py:174-205) and the dataset wrappers NoisyPoseDataset /
    NoisyDepthDataset (dataset.py:776-846) from the LIVE reference.  tf.transformations.euler_matrix (ROS) is substituted by
    the oracle's restatement, like pytorch3d's axis_angle_to_matrix."""
    import depth_correction.dataset as RD
    from depth_correction import utils as RU
    from numpy.lib.recfunctions import unstructured_to_structured, structured_to_unstructured
    RD.euler_matrix = O.euler_matrix
    rng = np.random.default_rng(11)
    out = dict(meta=np.array(META + '; tf.transformations.euler_matrix substituted by oracle restatement (unpinned)'))
    Ts = []
    for k in range(4):
        T = O.euler_matrix(*rng.normal(size=3) * 0.4)
        T[:3, 3] = rng.normal(size=3) * 2.0
        Ts.append(T)
    Ts = np.stack(Ts)
    out['poses'] = Ts
    out['delta'] = np.stack([RU.delta_transform(Ts[0], T) for T in Ts])
    out['rotation_angle'] = np.array([RU.rotation_angle(T) for T in Ts])
    out['translation_norm'] = np.array([RU.translation_norm(T) for T in Ts])
    out['transform_inv'] = np.stack([RU.transform_inv(T) for T in Ts])
    pts = rng.normal(size=(50, 3)) * [4.0, 3.0, 1.0]
    cloud = unstructured_to_structured(pts, names=['x', 'y', 'z'])
    base = [(cloud.copy(), T) for T in Ts]
    noise = [0.01, 0.02, 0.03, 0.1, 0.2, 0.3]
    out['pose_noise'] = np.array(noise)
    for mode in ('pose', 'common'):
        ds = RD.NoisyPoseDataset(base, noise=noise, mode=mode)
        out['noisy_pose_' + mode] = np.stack([p for _, p in ds])
    ds = RD.NoisyDepthDataset([(cloud.copy(), Ts[0])], noise=0.05)
    out['cloud_xyz'] = pts
    out['noisy_depth_xyz'] = structured_to_unstructured(next(iter(ds))[0][['x', 'y', 'z']])
    print('helpers: %d poses, pose noise modes pose / common, depth noise' % len(Ts))
    np.savez_compressed(os.path.join(GOLD, 'helpers.npz'), **out)


def gen_online():
    """The online correction node's per-scan statements (scripts/depth_correction:31-58) on the LIVE reference:
        dc = local_feature_cloud(input_cloud, cfg); dc = model(dc); dc.update_points(); out = dc.to_structured_array()
    for a ring scan with range steps (shadow points at their edges), with the shadow filter on, k-NN neighbourhoods, the
    default planarity mask and a ScaledPolynomial model; input with and without viewpoint fields."""
    ds = KittiLikeDataset(n_poses=1, n_rings=32, n_azimuth=1024)
    cloud, _ = ds[0]
    xyz = xyz_of(cloud)
    d = np.linalg.norm(xyz, axis=1)
    az = np.arange(len(xyz)) % 1024
    xyz[(az % 40) < 3] *= 0.6                                     # occluding poles
    keep = (np.linalg.norm(xyz, axis=1) > 1.0) & (np.linalg.norm(xyz, axis=1) < 25.0)   # "depth and grid filters are run earlier"
    xyz = xyz[keep].astype(np.float32)
    from numpy.lib.recfunctions import unstructured_to_structured
    out = dict(meta=np.array(META), xyz=xyz)
    w, e = [2e-3, -1e-3], [2.0, 4.0]
    out['w'], out['exponent'] = np.array(w), np.array(e)
    # (float64 clouds only: on a float32 cloud with a mask the reference's model raises -- its weights are always float64 and
    # index_put refuses the mixed dtypes, model.py:260)
    for tag, dtype, with_vp in (('f64', 'float64', False), ('f64vp', 'float64', True)):
        cfg = Config()
        cfg.log_dir = tempfile.mkdtemp()
        cfg.log_filters = False
        cfg.float_type = dtype
        cfg.nn_k, cfg.nn_r = 10, None
        cfg.shadow_neighborhood_angle = 0.017453 * 1.5
        cfg.shadow_angle_bounds = [float(np.radians(5.0)), float('inf')]
        if with_vp:
            vp = np.tile(np.array([[0.05, -0.02, 0.1]], dtype=np.float32), (len(xyz), 1))
            arr = unstructured_to_structured(np.concatenate([xyz + vp, vp], axis=1), names=['x', 'y', 'z', 'vp_x', 'vp_y', 'vp_z'])
        else:
            arr = unstructured_to_structured(xyz, names=['x', 'y', 'z'])
        model = ScaledPolynomial(w=w, exponent=e)
        with torch.no_grad():
            dc = local_feature_cloud(arr, cfg)
            n_after_shadow = len(dc)
            dc = model(dc)
            dc.update_points()
        # the fields to_structured_array publishes (depth_cloud.py:508-533), cast as it casts them; the call itself fails under
        # numpy >= 2 (merge_arrays' default fill value -1 does not fit the uint8 mask), which is not the path's arithmetic
        out[tag + '_points'] = npy(dc.get_points()).astype(np.float32)
        out[tag + '_vps'] = npy(dc.vps).astype(np.float32)
        out[tag + '_mask'] = npy(dc.mask).astype(np.uint8)
        out[tag + '_inc'] = npy(dc.inc_angles).astype(np.float32)[:, 0]
        out[tag + '_normals'] = npy(dc.normals).astype(np.float32)
        out[tag + '_depth'] = npy(dc.depth)
        out[tag + '_neighbors'] = compact(dc.neighbors)
        assert 0 < n_after_shadow < len(xyz) and bool(dc.mask.any()) and not bool(dc.mask.all())
        print('online %s: %d rays -> %d after the shadow filter, %d in the planarity mask' % (tag, len(xyz), n_after_shadow, int(dc.mask.sum())))
    out['shadow_neighborhood_angle'], out['shadow_bound_deg'], out['nn_k'] = 0.017453 * 1.5, 5.0, 10
    np.savez_compressed(os.path.join(GOLD, 'online.npz'), **out)


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['knn', 'grid', 'c0', 'room', 'icp', 'shadow', 'models', 'inliers', 'io', 'online', 'helpers']
    if 'io' in which:
        gen_io()
    if 'online' in which:
        gen_online()
    if 'helpers' in which:
        gen_helpers()
    if 'grid' in which:
        gen_grid()
    if 'knn' in which:
        gen_knn()
    if 'c0' in which:
        gen_c0_plane()
    if 'room' in which:
        gen_room()
    if 'icp' in which:
        gen_icp()
    if 'shadow' in which:
        gen_shadow()
    if 'models' in which:
        gen_models()
    if 'inliers' in which:
        gen_inliers()
