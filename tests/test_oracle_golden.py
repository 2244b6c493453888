"""The oracle (oracle/dc_oracle.py) against the golden vectors generated from the live reference
(oracle/gen_golden.py).  Runs on the CPU; this is what pins the checker every GPU parity test relies on."""
import numpy as np
import pytest
import torch

import dc_oracle as O
from helpers import t, npy, scans_from_golden

TIGHT = dict(rtol=1e-9, atol=1e-13)


def test_knn_contract(golden):
    g = golden('knn')
    for fun in (O.knn_ckdtree, O.knn_bruteforce):
        d, i = fun(g['points'], 10)
        assert np.array_equal(i, g['k10_ind']) and np.array_equal(d, g['k10_dist'])
        d, i = fun(g['points'], 8, r=0.15)
        assert np.array_equal(i, g['k8_r015_ind']) and np.array_equal(d, g['k8_r015_dist'])
    assert np.array_equal(O.radius_ckdtree(g['points'], 0.12), g['r012_ind'])
    assert np.array_equal(O.radius_bruteforce(g['points'], 0.12), g['r012_ind'])


@pytest.mark.parametrize('name', ['c0_plane', 'room_k10'])
def test_local_and_global_features(golden, name):
    g = golden(name)
    k = int(g['cfg_nn_k'])
    for s in range(int(g['n_scans'])):
        x = t(g['scan%d_xyz' % s])
        _, ind = O.knn_ckdtree(g['scan%d_xyz' % s], k)
        assert np.array_equal(ind, g['scan%d_neighbors' % s])
        f = O.features(x, torch.as_tensor(ind), t(g['scan%d_dirs' % s]))
        np.testing.assert_allclose(npy(f['eigvals']), g['scan%d_eigvals' % s], **TIGHT)
        np.testing.assert_allclose(npy(f['inc_angles']), g['scan%d_inc_angles' % s], rtol=0, atol=1e-9)
        m = O.local_mask(f['eigvals'], None, g['eigenvalue_ratio_bounds'].tolist())
        assert np.array_equal(npy(m), g['scan%d_mask' % s])
    _, ind = O.knn_bruteforce(g['g0_points'], k)
    assert np.array_equal(ind, g['g_neighbors'])


@pytest.mark.parametrize('name', ['c0_plane', 'room_k10'])
def test_iteration_all_variants(golden, name):
    g = golden(name)
    variants = [(k[:-5], ) for k in g if k.endswith('_loss') and not k.startswith(('poses_', 'poly_'))]
    scans = scans_from_golden(g)
    nbr, mask = t(g['g_neighbors']).long(), t(g['g_mask'])
    for (tag,) in variants:
        kind = 'trace_loss' if tag.startswith('trace') else 'min_eigval_loss'
        w = torch.tensor(g['w'].reshape(1, -1), requires_grad=True)
        loss, f = O.eval_sequence(scans, t(g['poses']), w, t(g['exponent'].reshape(1, -1)), nbr, mask, kind=kind,
                                  normalization='norm' in tag, sqrt=tag.endswith('sqrt'), reduction='mean')
        f['points'].retain_grad()
        loss.backward()
        np.testing.assert_allclose(npy(loss), g[tag + '_loss'], rtol=1e-10)
        np.testing.assert_allclose(npy(w.grad), g[tag + '_grad_w'], rtol=1e-8)
        if tag + '_grad_points' in g:
            np.testing.assert_allclose(npy(f['points'].grad), g[tag + '_grad_points'], rtol=1e-8, atol=1e-14)
            cf = O.closed_form_backward(g['g_points'], g['g_neighbors'], g['g_mask'], kind=kind, normalization='norm' in tag,
                                        sqrt=tag.endswith('sqrt'))
            ref = g[tag + '_grad_points']
            np.testing.assert_allclose(cf['grad_points'], ref, rtol=1e-6, atol=1e-10 * np.abs(ref).max())
            np.testing.assert_allclose(cf['loss'], g[tag + '_loss'], rtol=1e-10)


@pytest.mark.parametrize('tag,kind,norm,sqrt', [('norm_r07', 'min_eigval_loss', True, False),
                                                ('raw_sqrt_r09_m08', 'min_eigval_loss', False, True),
                                                ('trace_r05', 'trace_loss', False, False)])
def test_quantile_inlier_gating(golden, tag, kind, norm, sqrt):
    """loss.py:256-277 (inlier_ratio < 1, inlier_loss_mult) in the oracle vs the live reference's loss, dL/dw and
    number of inliers on the room_k10 inputs."""
    g, gi = golden('room_k10'), golden('inliers')
    scans = scans_from_golden(g)
    nbr, mask = t(g['g_neighbors']).long(), t(g['g_mask'])
    w = torch.tensor(gi['inl_w'].reshape(1, -1), requires_grad=True)
    ratio, mult = float(gi['inl_%s_ratio' % tag]), float(gi['inl_%s_mult' % tag])
    loss, _ = O.eval_sequence(scans, t(g['poses']), w, t(gi['inl_exponent'].reshape(1, -1)), nbr, mask, kind=kind,
                              normalization=norm, sqrt=sqrt, reduction='none', inlier_ratio=ratio, inlier_loss_mult=mult)
    assert len(loss) == int(gi['inl_%s_n_inliers' % tag])
    loss.mean().backward()
    np.testing.assert_allclose(npy(loss.mean()), gi['inl_%s_loss' % tag], rtol=1e-10)
    np.testing.assert_allclose(npy(w.grad), gi['inl_%s_grad_w' % tag], rtol=1e-8)


def test_masks_and_dispersion(golden):
    for name in ('c0_plane', 'room_k10'):
        g = golden(name)
        scans = scans_from_golden(g)
        vps, dirs = [], []
        for s, T in zip(scans, t(g['poses'])):
            v, d = O.transform_cloud(s['vps'], s['dirs'], T)
            vps.append(v), dirs.append(d)
        vps, dirs = torch.cat(vps), torch.cat(dirs)
        nbr = t(g['g_neighbors']).long()
        wts = (nbr >= 0).double()[..., None]
        np.testing.assert_allclose(npy(O.dispersion(vps, nbr, wts)), g['g0_vp_dispersion'], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(npy(O.dispersion(dirs, nbr, wts)), g['g0_dir_dispersion'], rtol=1e-9, atol=1e-15)
        m = O.global_mask(torch.cat([s['mask'] for s in scans]), nbr, t(g['g0_eigvals']), vps=vps, dirs=dirs, weights=wts,
                          min_valid_neighbors=int(g['cfg_min_valid_neighbors']),
                          eigenvalue_ratio_bounds=g['eigenvalue_ratio_bounds'].tolist(),
                          vp_dispersion_bounds=g['vp_dispersion_bounds'].tolist() or None)
        assert np.array_equal(npy(m), g['g_mask'])


def test_pose_corrections_and_polynomial(golden):
    g = golden('room_k10')
    scans = scans_from_golden(g)
    nbr, mask = t(g['g_neighbors']).long(), t(g['g_mask'])
    pd = torch.tensor(g['poses_pose_deltas'], requires_grad=True)
    w = torch.tensor(g['poses_w'].reshape(1, -1), requires_grad=True)
    loss, _ = O.eval_sequence(scans, t(g['poses']), w, t(g['poses_exponent'].reshape(1, -1)), nbr, mask,
                              pose_deltas=pd, reduction='mean')
    loss.backward()
    np.testing.assert_allclose(npy(loss), g['poses_mineig_norm_loss'], rtol=1e-10)
    np.testing.assert_allclose(npy(pd.grad), g['poses_mineig_norm_grad_pose_deltas'], rtol=1e-8, atol=1e-14)
    w = torch.tensor(g['poly_w'].reshape(1, -1), requires_grad=True)
    loss, _ = O.eval_sequence(scans, t(g['poses']), w, t(g['poly_exponent'].reshape(1, -1)), nbr, mask,
                              model='Polynomial', reduction='mean')
    loss.backward()
    np.testing.assert_allclose(npy(loss), g['poly_mineig_norm_loss'], rtol=1e-10)
    np.testing.assert_allclose(npy(w.grad), g['poly_mineig_norm_grad_w'], rtol=1e-8)


def test_point_to_plane(golden):
    g = golden('icp_pairs')
    ns = int(g['n_scans'])
    w = torch.tensor(g['w'], requires_grad=True)
    pd = torch.tensor(g['pose_deltas'], requires_grad=True)
    T = torch.matmul(t(g['poses']), O.xyz_axis_angle_to_matrix(pd))
    pts, nrm = [], []
    for s in range(ns):
        d = O.model_apply(t(g['scan%d_depth' % s]), t(g['scan%d_inc_angles' % s]), t(g['scan%d_mask' % s]), w, t(g['exponent']))
        v, r, n = O.transform_cloud(t(g['scan%d_vps' % s]), t(g['scan%d_dirs' % s]), T[s], normals=t(g['scan%d_normals' % s]))
        pts.append(O.points_from(v, r, d)), nrm.append(n)
    masks = [(t(g['pair%d_mask1' % j]), t(g['pair%d_idx2' % j])) for j in range(ns - 1)]
    loss = O.point_to_plane(pts, nrm, masks)
    loss.backward()
    np.testing.assert_allclose(npy(loss), g['loss'], rtol=1e-9)
    np.testing.assert_allclose(npy(w.grad), g['grad_w'], rtol=1e-6)
    np.testing.assert_allclose(npy(pd.grad), g['grad_pose_deltas'], rtol=1e-6, atol=1e-12)


def test_point_to_point(golden):
    """loss.point_to_point_dist (loss.py:491-565) through icp_loss with model and pose corrections, and as a metric."""
    g = golden('icp_pairs')
    ns = int(g['n_scans'])
    w = torch.tensor(g['w'], requires_grad=True)
    pd = torch.tensor(g['pose_deltas'], requires_grad=True)
    T = torch.matmul(t(g['poses']), O.xyz_axis_angle_to_matrix(pd))
    pts, posed = [], []
    for s in range(ns):
        d = O.model_apply(t(g['scan%d_depth' % s]), t(g['scan%d_inc_angles' % s]), t(g['scan%d_mask' % s]), w, t(g['exponent']))
        v, r = O.transform_cloud(t(g['scan%d_vps' % s]), t(g['scan%d_dirs' % s]), T[s])
        pts.append(O.points_from(v, r, d))
        v0, r0 = O.transform_cloud(t(g['scan%d_vps' % s]), t(g['scan%d_dirs' % s]), t(g['poses'])[s])
        posed.append(O.points_from(v0, r0, t(g['scan%d_depth' % s])))
    masks = [(t(g['pair%d_mask1' % j]), t(g['pair%d_idx2' % j])) for j in range(ns - 1)]
    loss = O.point_to_point(pts, masks)
    loss.backward()
    np.testing.assert_allclose(npy(loss), g['p2p_loss'], rtol=1e-9)
    np.testing.assert_allclose(npy(w.grad), g['p2p_grad_w'], rtol=1e-6)
    np.testing.assert_allclose(npy(pd.grad), g['p2p_grad_pose_deltas'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(npy(O.point_to_point(posed, masks)), g['p2p_metric'], rtol=1e-9)
    # correspondences found as point_to_point_dist does without masks (1-NN, quantile inliers) give the same pairs
    found = []
    for j in range(ns - 1):
        m1, i2, _ = O.nn1_correspondences(npy(posed[j].float()), npy(posed[j + 1].float()), float(g['ratio']))
        found.append((t(m1), t(i2)))
    np.testing.assert_allclose(npy(O.point_to_point(posed, found)), g['p2p_metric_nn'], rtol=1e-6)


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_shadow_filter(golden, tag):
    """filters.filter_shadow_points (filters.py:257-309) on the fixture's direction neighbourhoods."""
    g = golden('shadow')
    depth = t(g[tag + '_depth'])
    pts = t(g[tag + '_vps']) + depth * t(g[tag + '_dirs'])
    mask, ang = O.shadow_mask(pts, t(g[tag + '_vps']), t(g[tag + '_dir_neighbors']).long(),
                              [float(np.radians(float(g['bounds_deg']))), float('inf')])
    assert np.array_equal(npy(mask), g[tag + '_mask'])
    valid = g[tag + '_dir_neighbors'] >= 0
    a = np.where(valid, npy(ang), 10.0)
    np.testing.assert_allclose(a.min(-1), g[tag + '_angle_min'], rtol=0, atol=0)
    # the direction neighbourhoods themselves: radius search on the unit directions (depth_cloud.py:217-224)
    r = float(torch.sqrt(2. * (1. - torch.cos(torch.as_tensor(float(g['angle']))))))
    ind = O.radius_ckdtree(g[tag + '_dirs'][:4000].astype(np.float64), r)
    sub = g[tag + '_dir_neighbors'][:4000]
    # restricted to the first 4000 rays: neighbours beyond them are dropped from both sides
    for row_ref, row in zip(sub[:200], ind[:200]):
        assert sorted(x for x in row_ref if 0 <= x < 4000) == sorted(x for x in row if x >= 0)


@pytest.mark.parametrize('name', ['Linear', 'InvCos', 'ScaledInvCos'])
def test_other_models(golden, name):
    """Linear / InvCos / ScaledInvCos (model.py:113-146, 289-349) through one iteration on the room_k10 inputs."""
    g, m = golden('room_k10'), golden('models')
    scans = scans_from_golden(g, torch.float64)
    w = torch.tensor(m[name + '_w'], requires_grad=True)
    loss, f = O.eval_sequence(scans, t(g['poses']), w, torch.zeros_like(w), t(g['g_neighbors']).long(), t(g['g_mask']),
                              model=name, reduction='mean')
    loss.backward()
    np.testing.assert_allclose(npy(loss), m[name + '_loss'], rtol=1e-10)
    np.testing.assert_allclose(npy(w.grad), m[name + '_grad_w'], rtol=1e-6, atol=1e-12)
    d0 = O.model_apply(scans[0]['depth'], scans[0]['inc'], scans[0]['mask'], w.detach(), None, name)
    np.testing.assert_allclose(npy(d0), m[name + '_depth0'], rtol=1e-12)
