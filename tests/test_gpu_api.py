"""GPU parity through the reference-API layer (DepthCloud / preproc / eval / loss / train), written the way the
reference's own callers drive it (train.py:94-215 set-up, eval.py:85-112 iteration), against the goldens."""
import numpy as np
import pytest
import torch
from numpy.lib.recfunctions import unstructured_to_structured

import dc_oracle as O
from helpers import t, npy, scans_from_golden, assert_eigvals_close

pytestmark = pytest.mark.gpu


def _cfg(g, **kw):
    from depth_correction_amd.config import Config
    cfg = Config(nn_k=int(g['cfg_nn_k']), nn_r=None, min_depth=0.0, max_depth=float('inf'), grid_res=0.0,
                 min_valid_neighbors=int(g['cfg_min_valid_neighbors']),
                 eigenvalue_ratio_bounds=g['eigenvalue_ratio_bounds'].tolist(),
                 vp_dispersion_bounds=g['vp_dispersion_bounds'].tolist(), device='cuda:0')
    return cfg.from_dict(kw)


def _scan_arrays(g):
    return [unstructured_to_structured(np.ascontiguousarray(g['scan%d_xyz' % s]), names=['x', 'y', 'z'])
            for s in range(int(g['n_scans']))]


def _setup(g, cfg):
    """train.py:94-215 for one sequence."""
    from depth_correction_amd.preproc import (establish_neighborhoods, global_cloud, global_cloud_mask, local_feature_cloud)
    clouds = [local_feature_cloud(a, cfg) for a in _scan_arrays(g)]
    poses = torch.as_tensor(g['poses'], device=cfg.device)
    g0 = global_cloud(clouds=clouds, poses=poses)
    ns = establish_neighborhoods(cloud=g0, cfg=cfg)
    mask = global_cloud_mask(g0, g0.mask, cfg)
    return clouds, poses, g0, ns, mask


@pytest.mark.parametrize('name', ['c0_plane', 'room_k10'])
def test_setup_phase_golden(golden, name):
    g = golden(name)
    cfg = _cfg(g)
    clouds, poses, g0, ns, mask = _setup(g, cfg)
    for s, c in enumerate(clouds):
        assert c.neighbors.dtype == torch.int64 and np.array_equal(npy(c.neighbors), g['scan%d_neighbors' % s])
        assert_eigvals_close(npy(c.eigvals), g['scan%d_eigvals' % s], 1e-9)
        np.testing.assert_allclose(npy(c.inc_angles), g['scan%d_inc_angles' % s], rtol=0, atol=1e-7)
        np.testing.assert_allclose(npy(c.dirs), g['scan%d_dirs' % s], rtol=1e-14, atol=1e-15)
        assert np.array_equal(npy(c.mask), g['scan%d_mask' % s])
        assert c.weights.shape == (len(c), int(g['cfg_nn_k']), 1)
    np.testing.assert_allclose(npy(g0.points), g['g0_points'], rtol=1e-13, atol=1e-13)
    assert np.array_equal(npy(ns[0]), g['g_neighbors']) and ns[1].shape == ns[0].shape + (1,)
    assert_eigvals_close(npy(g0.eigvals), g['g0_eigvals'], 1e-9)
    assert np.array_equal(npy(mask), g['g_mask'])
    np.testing.assert_allclose(npy(g0.vp_dispersion()), g['g0_vp_dispersion'], rtol=1e-9, atol=1e-12)


VARIANTS = [('mineig_norm', 'min_eigval_loss', True, False), ('mineig_raw', 'min_eigval_loss', False, False),
            ('mineig_norm_sqrt', 'min_eigval_loss', True, True), ('trace', 'trace_loss', False, False),
            ('trace_sqrt', 'trace_loss', False, True)]


def test_linked_chain_over_several_sequences_equals_plain_trainer(golden):
    """Several sequences in ONE loss on one GPU (train.py:172-175; eval.py:85-112 pools their sums and counts): a chain over the
    sequences -- dc_sequence_step_linked: one launch per sequence and step, every launch first finishes the one before it (its
    rows onto the step's running sums; the first launch of a step completes the previous step: totals, Adam update) -- against the
    plain trainer (evaluate all -> sum -> dc_adam_step).  Sums are handed out one step late, as in a single-sequence chain; a flush
    in the middle; sequences with different masks and numbers of scans."""
    from depth_correction_amd.plan import SequencePlan, SequenceTrainer
    g = golden('room_k10')
    cfg = _cfg(g)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    mask2 = mask & (torch.arange(len(mask), device=mask.device) % 3 != 0)
    mk = lambda: [SequencePlan(clouds, poses, ns[0], mask), SequencePlan(clouds, poses, ns[0].clone(), mask2),
                  SequencePlan(clouds, poses, ns[0].clone(), mask)]
    plain = SequenceTrainer(mk(), g['w'], g['exponent'], 3 * [poses], lr=1e-2)
    chain = SequenceTrainer(mk(), g['w'], g['exponent'], 3 * [poses], lr=1e-2, chained=True)
    assert chain.linked and not chain.update_in_next and not plain.linked
    ref, got = [], []
    for it in range(9):
        ref.append(npy(plain.step()).copy())
        prev = npy(chain.step()).copy()
        assert chain.linked
        if it not in (0, 5):
            got.append(prev)
        if it == 4:
            got.append(npy(chain.flush()).copy())
            np.testing.assert_allclose(npy(chain.w), npy(plain.w), rtol=1e-10)
    got.append(npy(chain.flush()).copy())
    assert len(got) == 9
    for a, b in zip(got, ref):
        assert a[1] == b[1] == float(2 * mask.sum() + mask2.sum())
        np.testing.assert_allclose(a[0], b[0], rtol=1e-11)
        np.testing.assert_allclose(a[2:], b[2:], rtol=1e-9, atol=1e-12 * np.abs(b[2:]).max())
    np.testing.assert_allclose(npy(chain.w), npy(plain.w), rtol=1e-10)
    assert abs(npy(chain.w)[0] - g['w'][0]) > 1e-3


@pytest.mark.parametrize('fused', [True, False])
def test_iteration_golden(golden, fused):
    """eval_loss_clouds + backward for every loss variant, fused kernels and un-fused DepthCloud operators."""
    from depth_correction_amd.eval import eval_loss_clouds
    from depth_correction_amd.loss import create_loss
    from depth_correction_amd.model import ScaledPolynomial
    g = golden('room_k10')
    cfg = _cfg(g, fused=fused)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    for tag, loss_name, norm, sqrt in VARIANTS:
        cfg.loss = loss_name
        cfg.loss_kwargs.update(normalization=norm, sqrt=sqrt)
        model = ScaledPolynomial(w=g['w'].tolist(), exponent=g['exponent'].tolist(), device=cfg.device)
        loss, loss_clouds, poses_upd, feat = eval_loss_clouds([clouds], [poses], [None], [mask], [ns], model,
                                                              create_loss(cfg), cfg)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[tag + '_loss'], rtol=1e-9)
        np.testing.assert_allclose(npy(model.w.grad), g[tag + '_grad_w'], rtol=1e-7)
        np.testing.assert_allclose(npy(loss_clouds[0].loss)[g['g_mask']] if fused else npy(loss_clouds[0].loss),
                                   g[tag + '_pointwise'], rtol=1e-7, atol=1e-15)
        if tag == 'mineig_norm':
            assert_eigvals_close(npy(feat[0].eigvals), g['g_eigvals'], 1e-9)
            np.testing.assert_allclose(npy(feat[0].points), g['g_points'], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize('fused', [True, False])
def test_pose_corrections_golden(golden, fused):
    from depth_correction_amd.config import PoseCorrection
    from depth_correction_amd.eval import eval_loss_clouds
    from depth_correction_amd.loss import create_loss
    from depth_correction_amd.model import ScaledPolynomial
    g = golden('room_k10')
    cfg = _cfg(g, fused=fused, pose_correction=PoseCorrection.pose)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    for tag, loss_name in (('poses_mineig_norm', 'min_eigval_loss'), ('poses_trace', 'trace_loss')):
        cfg.loss = loss_name
        model = ScaledPolynomial(w=g['poses_w'].tolist(), exponent=g['poses_exponent'].tolist(), device=cfg.device)
        pd = torch.tensor(g['poses_pose_deltas'], device=cfg.device, requires_grad=True)
        loss, _, poses_upd, _ = eval_loss_clouds([clouds], [poses], [pd], [mask], [ns], model, create_loss(cfg), cfg)
        loss.backward()
        np.testing.assert_allclose(npy(poses_upd[0]), g['poses_poses_upd'], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(loss.item(), g[tag + '_loss'], rtol=1e-9)
        np.testing.assert_allclose(npy(model.w.grad), g[tag + '_grad_w'], rtol=1e-7)
        ref = g[tag + '_grad_pose_deltas']
        np.testing.assert_allclose(npy(pd.grad), ref, rtol=1e-6, atol=1e-9 * np.abs(ref).max())


def test_polynomial_and_two_sequences(golden):
    """A list of sequences: pointwise losses are pooled before the mean (batch_loss, loss.py:205-213)."""
    from depth_correction_amd.eval import eval_loss_clouds
    from depth_correction_amd.loss import create_loss
    from depth_correction_amd.model import Polynomial, ScaledPolynomial
    g = golden('room_k10')
    cfg = _cfg(g)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    model = Polynomial(w=g['poly_w'].tolist(), exponent=g['poly_exponent'].tolist(), device=cfg.device)
    loss, *_ = eval_loss_clouds([clouds], [poses], [None], [mask], [ns], model, create_loss(cfg), cfg)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['poly_mineig_norm_loss'], rtol=1e-9)
    np.testing.assert_allclose(npy(model.w.grad), g['poly_mineig_norm_grad_w'], rtol=1e-7)
    # second "sequence": the same scans with another mask
    mask2 = mask & (torch.arange(len(mask), device=mask.device) % 3 == 0)
    ns2 = (ns[0].clone(), ns[1])
    m = ScaledPolynomial(w=g['w'].tolist(), exponent=g['exponent'].tolist(), device=cfg.device)
    outs = {}
    for fused in (True, False):
        cfg.fused = fused
        m.zero_grad()
        loss, *_ = eval_loss_clouds([clouds, clouds], [poses, poses], [None, None], [mask, mask2], [ns, ns2], m,
                                    create_loss(cfg), cfg)
        loss.backward()
        outs[fused] = (loss.item(), npy(m.w.grad).copy())
    pw = g['mineig_norm_pointwise']
    sel = npy(mask2)[g['g_mask']]
    np.testing.assert_allclose(outs[True][0], np.concatenate([pw, pw[sel]]).mean(), rtol=1e-9)
    np.testing.assert_allclose(outs[True][0], outs[False][0], rtol=1e-10)
    np.testing.assert_allclose(outs[True][1], outs[False][1], rtol=1e-7)


def test_train_matches_oracle_adam_loop(golden, tmp_path):
    """train() for a few iterations against the same loop driven by the oracle + torch.optim.Adam on the CPU."""
    from depth_correction_amd.train import train
    g = golden('room_k10')
    cfg = _cfg(g, n_opt_iters=4, lr=5e-3, log_dir=str(tmp_path),
               model_kwargs={'w': g['w'].tolist(), 'exponent': g['exponent'].tolist()})
    ds = list(zip(_scan_arrays(g), g['poses']))
    seen = []

    class CB:
        def __getattr__(self, name):
            return lambda *a, **k: None

        def train_loss(self, it, model, clouds, pose_deltas, poses, masks, loss):
            seen.append((loss.item(), npy(model.w).copy()))
    best = train(cfg, callbacks=CB(), train_datasets=[ds], val_datasets=[ds])
    scans = scans_from_golden(g)
    w = torch.nn.Parameter(torch.tensor(g['w'].reshape(1, -1)))
    opt = torch.optim.Adam([w], lr=5e-3)
    for it in range(4):
        opt.zero_grad()
        loss, _ = O.eval_sequence(scans, t(g['poses']), w, t(g['exponent'].reshape(1, -1)), t(g['g_neighbors']).long(),
                                  t(g['g_mask']), reduction='mean')
        np.testing.assert_allclose(seen[it][0], loss.item(), rtol=1e-8)
        # (the bar on fp64 device data: 1e-9 on the loss, 1e-7 on gradients -- the staged rows carry float32 copies of u and c for the
        #  second sweep, 6e-8 per term; Adam normalises the gradient, so its error is the weights' error)
        np.testing.assert_allclose(seen[it][1], npy(w), rtol=1e-7)
        loss.backward()
        opt.step()
    assert best is not None and best.model_state_dict.endswith('_state_dict.pth')
    sd = torch.load(best.model_state_dict)
    assert list(sd) == ['w'] and sd['w'].shape == (1, 2)


def test_native_trainer_equals_autograd_adam(golden):
    """dc_sequence_eval + dc_adam_step (no Python in the loop) vs autograd Function + torch.optim.Adam."""
    from depth_correction_amd.plan import SequencePlan, SequenceTrainer, consistency_loss
    g = golden('room_k10')
    cfg = _cfg(g)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    plan = SequencePlan(clouds, poses, ns[0], mask)
    tr = SequenceTrainer([plan], g['w'], g['exponent'], [poses], lr=1e-2)
    w = torch.nn.Parameter(torch.tensor(g['w'].reshape(1, -1), device='cuda:0'))
    e = torch.tensor(g['exponent'].reshape(1, -1), device='cuda:0')
    opt = torch.optim.Adam([w], lr=1e-2)
    for it in range(5):
        opt.zero_grad()
        s, cnt = consistency_loss(plan, w, e, poses)
        (s / cnt).backward()
        acc = tr.step()
        np.testing.assert_allclose(npy(acc)[0], s.item(), rtol=1e-13)
        np.testing.assert_allclose(npy(acc)[2:] / cnt, npy(w.grad).ravel(), rtol=1e-12)
        opt.step()
        np.testing.assert_allclose(npy(tr.w), npy(w).ravel(), rtol=1e-12)
    assert abs(npy(tr.w)[0] - g['w'][0]) > 1e-3


@pytest.mark.parametrize('ragged', [False, True])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_chained_steps_equal_ordinary_steps(golden, dtype, ragged):
    """SequenceTrainer(chained=True): every launch also finishes the previous step (dc_sequence_step_chained), so a step is
    one launch instead of two.  step() hands out the sums of the PREVIOUS evaluation, flush() those of the last one; weights,
    losses and gradients follow the ordinary trainer's to the order of the fp64 additions (one partial row per block instead
    of one per wavefront), for a chain that is flushed in the middle and continued, with an unrelated evaluation of the same
    sequence in between (the chain's rows live behind the columns ordinary evaluations write)."""
    from depth_correction_amd.plan import SequencePlan, SequenceTrainer
    g = golden('room_k10')
    cfg = _cfg(g, float_type='float64' if dtype == torch.float64 else 'float32')
    clouds, poses, _, ns, mask = _setup(g, cfg)
    nbr = ns[0]
    if ragged:                                                 # a radius-style table: 13 columns, missing entries -> run-time slots
        gen = torch.Generator(device='cpu').manual_seed(3)
        drop = (torch.rand(nbr.shape, generator=gen) < 0.2).to(nbr.device)
        drop[:, :4] = False
        nbr = torch.where(drop, torch.full_like(nbr, -1), nbr)
        nbr = torch.cat([nbr, torch.full_like(nbr[:, :3], -1)], dim=1).contiguous()
    plan = SequencePlan(clouds, poses, nbr, mask)
    plain = SequenceTrainer([plan], g['w'], g['exponent'], [poses], lr=1e-2)
    ref = [npy(plain.step()).copy() for _ in range(12)]
    chain = SequenceTrainer([plan], g['w'], g['exponent'], [poses], lr=1e-2, chained=True)
    got = []
    scratch = torch.zeros((2 + 4 + 12 * plan.n_scans,), dtype=torch.float64, device='cuda:0')
    for it in range(12):
        prev = npy(chain.step()).copy()
        assert chain.chained                                  # this plan / model can chain
        if it == 4:                                            # another evaluation of the same sequence between two chained steps
            plan.eval_native(torch.tensor([5e-3, -1e-3], dtype=torch.float64, device='cuda:0'), chain.exponent, chain.poses12[0], scratch)
        if it in (1, 2, 3, 4, 5, 6, 8, 9, 10, 11):            # (not right after a flush: nothing is pending then)
            got.append(prev)
        if it in (6, 11):
            got.append(npy(chain.flush()).copy())             # finishes evaluation it + 1 (1-based)
    torch.cuda.synchronize()
    assert len(got) == 12 and chain.t == 12 and int(chain.ready[1]) == 12          # (the published words carry the launch's number)
    for a, b in zip(got, ref):
        assert a[1] == b[1]
        np.testing.assert_allclose(a[0], b[0], rtol=1e-11)
        np.testing.assert_allclose(a[2:], b[2:], rtol=1e-9, atol=1e-12 * np.abs(b[2:]).max())
    np.testing.assert_allclose(npy(chain.w), npy(plain.w), rtol=1e-10)
    assert abs(npy(chain.w)[0] - g['w'][0]) > 1e-3
    # a plan that cannot chain (no basis form: the general three-kernel path) falls back to ordinary steps on its own
    general = SequencePlan(clouds, poses, nbr, mask, basis=False)
    a = SequenceTrainer([general], g['w'], g['exponent'], [poses], lr=1e-2, chained=True)
    b = SequenceTrainer([general], g['w'], g['exponent'], [poses], lr=1e-2)
    for _ in range(4):
        sa, sb = npy(a.step()).copy(), npy(b.step()).copy()
        assert not a.chained and np.array_equal(sa, sb)
    a.flush()
    assert np.array_equal(npy(a.w), npy(b.w)) and abs(npy(a.w)[0] - g['w'][0]) > 1e-3


def test_update_in_next_launch_equals_separate_adam(golden):
    """The multi-rank form of a step (evaluation -> reduction -> all-reduce -> Adam) with the Adam update moved into the next
    evaluation's launch (dc_sequence_eval_after_update): the sums of every step are the ordinary ones (current, not lagged),
    the weights after flush() too.  Run on one rank (the all-reduce is the identity there)."""
    from depth_correction_amd.plan import SequencePlan, SequenceTrainer
    g = golden('room_k10')
    cfg = _cfg(g)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    plan = SequencePlan(clouds, poses, ns[0], mask)
    plain = SequenceTrainer([plan], g['w'], g['exponent'], [poses], lr=1e-2, distributed=True)
    moved = SequenceTrainer([plan], g['w'], g['exponent'], [poses], lr=1e-2, distributed=True, chained=True)
    assert moved.update_in_next and not moved.chained and not plain.update_in_next
    for it in range(9):
        a, b = npy(moved.step()).copy(), npy(plain.step()).copy()
        assert moved.update_in_next and a[1] == b[1]
        np.testing.assert_allclose(a[0], b[0], rtol=1e-11)
        np.testing.assert_allclose(a[2:], b[2:], rtol=1e-9, atol=1e-12 * np.abs(b[2:]).max())
        if it == 4:
            moved.flush()                                      # a flush in the middle: the chain simply continues
            np.testing.assert_allclose(npy(moved.w), npy(plain.w), rtol=1e-10)
    moved.flush()
    np.testing.assert_allclose(npy(moved.w), npy(plain.w), rtol=1e-10)
    assert abs(npy(moved.w)[0] - g['w'][0]) > 1e-3


@pytest.mark.parametrize('fused', [True, False])
def test_icp_loss_golden(golden, fused):
    from depth_correction_amd.depth_cloud import DepthCloud
    from depth_correction_amd.loss import icp_loss
    from depth_correction_amd.model import ScaledPolynomial
    from depth_correction_amd.transform import xyz_axis_angle_to_matrix
    g = golden('icp_pairs')
    dev = 'cuda:0'
    ns = int(g['n_scans'])
    clouds = []
    for s in range(ns):
        c = DepthCloud(t(g['scan%d_vps' % s], dev), t(g['scan%d_dirs' % s], dev), t(g['scan%d_depth' % s], dev),
                       inc_angles=t(g['scan%d_inc_angles' % s], dev), mask=t(g['scan%d_mask' % s], dev),
                       normals=t(g['scan%d_normals' % s], dev))
        clouds.append(c)
    masks = [(t(g['pair%d_mask1' % j], dev), t(g['pair%d_idx2' % j], dev)) for j in range(ns - 1)]
    model = ScaledPolynomial(w=g['w'].reshape(-1).tolist(), exponent=g['exponent'].reshape(-1).tolist(), device=dev)
    if not fused:
        model.kernel_kind = None            # forces the reference-style tensor path of icp_loss
    pd = torch.tensor(g['pose_deltas'], device=dev, requires_grad=True)
    poses = torch.matmul(t(g['poses'], dev), xyz_axis_angle_to_matrix(pd))
    loss, _ = icp_loss([clouds], [poses], model, masks=[masks], icp_point_to_plane=True, icp_inlier_ratio=float(g['ratio']))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-6)
    np.testing.assert_allclose(npy(model.w.grad), g['grad_w'], rtol=1e-5)
    ref = g['grad_pose_deltas']
    np.testing.assert_allclose(npy(pd.grad), ref, rtol=1e-5, atol=1e-7 * np.abs(ref).max())


def _icp_golden_clouds(g, dev, normals=True):
    from depth_correction_amd.depth_cloud import DepthCloud
    clouds = []
    for s in range(int(g['n_scans'])):
        kw = dict(inc_angles=t(g['scan%d_inc_angles' % s], dev), mask=t(g['scan%d_mask' % s], dev))
        if normals:
            kw['normals'] = t(g['scan%d_normals' % s], dev)
        clouds.append(DepthCloud(t(g['scan%d_vps' % s], dev), t(g['scan%d_dirs' % s], dev), t(g['scan%d_depth' % s], dev), **kw))
    masks = [(t(g['pair%d_mask1' % j], dev), t(g['pair%d_idx2' % j], dev)) for j in range(int(g['n_scans']) - 1)]
    return clouds, masks


@pytest.mark.parametrize('fused', [True, False])
def test_icp_point_to_point_golden(golden, fused):
    """icp_loss(icp_point_to_plane=False) (loss.py:373-403 -> point_to_point_dist :491-565) with model and per-pose
    corrections against the live-reference fixture: loss, dL/dw, dL/dpose_deltas -- through dc_p2point_sequence (fused)
    and through the tensor expressions."""
    from depth_correction_amd.loss import icp_loss
    from depth_correction_amd.model import ScaledPolynomial
    from depth_correction_amd.transform import xyz_axis_angle_to_matrix
    g = golden('icp_pairs')
    dev = 'cuda:0'
    clouds, masks = _icp_golden_clouds(g, dev, normals=False)       # point to point needs no normals
    model = ScaledPolynomial(w=g['w'].reshape(-1).tolist(), exponent=g['exponent'].reshape(-1).tolist(), device=dev)
    if not fused:
        model.kernel_kind = None
    pd = torch.tensor(g['pose_deltas'], device=dev, requires_grad=True)
    poses = torch.matmul(t(g['poses'], dev), xyz_axis_angle_to_matrix(pd))
    loss, loss_clouds = icp_loss([clouds], [poses], model, masks=[masks], icp_point_to_plane=False,
                                 icp_inlier_ratio=float(g['ratio']))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['p2p_loss'], rtol=1e-6)
    np.testing.assert_allclose(npy(model.w.grad), g['p2p_grad_w'], rtol=1e-5)
    ref = g['p2p_grad_pose_deltas']
    np.testing.assert_allclose(npy(pd.grad), ref, rtol=1e-5, atol=1e-7 * np.abs(ref).max())
    # the returned loss cloud is the corrected, posed sequence (loss.py:396-398), not the sensor-frame input
    lc = loss_clouds[0]
    assert len(lc) == sum(len(c) for c in clouds)
    moved = torch.cat([model(c).transform(p).to_points() for c, p in zip(clouds, poses.detach())])
    np.testing.assert_allclose(npy(lc.to_points()), npy(moved.detach()), rtol=1e-12, atol=1e-12)


def test_point_to_point_metric_golden(golden):
    """point_to_point_dist as the map-accuracy metric (scripts/model_poses_learning:142-146) on GPU clouds: one
    dc_p2point_sequence call, with given correspondences and with the GPU 1-NN builder finding them."""
    from depth_correction_amd.loss import point_to_point_dist
    g = golden('icp_pairs')
    dev = 'cuda:0'
    clouds, masks = _icp_golden_clouds(g, dev, normals=False)
    posed = [c.transform(p) for c, p in zip(clouds, t(g['poses'], dev))]
    d = point_to_point_dist(posed, icp_inlier_ratio=float(g['ratio']), masks=masks)
    np.testing.assert_allclose(d.item(), g['p2p_metric'], rtol=1e-6)
    d = point_to_point_dist(posed, icp_inlier_ratio=float(g['ratio']), differentiable=False)
    np.testing.assert_allclose(d.item(), g['p2p_metric_nn'], rtol=1e-6)
    # plain point tensors are accepted like clouds (loss.py:512-519)
    d = point_to_point_dist([c.to_points() for c in posed], icp_inlier_ratio=float(g['ratio']), masks=masks)
    np.testing.assert_allclose(d.item(), g['p2p_metric'], rtol=1e-6)


@pytest.mark.parametrize('tag,dtype', [('f64', torch.float64), ('f32', torch.float32)])
def test_shadow_filter_golden(golden, tag, dtype):
    """The online node's pre-processing (preproc.py:44-47): depth filter, direction neighbourhoods (radius search on unit
    directions) and the scan-shadow filter (dc_shadow_mask) against the live-reference fixture.  fp64 clouds: identical
    mask; fp32 clouds: identical except where an extreme angle lies within 1e-5 rad of a bound (acos in fp32 on the CPU
    and on the GPU differ in the last ulp)."""
    from depth_correction_amd.depth_cloud import DepthCloud
    from depth_correction_amd.filters import filter_depth, filter_shadow_points
    g = golden('shadow')
    dev = 'cuda:0'
    dc = DepthCloud(t(g[tag + '_vps'], dev), t(g[tag + '_dirs'], dev), t(g[tag + '_depth'], dev))
    assert dc.depth.dtype == dtype
    assert bool(filter_depth(dc, min=1.0, max=25.0, only_mask=True).all())           # the fixture holds the filtered scan
    dc.update_points()
    dc.update_dir_neighbors(angle=float(g['angle']))
    assert np.array_equal(npy(dc.dir_neighbors), g[tag + '_dir_neighbors'])
    bounds = [float(np.radians(float(g['bounds_deg']))), float('inf')]
    mask = npy(filter_shadow_points(dc, list(bounds), only_mask=True))
    want = g[tag + '_mask']
    lo = np.float32(bounds[0])
    near = (np.abs(g[tag + '_angle_min'] - lo) < 1e-5) | (np.abs(g[tag + '_angle_max'] - np.float32(np.pi)) < 1e-5)
    if dtype == torch.float64:
        near &= False
    assert np.array_equal(mask[~near], want[~near]) and near.sum() < 20
    assert 0.5 < mask.mean() < 0.95
    kept = filter_shadow_points(dc, list(bounds))
    assert len(kept) == int(mask.sum())


def test_icp_correspondences_vs_ckdtree(golden):
    from depth_correction_amd.loss import icp_correspondences
    g = golden('room_k10')
    a, b = g['scan0_xyz'] + g['poses'][0, :3, 3], g['scan1_xyz'] + g['poses'][1, :3, 3]
    m1, i2, _ = icp_correspondences(t(a, 'cuda:0'), t(b, 'cuda:0'), 0.3)
    rm1, ri2, _ = O.nn1_correspondences(a, b, 0.3)
    assert np.array_equal(npy(m1), rm1) and np.array_equal(npy(i2), ri2)


def test_nearest_neighbors_api(golden):
    from depth_correction_amd.nearest_neighbors import nearest_neighbors
    g = golden('knn')
    p = t(g['points'], 'cuda:0')
    d, i = nearest_neighbors(p, p, k=10)
    assert i.dtype == torch.int64 and d.dtype == torch.float64 and i.device == p.device
    assert np.array_equal(npy(i), g['k10_ind']) and np.array_equal(npy(d), g['k10_dist'])
    d, i = nearest_neighbors(p, p, k=8, r=0.15)
    assert np.array_equal(npy(i), g['k8_r015_ind'])
    d, i = nearest_neighbors(p, p, r=0.12)
    assert d is None and np.array_equal(npy(i), g['r012_ind'])
    # radius search of ANOTHER cloud's points (nearest_neighbors.py:50-51: query_ball_point takes any query) against cKDTree
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(12)
    qn = (g['points'][rng.choice(len(g['points']), 700, replace=False)] + rng.normal(size=(700, 3)) * 0.05).astype(np.float32).astype(np.float64)
    qn[5] = [50.0, 50.0, 50.0]                                  # a query with no neighbour at all
    d, i = nearest_neighbors(p, t(qn, 'cuda:0'), r=0.2)
    lists = cKDTree(g['points']).query_ball_point(qn, 0.2)
    kmax = max(len(x) for x in lists)
    ref = np.array([sorted(x) + (kmax - len(x)) * [-1] for x in lists])
    assert d is None and i.dtype == torch.int64 and np.array_equal(npy(i), ref) and (ref[5] == -1).all() and kmax > 20
    dc_r = t(g['points'][:500], 'cuda:0')
    from depth_correction_amd.depth_cloud import DepthCloud
    cloud = DepthCloud.from_points(dc_r)
    cloud.update_all(r=0.25)                                   # radius neighbourhoods: -1 padded, weights 0 there
    f = O.features(cloud.points.cpu(), cloud.neighbors.cpu(), cloud.dirs.cpu())
    keep = npy((cloud.neighbors >= 0).sum(1) >= 3)
    assert_eigvals_close(npy(cloud.eigvals)[keep], npy(f['eigvals'])[keep], 1e-9)
    assert torch.equal(cloud.weights[..., 0] > 0, cloud.neighbors >= 0)


def test_chamfer_metric_vs_ckdtree(golden):
    """metrics.chamfer_distance (map accuracy, scripts/model_poses_learning:142-146) on the GPU 1-NN builder."""
    from scipy.spatial import cKDTree
    from depth_correction_amd.metrics import chamfer_distance
    g = golden('room_k10')
    a, b = g['scan0_xyz'] + g['poses'][0, :3, 3], g['scan2_xyz'] + g['poses'][2, :3, 3]
    ref = cKDTree(b).query(a, k=1)[0]
    got = chamfer_distance(t(a, 'cuda:0'), t(b, 'cuda:0'))
    np.testing.assert_allclose(got.item(), ref.mean(), rtol=1e-14)
    per = chamfer_distance(t(a, 'cuda:0'), t(b, 'cuda:0'), apply_point_reduction=False)
    assert np.array_equal(npy(per), ref)
    batch = chamfer_distance([t(a, 'cuda:0'), t(b, 'cuda:0')], [t(b, 'cuda:0'), t(a, 'cuda:0')], batch_reduction='sum')
    np.testing.assert_allclose(batch.item(), ref.mean() + cKDTree(a).query(b, k=1)[0].mean(), rtol=1e-13)


def test_shadow_filter_and_dir_neighbors(golden):
    """update_dir_neighbors (radius search on unit directions) + filter_shadow_points against a plain tensor
    restatement of filters.py:257-309 with cKDTree neighbours."""
    from scipy.spatial import cKDTree
    from depth_correction_amd.depth_cloud import DepthCloud
    from depth_correction_amd.filters import filter_shadow_points
    from depth_correction_amd.nearest_neighbors import ball_angle_to_distance
    g = golden('room_k10')
    xyz = g['scan1_xyz'][:1500]
    cloud = DepthCloud.from_points(t(xyz, 'cuda:0'))
    angle = 0.06
    cloud.update_dir_neighbors(angle=angle)
    dirs = npy(cloud.dirs)
    r = float(ball_angle_to_distance(torch.as_tensor(angle)))
    lists = cKDTree(dirs).query_ball_point(dirs, r)
    kmax = max(map(len, lists))
    ref = np.full((len(dirs), kmax), -1)
    for i, l in enumerate(lists):
        ref[i, :len(l)] = sorted(l)
    assert np.array_equal(npy(cloud.dir_neighbors), ref)
    bounds = [0.3, float('inf')]
    mask = npy(filter_shadow_points(cloud, list(bounds), only_mask=True))
    x = xyz
    nb = np.where(ref >= 0, ref, 0)
    ox, nx = (0 - x)[:, None, :], x[nb] - x[:, None, :]
    cos = (ox * nx).sum(-1) / np.maximum(np.linalg.norm(ox, axis=-1) * np.linalg.norm(nx, axis=-1), 1e-8)
    with np.errstate(invalid='ignore'):
        a = np.arccos(cos)
    a[ref < 0] = 0.5 * (0.3 + np.pi)
    want = (np.nanmin(a, -1) >= 0.3) & (np.nanmax(a, -1) <= np.pi)
    nan_rows = np.isnan(a).any(-1)
    assert np.array_equal(mask[~nan_rows], want[~nan_rows]) and 0 < mask.sum() < len(mask)
    kept = filter_shadow_points(cloud, list(bounds))
    assert len(kept) == int(mask.sum())


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_voxel_grid_filter_gpu_golden(golden, dtype):
    """dc_voxel_filter against the reference's filter_grid output: same survivors in the same order for every keep mode."""
    from depth_correction_amd.depth_cloud import DepthCloud
    from depth_correction_amd.filters import filter_grid
    g = golden('grid')
    pts = t(g['points'], 'cuda:0', dtype)               # fixture points are fp32-representable
    for keep in ('first', 'last', 'random'):
        for po in (False, True):
            ind = filter_grid(pts, float(g['grid_res']), only_mask=True, keep=keep, preserve_order=po,
                              rng=np.random.default_rng(135))
            assert np.array_equal(np.asarray(ind), g['%s_%d' % (keep, po)]), (keep, po)
    cloud = DepthCloud.from_points(pts)
    kept = filter_grid(cloud, float(g['grid_res']), keep='random', rng=np.random.default_rng(135))
    assert len(kept) == len(g['random_0'])
    np.testing.assert_allclose(npy(kept.to_points()), g['points'][g['random_0']], rtol=1e-6, atol=1e-6)
    # host arrays (the datasets' structured clouds) are filtered on the GPU too and indexed on the host
    host = g['points'].astype(np.float32 if dtype == torch.float32 else np.float64)
    rec = np.zeros(len(host), dtype=[('x', host.dtype), ('y', host.dtype), ('z', host.dtype), ('i', 'i4')])
    rec['x'], rec['y'], rec['z'], rec['i'] = host[:, 0], host[:, 1], host[:, 2], np.arange(len(host))
    for keep in ('first', 'random'):
        ind = filter_grid(host, float(g['grid_res']), only_mask=True, keep=keep, rng=np.random.default_rng(135))
        assert np.array_equal(np.asarray(ind), g['%s_0' % keep])
        kept = filter_grid(rec, float(g['grid_res']), keep=keep, rng=np.random.default_rng(135))
        assert np.array_equal(kept['i'], g['%s_0' % keep])


def test_unfused_options_match_oracle(golden):
    """Options outside the fused kernels: distance-scaled neighbour weights (nn_scale, depth_cloud.py:356-364), loss
    offsets and quantile inliers (loss.py:256-281) -- forward values through the un-fused operators vs the oracle."""
    from depth_correction_amd.depth_cloud import DepthCloud
    from depth_correction_amd.loss import min_eigval_loss, trace_loss
    g = golden('room_k10')
    x = t(g['g0_points'], 'cuda:0')
    cloud = DepthCloud.from_points(x)
    cloud.update_all(k=int(g['cfg_nn_k']), scale=0.05)
    f = O.features(t(g['g0_points']), t(g['g_neighbors']).long(), cloud.dirs.cpu(), scale=0.05)
    np.testing.assert_allclose(npy(cloud.weights), npy(f['weights']), rtol=1e-12)
    np.testing.assert_allclose(npy(cloud.mean), npy(f['mean']), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(npy(cloud.cov), npy(f['cov']), rtol=1e-9, atol=1e-16)
    assert_eigvals_close(npy(cloud.eigvals), npy(f['eigvals']), 1e-9)
    # quantile inliers + offsets on the GPU tensors
    ev = f['eigvals']
    loss, lc = min_eigval_loss(cloud, normalization=True, inlier_ratio=0.7)
    raw = ev[:, 0] / ev.sum(-1).clamp(min=1e-6)
    keep = raw <= torch.quantile(raw, 0.7)
    np.testing.assert_allclose(loss.item(), raw[keep].mean().item(), rtol=1e-9)
    assert len(lc) == int(keep.sum())
    off = torch.full((len(cloud),), 1e-4, dtype=torch.float64, device='cuda:0')
    loss, _ = trace_loss(cloud, offset=off)
    np.testing.assert_allclose(loss.item(), torch.relu(O.trace(f['cov']) - 1e-4).mean().item(), rtol=1e-9)
    # and the fused path refuses nothing silently: eval_loss_clouds routes nn_scale configs to the un-fused operators
    from depth_correction_amd.config import Config
    from depth_correction_amd.eval import fused_supported
    from depth_correction_amd.model import ScaledPolynomial
    cloud.inc_angles = cloud.inc_angles
    cfg = Config(nn_k=10, nn_r=None, nn_scale=0.05)
    assert not fused_supported([[cloud]], ScaledPolynomial(w=[0.0, 0.0], exponent=[2.0, 4.0]), cfg)
    cfg = Config(nn_k=10, nn_r=None)
    assert fused_supported([[cloud]], ScaledPolynomial(w=[0.0, 0.0], exponent=[2.0, 4.0]), cfg)
    cfg.loss_kwargs['inlier_ratio'] = 0.5
    assert fused_supported([[cloud]], None, cfg)             # quantile inliers are gated inside the fused path
    cfg.loss_offset = True
    assert not fused_supported([[cloud]], None, cfg)
    cfg.loss_offset = False
    cfg.loss_kwargs.update(inlier_ratio=1.0, skip_nans=True)
    assert fused_supported([[cloud]], None, cfg)             # NaN-dropping reductions too (round 4)


@pytest.mark.parametrize('policy', ['skip_nans', 'only_finite'])
@pytest.mark.parametrize('form', ['one_pass', 'general'])
def test_nan_dropping_reductions_in_the_fused_kernels(golden, policy, form):
    """loss.py:125-137 inside the fused plan (round 4): pointwise losses that are NaN (here: neighbourhoods without a single valid
    member, W = 0 -> NaN mean) are left out of the sum, the count and the gradients -- by the one-pass kernel and by the
    forward / backward pair -- as the oracle's masked mean over the finite entries has it; without the policy the loss is NaN."""
    from depth_correction_amd import _native as nv
    from depth_correction_amd.plan import SequencePlan, consistency_loss
    g = golden('room_k10')
    cfg = _cfg(g, float_type='float64')
    clouds, poses, _, ns, mask = _setup(g, cfg)
    nbr = ns[0].clone()
    dead = torch.nonzero(mask).reshape(-1)[::97][:40]                 # masked-in centres that lose every neighbour
    nbr[dead] = -1
    w = torch.nn.Parameter(t(g['w'], 'cuda:0').reshape(1, -1).clone())
    e = t(g['exponent'], 'cuda:0').reshape(1, -1)
    nv.check(nv.lib().dc_set_option(3, 1 if form == 'general' else 0), 'dc_set_option')
    try:
        plain = SequencePlan(clouds, poses, nbr, mask)
        s0, c0 = consistency_loss(plain, w, e, poses)
        assert torch.isnan(s0) and c0 == float(mask.sum())
        plan = SequencePlan(clouds, poses, nbr, mask, nan_policy=policy)
        s, cnt = consistency_loss(plan, w, e, poses)
        (s / cnt).backward()
    finally:
        nv.check(nv.lib().dc_set_option(3, 0), 'dc_set_option')
    assert int(cnt) == int(mask.sum()) - len(dead)
    # oracle: the mean over the kept entries.  (LAPACK refuses the NaN covariances of the emptied neighbourhoods -- eigh raises,
    # in the reference too -- so the oracle gets the untouched table and the emptied centres masked out: the same set of terms)
    oc = [dict(vps=c.vps.cpu(), dirs=c.dirs.cpu(), depth=c.depth.cpu(), inc=c.inc_angles.cpu(), mask=c.mask.cpu()) for c in clouds]
    wo = torch.tensor(g['w'].reshape(1, -1), dtype=torch.float64, requires_grad=True)
    kept = mask.clone()
    kept[dead] = False
    lo, _ = O.eval_sequence(oc, poses.cpu(), wo, e.cpu(), ns[0].long().cpu(), kept.cpu(), reduction='none')
    assert int(lo.numel()) == int(cnt) and bool(lo.isfinite().all())
    ref = lo.mean()
    ref.backward()
    np.testing.assert_allclose((s / cnt).item(), ref.item(), rtol=1e-9)
    np.testing.assert_allclose(npy(w.grad).ravel(), npy(wo.grad).ravel(), rtol=1e-7, atol=1e-12 * np.abs(npy(wo.grad)).max())


def test_icp_training_matches_oracle_loop(tmp_path):
    """BASELINE config 4 shape, reduced: KITTI-like ring scans, depth + voxel-grid pre-filter, radius neighbourhoods
    (nn_r = 0.4), point-to-plane ICP loss, model weights AND per-pose corrections optimised by train(); every iteration's
    loss and parameters against the same loop on the CPU oracle (cKDTree, fp64, torch.optim.Adam)."""
    from depth_correction_amd.config import Config, Loss, PoseCorrection
    from depth_correction_amd.dataset import KittiLikeDataset
    from depth_correction_amd.preproc import filtered_cloud
    from depth_correction_amd.train import train
    cfg = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0,
                 max_depth=12.0, vp_dispersion_bounds=[], n_opt_iters=3, lr=2e-3, log_dir=str(tmp_path),
                 model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]})
    # 96 rings: every 0.4 m neighbourhood spans several rings, so no neighbourhood is collinear (for a collinear one
    # lambda0 = lambda1 and the normal is whatever vector of that eigenspace LAPACK happens to return)
    ds = KittiLikeDataset(n_poses=3, n_rings=96, n_azimuth=384)
    seq = [(filtered_cloud(cloud, cfg), pose) for cloud, pose in ds]
    assert all(5000 < len(c) < 96 * 384 for c, _ in seq)
    seen = []

    class CB:
        def __getattr__(self, name):
            return lambda *a, **k: None

        def train_loss(self, it, model, clouds, pose_deltas, poses, masks, loss):
            seen.append((loss.item(), npy(model.w).copy(), npy(pose_deltas[0]).copy()))
            if it == 0:          # the feature clouds the reference hands to its callbacks (eval.py:90-98), built on demand
                assert len(clouds) == 1 and len(clouds[0]) == sum(len(c) for c, _ in seq)
                assert clouds[0].eigvals.shape == (len(clouds[0]), 3) and clouds[0].points.is_cuda
    train(cfg, callbacks=CB(), train_datasets=[seq], val_datasets=[])

    # ---- the same on the oracle ----
    scans, poses = [], torch.as_tensor(np.stack([p for _, p in seq]))
    for cloud, _ in seq:
        pts = torch.as_tensor(np.stack([cloud[f] for f in 'xyz'], 1).astype(np.float64))
        depth = pts.norm(dim=-1, keepdim=True)
        dirs = pts / depth
        nbr = torch.as_tensor(O.radius_ckdtree(pts.numpy(), 0.4))
        f = O.features(pts, nbr, dirs)
        ev = f['eigvals']
        assert bool(((ev[:, 1] - ev[:, 0]) > 1e-6 * ev[:, 2]).all())                 # well-defined normals everywhere
        scans.append(dict(vps=torch.zeros_like(pts), dirs=dirs, depth=depth, inc=f['inc_angles'], normals=f['normals'],
                          mask=O.local_mask(f['eigvals'], None, cfg.eigenvalue_ratio_bounds)))
    masks = []
    for j in range(len(scans) - 1):
        pj = [O.points_from(*O.transform_cloud(s['vps'], s['dirs'], T), s['depth']) for s, T in ((scans[j], poses[j]), (scans[j + 1], poses[j + 1]))]
        m1, i2, _ = O.nn1_correspondences(pj[0].numpy(), pj[1].numpy(), 0.3)
        masks.append((torch.as_tensor(m1), torch.as_tensor(i2)))
    w = torch.nn.Parameter(torch.tensor([[1e-3, -1e-3]], dtype=torch.float64))
    e = torch.tensor([[2.0, 4.0]], dtype=torch.float64)
    pd = torch.nn.Parameter(torch.zeros((3, 6), dtype=torch.float64))
    opt = torch.optim.Adam([{'params': [w], 'lr': 2e-3}, {'params': [pd], 'lr': 2e-3}])
    for it in range(3):
        opt.zero_grad()
        T = torch.matmul(poses, O.xyz_axis_angle_to_matrix(pd))
        pts, nrm = [], []
        for s, Ts in zip(scans, T):
            d = O.model_apply(s['depth'], s['inc'], s['mask'], w, e)
            v, r, n = O.transform_cloud(s['vps'], s['dirs'], Ts, normals=s['normals'])
            pts.append(O.points_from(v, r, d)), nrm.append(n)
        loss = O.point_to_plane(pts, nrm, masks)
        np.testing.assert_allclose(seen[it][0], loss.item(), rtol=1e-5)
        np.testing.assert_allclose(seen[it][1], npy(w), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(seen[it][2], npy(pd), rtol=1e-4, atol=1e-8)
        loss.backward()
        pd.grad[0].zero_()
        opt.step()
    assert np.abs(seen[-1][2][1:]).max() > 1e-3 and np.all(seen[-1][2][0] == 0)     # poses moved, the first stays fixed


def test_active_only_plan_is_exact(golden):
    """Evaluating only the masked points as centres changes nothing in the loss, the count or any gradient."""
    from depth_correction_amd.plan import SequencePlan
    g = golden('room_k10')
    cfg = _cfg(g)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    dev = poses.device
    w = torch.tensor(g['w'].reshape(-1), device=dev)
    e = torch.tensor(g['exponent'].reshape(-1), device=dev)
    outs = []
    for active in (False, True):
        plan = SequencePlan(clouds, poses, ns[0], mask, active_only=active)
        out = torch.zeros(2 + 4 + 12 * plan.n_scans, dtype=torch.float64, device=dev)
        plan.eval_native(w, e, plan.poses12(poses), out, want_exponent=True, want_pose=True)
        outs.append(npy(out))
        assert plan.count == int(g['g_mask'].sum())
        full = plan.forward(w, e, poses, want_pointwise=True, want_eigvals=True)
        assert_eigvals_close(npy(plan.unpermute(full['eigvals'])), g['g_eigvals'], 1e-9)
    np.testing.assert_allclose(outs[1][:2], outs[0][:2], rtol=1e-13)
    np.testing.assert_allclose(outs[1][2:], outs[0][2:], rtol=1e-10, atol=1e-12 * np.abs(outs[0][2:]).max())
    M = g['g_mask'].sum()
    np.testing.assert_allclose(outs[1][0] / M, g['mineig_norm_loss'], rtol=1e-9)
    np.testing.assert_allclose(outs[1][2:4] / M, g['mineig_norm_grad_w'].reshape(-1), rtol=1e-7)


def test_trainer_with_two_sequences_on_one_gpu(golden):
    """Config 3 shape on one device: two sequences per rank, accumulated before the (here trivial) rank reduction."""
    from depth_correction_amd.plan import SequencePlan, SequenceTrainer
    g = golden('room_k10')
    cfg = _cfg(g)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    mask2 = mask & (torch.arange(len(mask), device=mask.device) % 2 == 0)
    plans = [SequencePlan(clouds, poses, ns[0], mask), SequencePlan(clouds, poses, ns[0].clone(), mask2, active_only=True)]
    tr = SequenceTrainer(plans, g['w'], g['exponent'], [poses, poses], lr=1e-2)
    acc = npy(tr.step())
    singles = []
    for p in plans:
        out = torch.zeros(2 + 4 + 12 * p.n_scans, dtype=torch.float64, device=poses.device)
        p.eval_native(torch.tensor(g['w'].reshape(-1), device=poses.device), torch.tensor(g['exponent'].reshape(-1), device=poses.device),
                      p.poses12(poses), out)
        singles.append(npy(out))
    np.testing.assert_allclose(acc[:4], singles[0][:4] + singles[1][:4], rtol=1e-13)
    assert tr.count == float(mask.sum() + mask2.sum()) and acc[1] == tr.count
    assert not np.allclose(npy(tr.w), g['w'].reshape(-1))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_device_scan_loaders_match_host_pipeline(tmp_path, dtype):
    """Direct-to-device loaders: KITTI-360 .bin with the ego-box crop, ASL CSV, plain and structured .npz --
    one kernel pass (dc_cloud_from_points: crop + depth pre-filter + from_points) against the reference's host pipeline
    on the same synthetic files: numpy reader -> filter_depth on the array (filters.py:116-141) -> from_points
    (depth_cloud.py:592-638).  Kept rows and their order identical; depth / dirs to the last ulp of the sqrt / division."""
    from numpy.lib.recfunctions import unstructured_to_structured
    from depth_correction_amd.scan_io import (DeviceScanDataset, load_kitti_bin_device, load_points_csv_device,
                                              load_points_npz_device, read_kitti_bin, read_points_csv, write_poses_csv)
    rng = np.random.default_rng(3)
    n = 5000
    raw = (rng.normal(size=(n, 4)) * [8, 8, 1.5, 1]).astype(np.float32)
    raw[:50, :2] = rng.uniform(-0.95, 0.95, size=(50, 2))                 # inside the ego box
    raw[50] = 0.0                                                          # a zero-depth ray
    d = tmp_path / 'velodyne'
    d.mkdir()
    raw.tofile(str(d / ('%010d.bin' % 3)))
    tol = dict(rtol=3e-7, atol=0) if dtype == np.float32 else dict(rtol=1e-15, atol=0)

    def host_cloud(xyz, vps, lo, hi):
        xyz, vps = np.asarray(xyz), (np.zeros_like(xyz) if vps is None else np.asarray(vps))
        depth_raw = np.sqrt(((xyz - vps) ** 2).sum(-1, dtype=xyz.dtype))
        keep = np.ones(len(xyz), bool)
        if lo is not None:
            keep &= depth_raw >= xyz.dtype.type(lo)
        if hi is not None:
            keep &= depth_raw <= xyz.dtype.type(hi)
        rays = xyz[keep].astype(dtype) - vps[keep].astype(dtype)
        depth = np.sqrt((rays ** 2).sum(-1))[:, None]
        with np.errstate(invalid='ignore', divide='ignore'):
            dirs = np.where(depth > 0, rays / depth, rays)
        return keep, vps[keep].astype(dtype), dirs, depth

    def check(dc, keep, vps, dirs, depth):
        assert len(dc) == int(keep.sum()) and dc.dirs.is_cuda and dc.dirs.dtype == getattr(torch, np.dtype(dtype).name)
        np.testing.assert_allclose(npy(dc.depth), depth, **tol)
        np.testing.assert_allclose(npy(dc.dirs), dirs, atol=3e-7 if dtype == np.float32 else 1e-15, rtol=0)
        np.testing.assert_array_equal(npy(dc.vps), vps)

    # KITTI .bin: ego crop (reader) then depth bounds
    host = read_kitti_bin(str(d / ('%010d.bin' % 3)))
    xyz = np.stack([host[f] for f in 'xyz'], 1)
    for lo, hi in ((None, None), (2.0, 15.0)):
        keep, vps, dirs, depth = host_cloud(xyz, None, lo, hi)
        dc = load_kitti_bin_device(str(d / ('%010d.bin' % 3)), dtype=dtype, min_depth=lo, max_depth=hi)
        check(dc, keep, vps, dirs, depth)
    assert len(load_kitti_bin_device(str(d / ('%010d.bin' % 3)), filter_ego_pts_depth=None)) == n
    # ASL CSV
    csv = tmp_path / 'scan.csv'
    np.savetxt(str(csv), np.concatenate([np.arange(300)[:, None], raw[100:400, :3].astype(np.float64), np.ones((300, 2))], 1),
               delimiter=',', header='id,x,y,z,a,b')
    keep, vps, dirs, depth = host_cloud(read_points_csv(str(csv)), None, 1.0, None)
    check(load_points_csv_device(str(csv), dtype=dtype, min_depth=1.0), keep, vps, dirs, depth)
    # FEE-corridor style structured .npz with viewpoints
    vp = (rng.normal(size=(n, 3)) * 0.05).astype(np.float32)
    np.savez(str(tmp_path / 'c.npz'), cloud=unstructured_to_structured(np.concatenate([raw[:, :3], vp], 1),
                                                                      names=['x', 'y', 'z', 'vp_x', 'vp_y', 'vp_z']))
    keep, vps, dirs, depth = host_cloud(raw[:, :3], vp, None, 12.0)
    check(load_points_npz_device(str(tmp_path / 'c.npz'), dtype=dtype, max_depth=12.0), keep, vps, dirs, depth)
    # the dataset wrapper feeds local_feature_cloud / train() with device clouds
    write_poses_csv([3], [np.eye(4)], str(tmp_path / 'poses.csv'))
    ds = DeviceScanDataset(str(d), str(tmp_path / 'poses.csv'), dtype=dtype, min_depth=2.0, max_depth=15.0)
    cloud, pose = ds[0]
    keep, _, _, depth = host_cloud(xyz, None, 2.0, 15.0)
    assert len(cloud) == int(keep.sum()) and np.allclose(pose, np.eye(4))


def test_from_points_device_kernel_matches_torch_expressions(golden):
    """DepthCloud.from_points on device tensors (one kernel) equals the reference's tensor expressions."""
    from depth_correction_amd.depth_cloud import DepthCloud
    g = golden('room_k10')
    for dt in (torch.float64, torch.float32):
        pts = t(g['scan0_xyz'], 'cuda:0').to(dt)
        vps = torch.randn_like(pts) * 0.01
        dc = DepthCloud.from_points(pts, vps=vps)
        rays = pts - vps
        depth = rays.norm(dim=-1, keepdim=True)
        tol = 1e-15 if dt == torch.float64 else 3e-7
        np.testing.assert_allclose(npy(dc.depth), npy(depth), rtol=tol)
        np.testing.assert_allclose(npy(dc.dirs), npy(rays / depth), atol=tol, rtol=0)
        assert torch.equal(dc.vps, vps)
        dc32 = DepthCloud.from_points(npy(pts), dtype=np.float32, device='cuda:0')
        assert dc32.dirs.dtype == torch.float32 and dc32.dirs.is_cuda


def test_fused_pose_correction_equals_tensor_chain():
    """eval.create_corrected_poses' fused kernel pair (dc_pose_correct_fwd / _bwd) against the tensor expressions
    poses @ xyz_axis_angle_to_matrix(deltas) and their autograd: per-pose and shared corrections, zero (the start of every
    optimisation), below the small-angle switch, generic and large rotations."""
    from depth_correction_amd.transform import corrected_poses, xyz_axis_angle_to_matrix
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(0)
    n = 7
    poses = torch.eye(4, dtype=torch.float64).repeat(n, 1, 1)
    poses[:, :3, :] = torch.randn((n, 3, 4), generator=g, dtype=torch.float64)
    poses = poses.to(dev)
    cases = {'zero': torch.zeros((n, 6), dtype=torch.float64),
             'tiny': torch.randn((n, 6), generator=g, dtype=torch.float64) * 1e-8,
             'generic': torch.randn((n, 6), generator=g, dtype=torch.float64) * 0.3,
             'large': torch.randn((n, 6), generator=g, dtype=torch.float64) * 2.0,
             'shared': torch.randn((1, 6), generator=g, dtype=torch.float64) * 0.1,
             'shared_zero': torch.zeros((1, 6), dtype=torch.float64)}
    up = torch.randn((n, 4, 4), generator=g, dtype=torch.float64).to(dev)
    for name, d0 in cases.items():
        da = d0.clone().to(dev).requires_grad_(True)
        db = d0.clone().to(dev).requires_grad_(True)
        Ta = corrected_poses(poses, da)
        Tb = torch.matmul(poses, xyz_axis_angle_to_matrix(db))
        torch.testing.assert_close(Ta, Tb, rtol=1e-13, atol=1e-14, msg=name)
        (Ta * up).sum().backward()
        (Tb * up).sum().backward()
        torch.testing.assert_close(da.grad, db.grad, rtol=1e-11, atol=1e-13, msg=name)
        assert torch.isfinite(da.grad).all()
    # float32 poses / corrections keep their dtypes
    T32 = corrected_poses(poses.float(), cases['generic'].float().to(dev))
    assert T32.dtype == torch.float32
    torch.testing.assert_close(T32.double(), torch.matmul(poses, xyz_axis_angle_to_matrix(cases['generic'].to(dev))), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('optimizer', ['torch', 'dc'])
def test_training_iteration_replays_as_one_graph(golden, optimizer):
    """The drop-in iteration (fused loss -> backward -> Adam) captured by torch.cuda.graph: the library's launches go to
    torch's current stream, so the capture holds them; replaying it takes the same steps as the eager loop -- with
    torch.optim.Adam(capturable=True) and with optim.Adam, whose step counter lives on the device for exactly this reason
    (a host-side counter would be frozen into the graph and every replay would apply the same bias correction)."""
    from depth_correction_amd.plan import consistency_loss
    g = golden('room_k10')
    cfg = _cfg(g)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    from depth_correction_amd.plan import SequencePlan
    plan = SequencePlan(clouds, poses, ns[0], mask)
    dev = poses.device
    e = torch.tensor(g['exponent'].reshape(1, -1), device=dev)

    def make():
        from depth_correction_amd.optim import Adam as DcAdam
        w = torch.nn.Parameter(torch.tensor(g['w'].reshape(1, -1), device=dev))
        return w, (torch.optim.Adam([w], lr=1e-3, capturable=True) if optimizer == 'torch' else DcAdam([w], lr=1e-3))

    def iteration(w, opt):
        opt.zero_grad(set_to_none=False)
        s, cnt = consistency_loss(plan, w, e, poses)
        loss = s / cnt
        loss.backward()
        opt.step()
        return loss

    w_e, opt_e = make()
    for _ in range(3 + 5):
        iteration(w_e, opt_e)
    w_g, opt_g = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            iteration(w_g, opt_g)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = iteration(w_g, opt_g)
    for _ in range(5):                       # eager run: 8 steps; here 3 eager + 5 replays (capturing executes nothing)
        graph.replay()
    torch.cuda.synchronize()
    torch.testing.assert_close(w_g.detach(), w_e.detach(), rtol=1e-12, atol=0)
    assert torch.isfinite(static_loss).all()


@pytest.mark.parametrize('name', ['Linear', 'InvCos', 'ScaledInvCos'])
@pytest.mark.parametrize('fused', [True, False])
def test_other_models_iteration_golden(golden, name, fused):
    """The reference's other depth-correction models (model.py:113-146, 289-349) through eval_loss_clouds: inside the
    fused kernels (model kinds 3-5) and through the un-fused operators, against the live-reference fixture."""
    from depth_correction_amd import model as M
    from depth_correction_amd.eval import eval_loss_clouds, fused_supported
    from depth_correction_amd.loss import create_loss
    g, m = golden('room_k10'), golden('models')
    cfg = _cfg(g, loss='min_eigval_loss')
    cfg.loss_kwargs.update(normalization=True, sqrt=False)
    cfg.fused = fused
    clouds, poses, _, ns, mask = _setup(g, cfg)
    w0 = m[name + '_w'].ravel()
    kw = dict(w0=float(w0[0]), w1=float(w0[1]), b=float(w0[2])) if name == 'Linear' else dict(p0=float(w0[0]))
    model = getattr(M, name)(device=cfg.device, **kw)
    assert fused_supported([clouds], model, cfg) == fused
    loss, loss_clouds, _, _ = eval_loss_clouds([clouds], [poses], [None], [mask], [ns], model, create_loss(cfg), cfg)
    loss.backward()
    grads = np.array([float(p.grad) for p in model.parameters()])
    np.testing.assert_allclose(loss.item(), m[name + '_loss'], rtol=1e-9)
    np.testing.assert_allclose(grads, m[name + '_grad_w'].ravel(), rtol=2e-6, atol=1e-10)
    np.testing.assert_allclose(npy(model(clouds[0]).depth), m[name + '_depth0'], rtol=1e-12)


@pytest.mark.parametrize('tag,loss,kw', [('norm_r07', 'min_eigval_loss', dict(normalization=True, sqrt=False)),
                                         ('raw_sqrt_r09_m08', 'min_eigval_loss', dict(normalization=False, sqrt=True)),
                                         ('trace_r05', 'trace_loss', dict(sqrt=False))])
@pytest.mark.parametrize('fused', [True, False])
def test_quantile_inlier_gating_golden(golden, tag, loss, kw, fused):
    """eval_loss_clouds with inlier_ratio < 1 (loss.py:256-277) against the live reference's loss, dL/dw and number of
    inliers (tests/golden/inliers.npz, same inputs as room_k10.npz): through the fused plan -- forward with raw pointwise
    losses, torch.quantile bound, dc_consistency_gate, backward -- and through the un-fused operators."""
    from depth_correction_amd.eval import eval_loss_clouds, fused_supported
    from depth_correction_amd.loss import create_loss
    from depth_correction_amd.model import ScaledPolynomial
    g, gi = golden('room_k10'), golden('inliers')
    cfg = _cfg(g, fused=fused)
    clouds, poses, _, ns, mask = _setup(g, cfg)
    cfg.loss = loss
    cfg.loss_kwargs.update(kw, inlier_ratio=float(gi['inl_%s_ratio' % tag]), inlier_loss_mult=float(gi['inl_%s_mult' % tag]))
    model = ScaledPolynomial(w=gi['inl_w'].tolist(), exponent=gi['inl_exponent'].tolist(), device=cfg.device)
    assert fused_supported([clouds], model, cfg) == fused
    out, views, _, _ = eval_loss_clouds([clouds], [poses], [None], [mask], [ns], model, create_loss(cfg), cfg)
    out.backward()
    n_ref = int(gi['inl_%s_n_inliers' % tag])
    if fused:
        assert int(views[0].count) == n_ref == int(views[0].mask.sum())
    else:
        assert len(views[0]) == n_ref
    np.testing.assert_allclose(out.item(), float(gi['inl_%s_loss' % tag]), rtol=1e-9)
    np.testing.assert_allclose(npy(model.w.grad).ravel(), gi['inl_%s_grad_w' % tag].ravel(), rtol=1e-7)


def test_dc_adam_equals_torch_adam():
    """optim.Adam (one dc_adam_step launch per fp64 GPU parameter) takes torch.optim.Adam's steps: 60 iterations on two
    parameters of the shapes train() optimises, with and without weight decay, to round-off; fp32 parameters take the tensor
    expressions of the same update (torch's multi-tensor path orders a few operations differently: float32 round-off); the
    version counter of a parameter written through its pointer moves."""
    from depth_correction_amd.optim import Adam
    torch.manual_seed(0)
    for dtype, wd, tol in ((torch.float64, 0.0, 1e-13), (torch.float64, 0.05, 1e-13), (torch.float32, 0.0, 2e-6)):
        ps = [[torch.nn.Parameter(torch.randn(s, dtype=dtype, device='cuda:0') * 0.1) for s in ((1, 2), (10, 6))]]
        ps.append([torch.nn.Parameter(p.detach().clone()) for p in ps[0]])
        opts = [Adam([{'params': ps[0][:1], 'lr': 2e-3}, {'params': ps[0][1:], 'lr': 1e-3}], weight_decay=wd),
                torch.optim.Adam([{'params': ps[1][:1], 'lr': 2e-3}, {'params': ps[1][1:], 'lr': 1e-3}], weight_decay=wd)]
        target = torch.linspace(-1, 1, 60, dtype=dtype, device='cuda:0').reshape(10, 6)
        v0 = ps[0][0]._version
        for it in range(60):
            for params, opt in zip(ps, opts):
                opt.zero_grad()
                loss = (params[0] ** 2).sum() * (1 + 0.1 * it) + ((params[1] - target) ** 2 * (params[0].sum() + 2)).sum()
                loss.backward()
                opt.step()
        assert ps[0][0]._version > v0
        for a, b in zip(ps[0], ps[1]):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=tol, atol=tol * 1e-2)


def test_local_clouds_on_several_streams_are_identical():
    """pipeline.on_streams: the scans' local feature clouds issued round robin on four streams give bit for bit what one
    stream gives (independent jobs; the join orders every later use after their producers)."""
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd.pipeline import build_sequence
    ds = RoomBoxDataset(n_pts=20000, n_poses=6, dtype=np.float32)
    scans = [np.stack([c[f] for f in 'xyz'], 1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    ref_plan, ref = build_sequence(scans, poses, local_streams=1)
    for _ in range(3):
        plan, info = build_sequence(scans, poses, local_streams=4)
        assert plan.count == ref_plan.count
        for a, b in zip(info['clouds'], ref['clouds']):
            for f in ('dirs', 'depth', 'inc_angles', 'mask', 'normals', 'eigvals', 'neighbors'):
                assert torch.equal(a[f], b[f]), f
        assert torch.equal(info['neighbors'], ref['neighbors']) and torch.equal(info['mask'], ref['mask'])


def _train_and_collect(cfg, train_datasets, val_datasets, capsys):
    import re
    from depth_correction_amd.train import train
    capsys.readouterr()
    best = train(cfg, train_datasets=train_datasets, val_datasets=val_datasets)
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.startswith('It. ')]
    vals = [tuple(float(x) for x in re.findall(r'(?:train loss|val\.): (-?[0-9]+\.[0-9]+|nan)', ln)) for ln in lines]
    flags = [ln.rstrip('.').endswith(' saved') and not ln.rstrip('.').endswith('not saved') for ln in lines]
    return best, lines, vals, flags, out


@pytest.mark.parametrize('mode', ['min_eigval_model', 'min_eigval_model_graph', 'icp_pose'])
def test_train_batched_graph_loop_equals_plain_loop(golden, tmp_path, capsys, mode, monkeypatch):
    """train() with its default no-op callbacks runs without a host round trip per iteration (train._batched_loop: a device
    ring of per-iteration records drained every cfg.loop_batch iterations, the iteration replayed as one hipGraph after three
    eager ones).  Against cfg.loop_batch = 1 (the reference's per-iteration bookkeeping, train.py:220-322): the same
    progress lines -- losses, model string, saved / not saved -- and the same final checkpoint, for the model-only
    min-eigenvalue loop (validation sequence included) and for ICP with per-pose corrections."""
    from depth_correction_amd.config import Config, Loss, PoseCorrection
    from depth_correction_amd import train as train_mod
    took = []
    for name in ('_native_loop', '_batched_loop'):
        fn = getattr(train_mod, name)
        monkeypatch.setattr(train_mod, name, (lambda f, n: (lambda *a, **k: (took.append(n), f(*a, **k))[1]))(fn, name))
    if mode.startswith('min_eigval_model'):
        # model-only: the library's chained step drives the loop (one launch per iteration, train._native_loop) unless
        # cfg.loop_native is off, in which case the captured autograd iteration does (train._batched_loop)
        g = golden('room_k10')
        mk = lambda d, **kw: _cfg(g, n_opt_iters=13, lr=5e-3, log_dir=str(d), loop_native=mode == 'min_eigval_model',
                                  model_kwargs={'w': g['w'].tolist(), 'exponent': g['exponent'].tolist()}, **kw)
        ds = list(zip(_scan_arrays(g), g['poses']))
        tr_ds, va_ds = [ds], [ds[:2]]
    else:
        from depth_correction_amd.dataset import KittiLikeDataset
        from depth_correction_amd.preproc import filtered_cloud
        mk = lambda d, **kw: Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2,
                                    min_depth=5.0, max_depth=12.0, vp_dispersion_bounds=[], n_opt_iters=13, lr=2e-3,
                                    log_dir=str(d), loop_native=False,        # (the native loop: test_train_native_icp_pose_loop_...)
                                    model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]}).from_dict(kw)
        c0 = mk(tmp_path)
        seq = [(filtered_cloud(cloud, c0), pose) for cloud, pose in KittiLikeDataset(n_poses=3, n_rings=96, n_azimuth=384)]
        tr_ds, va_ds = [seq], []
    (tmp_path / 'plain').mkdir()
    (tmp_path / 'fast').mkdir()
    b0, l0, v0, f0, _ = _train_and_collect(mk(tmp_path / 'plain', loop_batch=1), tr_ds, va_ds, capsys)
    b1, l1, v1, f1, out1 = _train_and_collect(mk(tmp_path / 'fast', loop_batch=5), tr_ds, va_ds, capsys)
    assert 'could not be captured' not in out1, out1                 # the iteration replays as a graph
    assert took == (['_native_loop'] if mode == 'min_eigval_model' else ['_batched_loop']), took
    assert len(l0) == len(l1) == 13 and f0 == f1 and any(f0)
    np.testing.assert_allclose(np.array(v1), np.array(v0), rtol=1e-7, atol=1e-12)
    for a, b in zip(l0, l1):                                           # model string: weights of the iteration, 6 digits
        ma, mb = a.split('Model ')[1], b.split('Model ')[1]
        assert ma == mb, (a, b)
    assert v0[-1][0] != v0[0][0]                                       # the parameters move
    sa, sb = torch.load(b0.model_state_dict), torch.load(b1.model_state_dict)
    assert list(sa) == list(sb)
    for k in sa:
        np.testing.assert_allclose(sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-9, atol=1e-15)
    da, db = torch.load(b0.train_pose_deltas), torch.load(b1.train_pose_deltas)
    assert len(da) == len(db)
    for x, y in zip(da, db):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=1e-8, atol=1e-14)
    import os
    assert os.path.basename(b0.model_state_dict) == os.path.basename(b1.model_state_dict)        # same iteration, same loss
    # the plain loop writes a file set per improvement, the batched one per drained batch
    n_plain = len([f for f in os.listdir(tmp_path / 'plain') if f.endswith('_state_dict.pth')])
    n_fast = len([f for f in os.listdir(tmp_path / 'fast') if f.endswith('_state_dict.pth')])
    assert n_plain == sum(f0) and 1 <= n_fast <= 3


@pytest.mark.parametrize('correction, float_type, with_val', [('pose', 'float64', True), ('pose', 'float32', False), ('sequence', 'float64', True),
                                                              ('pose', 'float64', 'multi'), ('sequence', 'float64', 'multi')])
def test_train_native_pose_loop_equals_plain_loop(golden, tmp_path, capsys, monkeypatch, correction, float_type, with_val):
    """train() with per-pose / per-sequence corrections on the map-consistency loss (scripts/model_poses_learning:71) runs on
    train._native_pose_loop: per iteration the evaluation and ONE finishing launch (dc_pose_train_finish: pose-chain adjoint, first
    pose fixed, both Adam updates, next poses, the record).  Against cfg.loop_batch = 1 (the reference's loop, train.py:220-322,
    through autograd and the tensor-level optimisers): the same progress lines and the same checkpoint -- weights, corrections,
    corrected poses.  float64 clouds: both optimise in fp64 (1e-9); float32: the corrections of the plain loop are float32
    tensors, the native loop keeps them in fp64 (1e-5)."""
    from depth_correction_amd import train as train_mod
    took = []
    for name in ('_native_loop', '_native_pose_loop', '_batched_loop'):
        fn = getattr(train_mod, name)
        monkeypatch.setattr(train_mod, name, (lambda f, n: (lambda *a, **k: (took.append(n), f(*a, **k))[1]))(fn, name))
    g = golden('room_k10')
    mk = lambda d, **kw: _cfg(g, n_opt_iters=11, lr=2e-3, log_dir=str(d), pose_correction=correction, float_type=float_type,
                              model_kwargs={'w': g['w'].tolist(), 'exponent': g['exponent'].tolist()}, **kw)
    ds = list(zip(_scan_arrays(g), g['poses']))
    tr_ds, va_ds = [ds], ([ds[:2]] if with_val else [])
    if with_val == 'multi':                  # several sequences in both losses: sums joined on the device (dc_pose_train_combine)
        tr_ds, va_ds = [ds, ds[1:]], [ds[:2], ds[2:]]
    (tmp_path / 'plain').mkdir()
    (tmp_path / 'fast').mkdir()
    b0, l0, v0, f0, _ = _train_and_collect(mk(tmp_path / 'plain', loop_batch=1), tr_ds, va_ds, capsys)
    assert took == []
    b1, l1, v1, f1, _ = _train_and_collect(mk(tmp_path / 'fast', loop_batch=4), tr_ds, va_ds, capsys)
    assert took == ['_native_pose_loop'], took
    tol = 1e-9 if float_type == 'float64' else 1e-5
    assert len(l0) == len(l1) == 11 and f0 == f1 and any(f0)
    np.testing.assert_allclose(np.array(v1), np.array(v0), rtol=tol, atol=1e-12)
    assert v0[-1][0] != v0[0][0]
    sa, sb = torch.load(b0.model_state_dict), torch.load(b1.model_state_dict)
    for k in sa:
        np.testing.assert_allclose(sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=tol, atol=1e-12)
    da, db = torch.load(b0.train_pose_deltas), torch.load(b1.train_pose_deltas)
    assert len(da) == len(db) == len(tr_ds)
    # (float32: the plain loop's Adam rounds moments and corrections to float32 at every step)
    dtol = 1e-7 if float_type == 'float64' else 1e-4
    pa = torch.load(b0.model_state_dict.replace('_state_dict.pth', '_poses_upd.pth'))
    pb = torch.load(b1.model_state_dict.replace('_state_dict.pth', '_poses_upd.pth'))
    assert len(pa) == len(pb) == len(tr_ds)
    for q in range(len(tr_ds)):
        assert da[q].shape == db[q].shape and da[q].dtype == db[q].dtype
        scale = np.abs(da[q].cpu().numpy()).max()
        assert scale > 0
        np.testing.assert_allclose(db[q].cpu().numpy(), da[q].cpu().numpy(), rtol=dtol, atol=dtol * scale)
        if correction == 'pose':
            assert not db[q][0].any()                                  # the first pose stays where it is (train.py:309-311)
        assert pa[q].shape == pb[q].shape
        np.testing.assert_allclose(pb[q].cpu().numpy(), pa[q].cpu().numpy(), rtol=0, atol=dtol * 10)


@pytest.mark.parametrize('n_it, batch, graph', [(1, 4, True), (2, 2, True), (7, 2, True), (7, 3, False)])
def test_train_native_pose_loop_short_runs_and_small_batches(golden, tmp_path, capsys, monkeypatch, n_it, batch, graph):
    """train._native_pose_loop at its edges: runs shorter than the three eager iterations before the capture, rings smaller than
    that, an uncaptured loop (cfg.loop_graph = False) -- the same progress lines as the reference's loop."""
    from depth_correction_amd import train as train_mod
    took = []
    fn = train_mod._native_pose_loop
    monkeypatch.setattr(train_mod, '_native_pose_loop', lambda *a, **k: (took.append(1), fn(*a, **k))[1])
    g = golden('room_k10')
    mk = lambda d, **kw: _cfg(g, n_opt_iters=n_it, lr=2e-3, log_dir=str(d), pose_correction='pose', float_type='float64',
                              loop_graph=graph, model_kwargs={'w': g['w'].tolist(), 'exponent': g['exponent'].tolist()}, **kw)
    ds = list(zip(_scan_arrays(g), g['poses']))
    (tmp_path / 'plain').mkdir()
    (tmp_path / 'fast').mkdir()
    b0, l0, v0, f0, _ = _train_and_collect(mk(tmp_path / 'plain', loop_batch=1), [ds], [], capsys)
    b1, l1, v1, f1, _ = _train_and_collect(mk(tmp_path / 'fast', loop_batch=batch), [ds], [], capsys)
    assert took == [1] and len(l0) == len(l1) == n_it and f0 == f1
    np.testing.assert_allclose(np.array(v1), np.array(v0), rtol=1e-9, atol=1e-12)
    da, db = torch.load(b0.train_pose_deltas), torch.load(b1.train_pose_deltas)
    np.testing.assert_allclose(db[0].cpu().numpy(), da[0].cpu().numpy(), rtol=1e-7, atol=1e-10)


def test_train_model_only_over_two_sequences_takes_the_linked_chain(golden, tmp_path, capsys, monkeypatch):
    """train() with TWO training sequences in the loss and only the model optimised (train.py:172-175): train._native_loop on the
    chain over the sequences (one launch per sequence and iteration, dc_sequence_step_linked) against the reference's
    per-iteration loop -- the same progress lines and final checkpoint; a validation sequence included."""
    from depth_correction_amd import train as train_mod
    g = golden('room_k10')
    took = []
    fn = train_mod._native_loop
    monkeypatch.setattr(train_mod, '_native_loop', lambda *a, **k: (took.append(1), fn(*a, **k))[1])
    mk = lambda d, **kw: _cfg(g, n_opt_iters=13, lr=5e-3, log_dir=str(d), model_kwargs={'w': g['w'].tolist(), 'exponent': g['exponent'].tolist()}, **kw)
    ds = list(zip(_scan_arrays(g), g['poses']))
    tr_ds, va_ds = [ds, ds[1:]], [ds[:2]]
    (tmp_path / 'plain').mkdir()
    (tmp_path / 'fast').mkdir()
    b0, l0, v0, f0, _ = _train_and_collect(mk(tmp_path / 'plain', loop_batch=1), tr_ds, va_ds, capsys)
    assert took == []
    b1, l1, v1, f1, _ = _train_and_collect(mk(tmp_path / 'fast', loop_batch=5), tr_ds, va_ds, capsys)
    assert took == [1] and len(l0) == len(l1) == 13 and f0 == f1 and any(f0)
    np.testing.assert_allclose(np.array(v1), np.array(v0), rtol=1e-8, atol=1e-12)
    sa, sb = torch.load(b0.model_state_dict), torch.load(b1.model_state_dict)
    for k in sa:
        np.testing.assert_allclose(sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-9, atol=1e-15)


@pytest.mark.parametrize('plane, multi', [(True, False), (False, False), (True, True)])
def test_train_native_icp_pose_loop_equals_plain_loop(tmp_path, capsys, monkeypatch, plane, multi):
    """The C4 shape -- ICP loss over consecutive scan pairs (point to plane / point to point), model weights and per-pose corrections
    optimised (scripts/model_poses_learning_icp) -- on train._native_pose_loop: dc_p2plane_sequence / dc_p2point_sequence +
    dc_pose_train_finish (layout 1) per iteration, against the reference's loop with cfg.loop_batch = 1."""
    from depth_correction_amd import train as train_mod
    from depth_correction_amd.config import Config, Loss, PoseCorrection
    from depth_correction_amd.dataset import KittiLikeDataset
    from depth_correction_amd.preproc import filtered_cloud
    took = []
    for name in ('_native_loop', '_native_pose_loop', '_batched_loop'):
        fn = getattr(train_mod, name)
        monkeypatch.setattr(train_mod, name, (lambda f, n: (lambda *a, **k: (took.append(n), f(*a, **k))[1]))(fn, name))
    def mk(d, **kw):
        c = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0, max_depth=12.0,
                   vp_dispersion_bounds=[], n_opt_iters=11, lr=2e-3, log_dir=str(d), float_type='float64',
                   model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]}).from_dict(kw)
        c.loss_kwargs['icp_point_to_plane'] = plane
        return c
    c0 = mk(tmp_path)
    seq = [(filtered_cloud(cloud, c0), pose) for cloud, pose in KittiLikeDataset(n_poses=3, n_rings=96, n_azimuth=384)]
    seqs = [seq, seq[1:]] if multi else [seq]                        # (several sequences: icp_loss averages their losses)
    (tmp_path / 'plain').mkdir()
    (tmp_path / 'fast').mkdir()
    b0, l0, v0, f0, _ = _train_and_collect(mk(tmp_path / 'plain', loop_batch=1), seqs, [], capsys)
    assert took == []
    b1, l1, v1, f1, _ = _train_and_collect(mk(tmp_path / 'fast', loop_batch=4), seqs, [], capsys)
    assert took == ['_native_pose_loop'], took
    assert len(l0) == len(l1) == 11 and f0 == f1 and any(f0)
    np.testing.assert_allclose(np.array(v1), np.array(v0), rtol=1e-8, atol=1e-12)
    assert v0[-1][0] != v0[0][0]
    sa, sb = torch.load(b0.model_state_dict), torch.load(b1.model_state_dict)
    for k in sa:
        np.testing.assert_allclose(sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-8, atol=1e-13)
    da, db = torch.load(b0.train_pose_deltas), torch.load(b1.train_pose_deltas)
    assert len(da) == len(db) == len(seqs)
    for q in range(len(seqs)):
        assert not db[q][0].any()                                    # (the checkpoint is the best iteration's: possibly the first)
        np.testing.assert_allclose(db[q].cpu().numpy(), da[q].cpu().numpy(), rtol=1e-7, atol=1e-10)


def test_train_native_loop_recovers_from_a_chain_timeout(golden, tmp_path, capsys, monkeypatch):
    """train._native_loop reads the plan's status word at every drain: a chained launch whose wait for its weights
    ran out (forced here: dc_set_option(5, 0), every wait gives up at once) poisons the sums and, through Adam, the weights.
    The loop must notice, go back to the optimiser state of the last drain, and finish the run with ordinary steps: the same
    13 progress lines and the same final checkpoint as the loop with per-iteration bookkeeping."""
    import warnings
    from depth_correction_amd import _native as nv
    from depth_correction_amd import train as train_mod
    g = golden('room_k10')
    mk = lambda d, **kw: _cfg(g, n_opt_iters=13, lr=5e-3, log_dir=str(d), model_kwargs={'w': g['w'].tolist(), 'exponent': g['exponent'].tolist()}, **kw)
    ds = list(zip(_scan_arrays(g), g['poses']))
    took = []
    fn = train_mod._native_loop
    monkeypatch.setattr(train_mod, '_native_loop', lambda *a, **k: (took.append(1), fn(*a, **k))[1])
    (tmp_path / 'plain').mkdir()
    (tmp_path / 'forced').mkdir()
    b0, l0, v0, f0, _ = _train_and_collect(mk(tmp_path / 'plain', loop_batch=1, loop_native=False), [ds], [ds[:2]], capsys)
    nv.check(nv.lib().dc_set_option(5, 0), 'dc_set_option')
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            b1, l1, v1, f1, _ = _train_and_collect(mk(tmp_path / 'forced', loop_batch=5), [ds], [ds[:2]], capsys)
    finally:
        nv.check(nv.lib().dc_set_option(5, -1), 'dc_set_option')
    assert took == [1]
    assert any('gave up waiting' in str(w.message) for w in caught), [str(w.message) for w in caught]
    assert len(l1) == len(l0) == 13 and f0 == f1
    assert not np.isnan(np.array(v1)).any()
    np.testing.assert_allclose(np.array(v1), np.array(v0), rtol=1e-7, atol=1e-12)
    sa, sb = torch.load(b0.model_state_dict), torch.load(b1.model_state_dict)
    for k in sa:
        np.testing.assert_allclose(sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-9, atol=1e-15)


def test_train_releases_only_the_plans_it_built(golden, tmp_path, capsys):
    """train() drops the per-sequence plans it added to the process-wide caches and leaves alone what a caller had cached
    before; cfg.keep_plans (a declared Config field) keeps its own too."""
    from depth_correction_amd import eval as eval_mod
    from depth_correction_amd.config import Config
    from depth_correction_amd.plan import PlanRegistry
    assert 'keep_plans' in Config().__dict__
    g = golden('room_k10')
    ds = list(zip(_scan_arrays(g), g['poses']))
    sentinel = torch.zeros(1)
    eval_mod._plans.clear()                                           # (entries earlier tests of this process left behind)
    eval_mod._plans.get([sentinel], ('outside',), lambda: 'a plan of somebody else')
    mk = lambda d, **kw: _cfg(g, n_opt_iters=3, lr=5e-3, log_dir=str(d), model_kwargs={'w': g['w'].tolist(), 'exponent': g['exponent'].tolist()}, **kw)
    try:
        _train_and_collect(mk(tmp_path / 'a'), [ds], [], capsys)
        keys = eval_mod._plans.keys()
        assert keys == [PlanRegistry.key_of([sentinel], ('outside',))], keys
        _train_and_collect(mk(tmp_path / 'b', keep_plans=True), [ds], [], capsys)
        assert len(eval_mod._plans.keys()) > 1
    finally:
        eval_mod._plans.clear()


def test_device_scan_loaders_match_the_reference_readers(golden, tmp_path):
    """scan_io.load_*_device on the files of tests/golden/io.npz against what the LIVE reference made of them: its readers
    (datasets/kitti360.py:96-109 with the ego-box crop, asl_laser.py:33-45, fee_corridor.py:35-38) followed by
    DepthCloud.from_structured_array / from_points (depth_cloud.py:577-638) and filter_depth (filters.py:116-141)."""
    from depth_correction_amd import scan_io
    g = golden('io')

    def wr(name, key):
        p = tmp_path / name
        p.write_bytes(g[key].tobytes())
        return str(p)
    path = wr('0000000003.bin', 'kitti_bin')
    for tag, dtype, rtol in (('f32', torch.float32, 2e-7), ('f64', torch.float64, 1e-15)):
        dc = scan_io.load_kitti_bin_device(path, dtype=dtype)
        assert dc.dirs.dtype == dtype and len(dc) == len(g['kitti_%s_depth' % tag])                # same rows survive the crop
        np.testing.assert_allclose(npy(dc.depth), g['kitti_%s_depth' % tag], rtol=rtol)
        np.testing.assert_allclose(npy(dc.dirs), g['kitti_%s_dirs' % tag], rtol=0, atol=rtol)
        assert bool((dc.vps == 0).all())
        keep = g['kitti_%s_keep' % tag]
        dcf = scan_io.load_kitti_bin_device(path, dtype=dtype, min_depth=2.0, max_depth=25.0)      # crop + depth filter fused
        assert len(dcf) == int(keep.sum()) and 0 < len(dcf) < len(dc)
        np.testing.assert_allclose(npy(dcf.depth), g['kitti_%s_depth' % tag][keep], rtol=rtol)
    dc = scan_io.load_points_csv_device(wr('PointCloud7.csv', 'asl_csv'), dtype=torch.float64)
    np.testing.assert_allclose(npy(dc.depth), g['asl_f64_depth'], rtol=1e-15)
    np.testing.assert_allclose(npy(dc.dirs), g['asl_f64_dirs'], rtol=0, atol=1e-15)
    dc = scan_io.load_points_npz_device(wr('cloud7.npz', 'asl_npz'), dtype=torch.float64)
    np.testing.assert_allclose(npy(dc.depth), g['asl_f64_depth'], rtol=1e-15)
    dc = scan_io.load_points_npz_device(wr('scan.npz', 'fee_npz'), dtype=torch.float64)            # structured, with viewpoints
    np.testing.assert_allclose(npy(dc.vps), g['fee_f64_vps'], rtol=1e-15)
    np.testing.assert_allclose(npy(dc.depth), g['fee_f64_depth'], rtol=1e-14)
    np.testing.assert_allclose(npy(dc.dirs), g['fee_f64_dirs'], rtol=0, atol=1e-14)


@pytest.mark.parametrize('tag', ['f64', 'f64vp'])
def test_online_correction_sequence_golden(golden, tag):
    """online.correct_cloud = the correction node's per-scan statements (scripts/depth_correction:31-58: local_feature_cloud
    with the shadow filter -> model -> update_points) against the LIVE reference run on the same scan (tests/golden/
    online.npz, oracle/gen_golden.py:gen_online): the surviving points, their neighbourhoods (bit-exact), planarity mask,
    incidence angles, oriented normals, corrected depths and published points -- from a structured array without and
    with viewpoint fields, and from a cloud already on the device."""
    from numpy.lib.recfunctions import unstructured_to_structured
    from depth_correction_amd.config import Config
    from depth_correction_amd.model import ScaledPolynomial
    from depth_correction_amd.online import correct_cloud
    from depth_correction_amd.scan_io import cloud_on_device
    g = golden('online')
    cfg = Config(nn_k=int(g['nn_k']), nn_r=None, float_type='float64', device='cuda:0', log_filters=False,
                 shadow_neighborhood_angle=float(g['shadow_neighborhood_angle']),
                 shadow_angle_bounds=[float(np.radians(float(g['shadow_bound_deg']))), float('inf')])
    model = ScaledPolynomial(w=g['w'].tolist(), exponent=g['exponent'].tolist(), device='cuda:0')
    xyz = g['xyz']
    if tag == 'f64vp':
        vp = np.tile(np.array([[0.05, -0.02, 0.1]], dtype=np.float32), (len(xyz), 1))
        arr = unstructured_to_structured(np.concatenate([xyz + vp, vp], axis=1), names=['x', 'y', 'z', 'vp_x', 'vp_y', 'vp_z'])
        inputs = [arr]
    else:
        # (a structured array, a DepthCloud on the device, and the uploaded raw rows: the last takes dc_scan_prefilter -- from_points,
        #  shadow filter and cloud[mask] in one native call)
        inputs = [unstructured_to_structured(xyz, names=['x', 'y', 'z']),
                  cloud_on_device(torch.as_tensor(xyz, device='cuda:0'), dtype=torch.float64),
                  torch.as_tensor(xyz, device='cuda:0')]
    for inp in inputs:
        dc = correct_cloud(inp, model, cfg)
        assert len(dc) == len(g[tag + '_depth']) < len(xyz)                                   # the shadow filter removed the same rays
        assert np.array_equal(npy(dc.neighbors), g[tag + '_neighbors'])
        assert np.array_equal(npy(dc.mask).astype(np.uint8), g[tag + '_mask'])
        np.testing.assert_allclose(npy(dc.depth), g[tag + '_depth'], rtol=1e-12)
        np.testing.assert_allclose(npy(dc.inc_angles)[:, 0], g[tag + '_inc'], rtol=0, atol=2e-6)       # float32 in the fixture
        pts = npy(dc.get_points()).astype(np.float32)
        np.testing.assert_allclose(pts, g[tag + '_points'], rtol=0, atol=2e-6)
        np.testing.assert_allclose(npy(dc.vps).astype(np.float32), g[tag + '_vps'], rtol=0, atol=1e-7)
        # normals: eigenvectors of well separated smallest eigenvalues, oriented towards the sensor (depth_cloud.py:401-415)
        cos = np.abs(np.einsum('ni,ni->n', npy(dc.normals), g[tag + '_normals'].astype(np.float64)))
        planar = g[tag + '_mask'].astype(bool)
        assert cos[planar].min() > 1 - 1e-6
        same_side = np.einsum('ni,ni->n', npy(dc.normals), g[tag + '_normals'].astype(np.float64))[planar] > 0
        assert same_side.all()


@pytest.mark.parametrize('name', ['Linear', 'InvCos', 'ScaledPolynomial'])
def test_basis_rows_are_built_once_over_iterations(golden, name):
    """The basis rows of a plan (dc_points_basis: one pass over the points) are keyed by the identity and version of the pose and
    exponent tensors: models without exponents hand the kernels ONE constant exponent tensor (model._zero_exponent), so the rows
    are built by the first evaluation and reused by the following ones -- also while the weights change."""
    from depth_correction_amd import model as M
    from depth_correction_amd.eval import eval_loss_clouds, _plans
    from depth_correction_amd.loss import create_loss
    g = golden('room_k10')
    cfg = _cfg(g, loss='min_eigval_loss')
    clouds, poses, _, ns, mask = _setup(g, cfg)
    kw = dict(Linear=dict(w0=0.999, w1=1e-3, b=1e-3), InvCos=dict(p0=1e-3), ScaledPolynomial=dict(w=[1e-3, 2e-3], exponent=[2.0, 4.0]))[name]
    model = getattr(M, name)(device=cfg.device, **kw)
    loss_fun = create_loss(cfg)
    _plans.clear()
    rows = []
    for it in range(4):
        loss, _, _, _ = eval_loss_clouds([clouds], [poses], [None], [mask], [ns], model, loss_fun, cfg)
        loss.backward()
        with torch.no_grad():
            for p in model.parameters():
                p -= 1e-4 * p.grad
                p.grad = None
        plan = _plans.entries[0][-1] if hasattr(_plans, 'entries') and _plans.entries else None
        assert plan is not None and plan._basis is not None
        rows.append(plan._basis[1].data_ptr())
    assert len(set(rows)) == 1, rows


@pytest.mark.parametrize('tag,dtype', [('f64', torch.float64), ('f32', torch.float32)])
def test_shadow_mask_without_the_neighbour_table(golden, tag, dtype):
    """dc_shadow_filter (grid over the directions + one walk evaluating the angles, what local_feature_cloud runs on device
    clouds) gives the mask of update_dir_neighbors + filter_shadow_points bit for bit: the mask depends on the set of
    direction neighbours only (preproc.py:44-47 drops the table right after)."""
    from depth_correction_amd.depth_cloud import DepthCloud
    from depth_correction_amd.filters import filter_shadow_points, shadow_points_mask
    g = golden('shadow')
    dev = 'cuda:0'
    dc = DepthCloud(t(g[tag + '_vps'], dev), t(g[tag + '_dirs'], dev), t(g[tag + '_depth'], dev))
    dc.update_points()
    removed = 0
    for angle, lo_deg, hi in ((float(g['angle']), float(g['bounds_deg']), float('inf')), (0.03, 10.0, 2.5), (0.004, 0.0, 3.0)):
        bounds = [float(np.radians(lo_deg)), hi]
        fused = shadow_points_mask(dc, angle, list(bounds))
        two = dc.copy()
        two.update_dir_neighbors(angle=angle)
        want = filter_shadow_points(two, list(bounds), only_mask=True)
        assert torch.equal(fused, want)
        assert 0.05 < float(fused.double().mean()) <= 1.0
        removed = removed + int((~fused).sum())
    assert removed > 0
    # the mask of a device cloud through local_feature_cloud is the same as the reference's statement sequence
    from depth_correction_amd.config import Config
    from depth_correction_amd.preproc import local_feature_cloud
    cfg = Config(nn_k=8, nn_r=None, device=dev, log_filters=False, shadow_neighborhood_angle=float(g['angle']),
                 shadow_angle_bounds=[float(np.radians(float(g['bounds_deg']))), float('inf')])
    out = local_feature_cloud(dc.copy(), cfg)
    two = dc.copy()
    two.update_dir_neighbors(angle=float(g['angle']))
    kept = filter_shadow_points(two, cfg.shadow_angle_bounds)
    assert torch.equal(out.depth, kept.depth) and torch.equal(out.dirs, kept.dirs) and torch.equal(out.vps, kept.vps)


def test_cloud_slicing_with_a_device_mask_finds_the_rows_once(golden):
    """cloud[mask] with a bool mask on the device: every per-point field gathered through ONE nonzero(); same tensors as
    field[mask], neighbourhood fields dropped (depth_cloud.py:75-82)."""
    from depth_correction_amd.depth_cloud import DepthCloud
    g = golden('room_k10')
    x = t(g['scan0_xyz'], 'cuda:0')
    dc = DepthCloud.from_points(x)
    dc.update_all(k=6)
    mask = dc.eigvals[:, 0] < dc.eigvals[:, 0].median()
    cut = dc[mask]
    for f in DepthCloud.sliced_fields:
        a, b = getattr(dc, f), getattr(cut, f)
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a[mask], b)
    assert cut.neighbors is None and cut.weights is None and len(cut) == int(mask.sum())
    w = torch.ones((len(dc), 1), dtype=dc.depth.dtype, device='cuda:0', requires_grad=True)
    dd = DepthCloud(dc.vps, dc.dirs, dc.depth * w)[mask]
    dd.depth.sum().backward()
    assert torch.equal(w.grad[:, 0] != 0, mask)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('name', ['Polynomial', 'ScaledPolynomial'])
def test_polynomial_models_without_gradients_run_one_kernel(golden, name, dtype):
    """model(dc) / model.inverse(dc) outside the training loop (the node, evaluation under no_grad): dc_correct_depth against
    the tensor expression the same classes evaluate when a gradient is needed (model.py:181-215, 250-274) -- fp64 clouds to
    2 ulp (the product of the [n,P] x [P,1] GEMM is accumulated in a library-chosen order), float32 clouds to 1 float32 ulp;
    unmasked points keep their depth bit for bit."""
    from depth_correction_amd import model as M
    from depth_correction_amd.depth_cloud import DepthCloud
    g = golden('room_k10')
    dc = DepthCloud.from_points(t(g['scan0_xyz'], 'cuda:0').to(dtype))
    dc.update_all(k=8)
    model = getattr(M, name)(w=[1.5e-3, -2e-3, 4e-4], exponent=[2.0, 4.0, 0.5], device='cuda:0')
    masks = [None, dc.eigvals[:, 0] < dc.eigvals[:, 0].median()]
    eps = torch.finfo(dtype).eps
    for mask in masks:
        for fun in (model.correct_depth, model.inverse):
            with torch.no_grad():
                fast = fun(dc, mask).depth
            slow = fun(dc, mask).depth                      # w requires grad and grad mode is on: the tensor expression
            assert slow.requires_grad and not fast.requires_grad and fast.dtype == dtype and fast.shape == dc.depth.shape
            rel = ((fast - slow.detach()).abs() / slow.detach().abs()).max().item()
            assert rel <= 2 * eps, rel
            if mask is not None:
                assert torch.equal(fast[~mask], dc.depth[~mask]) and not torch.equal(fast[mask], dc.depth[mask])


@pytest.mark.parametrize('float_type', ['float32', 'float64'])
def test_local_feature_cloud_of_raw_device_rows_equals_the_cloud_path(golden, float_type):
    """local_feature_cloud on the uploaded rows [N, 3] (dc_scan_prefilter: from_points, points, shadow mask and cloud[mask] launched by
    one native call, update_points skipped, the validity weights left to update_features) against the same function on a DepthCloud
    (the separate calls): every field bit for bit, float32 and float64 clouds, with the eigenvalue bounds of the default
    configuration and with none."""
    from depth_correction_amd.config import Config
    from depth_correction_amd.preproc import local_feature_cloud
    from depth_correction_amd.scan_io import cloud_on_device
    g = golden('online')
    xyz = torch.as_tensor(np.concatenate([g['xyz'], g['xyz'][::7] * 1.01]), device='cuda:0')
    for bounds in (None, []):
        kw = {} if bounds is None else dict(eigenvalue_bounds=[], eigenvalue_ratio_bounds=[])
        cfg = Config(nn_k=8, nn_r=None, float_type=float_type, device='cuda:0', log_filters=False, shadow_neighborhood_angle=0.02,
                     shadow_angle_bounds=[float(np.radians(4.0)), float('inf')], **kw)
        a = local_feature_cloud(xyz, cfg)
        b = local_feature_cloud(cloud_on_device(xyz, dtype=cfg.torch_float_type()), cfg)
        assert 0 < len(a) == len(b) < len(xyz)
        for f in ('vps', 'dirs', 'depth', 'points', 'mean', 'cov', 'eigvals', 'eigvecs', 'normals', 'inc_angles', 'mask', 'neighbors',
                  'weights', 'distances'):
            x, y = getattr(a, f), getattr(b, f)
            assert (x is None) == (y is None), f
            if x is not None:
                assert x.dtype == y.dtype and torch.equal(x, y), f
    # an empty scan and one with a single ray go through both ways in
    for n in (0, 1):
        few = xyz[:n].contiguous()
        a, b = local_feature_cloud(few, cfg), local_feature_cloud(cloud_on_device(few, dtype=cfg.torch_float_type()), cfg)
        assert len(a) == len(b) == n and a.neighbors.shape == b.neighbors.shape == (n, 8)
