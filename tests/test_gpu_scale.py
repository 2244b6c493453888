"""BASELINE.json's full sizes (200 000-point scans, 2 M-point global cloud) through size-independent properties,
plus direct cKDTree / oracle checks where the CPU side finishes in seconds."""
import numpy as np
import pytest
import torch

import dc_oracle as O
from helpers import npy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def room():
    from depth_correction_amd.dataset import RoomBoxDataset
    ds = RoomBoxDataset(n_pts=200_000, n_poses=10, dtype=np.float32)
    scans = [np.stack([c[f] for f in 'xyz'], 1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    return scans, poses


def test_knn_200k_scan_bit_exact_vs_ckdtree(room):
    """Config 1 shape: one 200k-point scan, nn_k = 10 -- every index equal to cKDTree's, distances equal in fp64."""
    from depth_correction_amd import ops
    x = torch.as_tensor(room[0][0], device='cuda:0')
    dist, idx = ops.knn(x, 10)
    dref, iref = O.knn_ckdtree(room[0][0], 10)
    assert np.array_equal(npy(idx), iref)
    assert np.array_equal(npy(dist), dref)


def test_knn_2m_properties(room):
    """2 M-point global cloud: self first, ascending distances, distances recomputable, and every row of the table -- indices and
    fp64 distances -- equal to cKDTree's."""
    from depth_correction_amd import ops
    scans, poses = room
    xyz = np.concatenate([s.astype(np.float64) + p[:3, 3] for s, p in zip(scans, poses)]).astype(np.float32)
    x = torch.as_tensor(xyz, device='cuda:0')
    dist, idx = ops.knn(x, 10)
    n = len(xyz)
    assert torch.equal(idx[:, 0].long(), torch.arange(n, device='cuda:0')) and bool((dist[:, 0] == 0).all())
    assert bool((dist[:, 1:] >= dist[:, :-1]).all()) and bool((idx >= 0).all()) and bool((idx < n).all())
    d = (x.double()[idx.long()] - x.double()[:, None, :])
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    assert torch.allclose(torch.sqrt(d2), dist, rtol=1e-14, atol=0)
    from scipy.spatial import cKDTree
    # the WHOLE table against cKDTree (2 M queries, ~5 s on the box's 16 threads), not a sample of its rows
    x64 = xyz.astype(np.float64)
    dref, iref = cKDTree(x64).query(x64, 10, workers=-1)
    assert np.array_equal(npy(idx), iref) and np.array_equal(npy(dist), dref)
    ptr_, src = ops.knn_transpose(idx)
    assert int(ptr_[-1]) == n * 10 and bool((ptr_[1:] >= ptr_[:-1]).all())
    j = np.random.default_rng(1).integers(0, n, 1000)
    p, s, nb = npy(ptr_), npy(src), npy(idx)
    for jj in j:                                            # every listed centre really has jj as a neighbour
        assert all(jj in nb[i] for i in s[p[jj]:p[jj + 1]])


def test_local_features_batch_equals_per_scan(room):
    """The set-up's local feature clouds in one pass (pipeline.local_features_batch: the scans side by side on a lattice, one
    k-NN build, one feature launch) against the scans one by one: every array bit for bit -- and None (the caller's cue to
    run them one by one) when the batch cannot be proven identical: a neighbourhood that would reach into another scan, a
    coordinate the shift would round."""
    from depth_correction_amd.pipeline import local_features, local_features_batch
    scans = [s[:40_000 + 1_111 * i] for i, s in enumerate(room[0][:5])]
    got = local_features_batch(scans, 10, dtype=torch.float32)
    assert got is not None and len(got) == len(scans)
    for xyz, g in zip(scans, got):
        ref = local_features(xyz, k=10, dtype=torch.float32)
        assert set(ref) == set(g)
        for name in ref:
            assert torch.equal(ref[name], g[name]), name
    # two clusters a kilometre apart in a flat scan: the k nearest of a point reach across, and another scan's copy (stacked along
    # the thin axis) would be nearer
    rng = np.random.default_rng(0)
    flat = np.concatenate([rng.normal(size=(6, 3)) * [1, 1, 0.01], rng.normal(size=(6, 3)) * [1, 1, 0.01] + [1000.0, 0, 0]]).astype(np.float32)
    assert local_features_batch([flat, flat + np.float32(0.5)], 8, dtype=torch.float32) is None
    # coordinates that vanish next to the shift (1e-12 + 9.0 is not exact), whichever axis the scans are stacked along: not
    # provably identical
    tiny = scans[0].copy()
    tiny[7] = [1e-12, 2e-12, 3e-12]
    assert local_features_batch([scans[1], tiny], 10, dtype=torch.float32) is None


def test_fused_loss_2m_properties(room):
    """Config 2 shape.  Layout independence (Morton-sorted vs scan-major, q32 vs fp32 points), rigid-motion
    invariance, and the hand-derived dL/dw against central differences of the loss, all on the 2 M-point cloud."""
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import SequencePlan
    scans, poses = room
    plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32)
    dev = plan.device
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
    T = info['poses']

    def run(p, ww=w, TT=T):
        out = torch.zeros(2 + 4 + 12 * p.n_scans, dtype=torch.float64, device=dev)
        p.eval_native(ww, e, p.poses12(TT), out, want_pose=True)
        return npy(out)
    base = run(plan)
    assert base[1] == plan.count and 0.3 * plan.n < base[1] < plan.n
    flat = SequencePlan(info['clouds'], T, info['neighbors'], info['mask'], spatial_sort=False)
    # same numbers in any point order.  (The sorted plan's pose evaluation is the one-launch kernel, whose rows carry dd'/dw_k
    # rounded to float32 like the basis rows of the model-only step: d' moves by ~1e-9 m, a few points round to the next q32
    # grid value (4.7e-8 m), the loss moves by parts in 1e10; the scan-major plan runs the general path.)
    np.testing.assert_allclose(run(flat)[:2], base[:2], rtol=2e-9)
    # the pose kernel's second sweep is float32 (as the model-only step's): parts in 1e8 on dL/dw
    np.testing.assert_allclose(run(flat)[2:4], base[2:4], rtol=1e-6)
    # (the small entries of a scan's 3 x 4 gradient are differences of large sums of float32 edge terms, summed in different orders)
    np.testing.assert_allclose(run(flat)[6:], base[6:], rtol=1e-6, atol=1e-6 * np.abs(base[6:]).max())
    f32 = SequencePlan(info['clouds'], T, info['neighbors'], info['mask'], point_format='float')
    np.testing.assert_allclose(run(f32)[:2], base[:2], rtol=1e-5)                  # fp32 points: the 1e-5 bar
    # rigid motion of the whole map leaves the loss and dL/dw unchanged
    a = 0.3
    G = torch.tensor([[np.cos(a), -np.sin(a), 0, 5.0], [np.sin(a), np.cos(a), 0, -3.0], [0, 0, 1, 2.0], [0, 0, 0, 1]],
                     dtype=torch.float64, device=dev)
    moved = run(plan, TT=G @ T)
    np.testing.assert_allclose(moved[0], base[0], rtol=2e-6)       # q32 grid is not rotation invariant: 3e-8 m steps
    np.testing.assert_allclose(moved[2:4], base[2:4], rtol=2e-4, atol=1e-3 * np.abs(base[2:4]).max())
    # dL/dw vs central differences (loss is smooth in w; fp64 accumulation)
    for k in range(2):
        h = 2e-5
        dw = torch.zeros(2, dtype=torch.float64, device=dev)
        dw[k] = h
        num = (run(plan, ww=w + dw)[0] - run(plan, ww=w - dw)[0]) / (2 * h)
        np.testing.assert_allclose(base[2 + k], num, rtol=5e-4)      # O(h^2) truncation + relu kinks
    # translation gradient of all poses sums to ~0 (loss invariant to a common shift)
    gT = base[6:].reshape(plan.n_scans, 3, 4)
    assert np.abs(gT[:, :, 3].sum(0)).max() <= 1e-6 * np.abs(gT[:, :, 3]).sum()


def test_oracle_on_200k_sample(room):
    """One 200k-point scan pair region: the fused loss / gradient against the oracle (fp64) on a 2-scan sequence."""
    from depth_correction_amd.pipeline import build_sequence
    scans, poses = room
    sub = [s[:60_000] for s in scans[:2]]
    plan, info = build_sequence(sub, poses[:2], k=10, dtype=torch.float32)
    dev = plan.device
    out = torch.zeros(2 + 4 + 24, dtype=torch.float64, device=dev)
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
    plan.eval_native(w, e, plan.poses12(info['poses']), out)
    oc = [dict(vps=c['vps'].double().cpu(), dirs=c['dirs'].double().cpu(), depth=c['depth'].double().cpu(),
               inc=c['inc_angles'].double().cpu(), mask=c['mask'].cpu()) for c in info['clouds']]
    wo = torch.tensor([[1e-3, 2e-3]], dtype=torch.float64, requires_grad=True)
    _, ind = O.knn_ckdtree(info['points0'].double().cpu().numpy(), 10)
    assert np.array_equal(ind, npy(info['neighbors']))
    lo, _ = O.eval_sequence(oc, info['poses'].cpu(), wo, e.cpu().reshape(1, -1), torch.as_tensor(ind), info['mask'].cpu(),
                            reduction='sum')
    lo.backward()
    o = npy(out)
    np.testing.assert_allclose(o[0], lo.item(), rtol=1e-5)
    np.testing.assert_allclose(o[2:4], npy(wo.grad).ravel(), rtol=1e-5, atol=1e-6 * np.abs(npy(wo.grad)).max())


@pytest.mark.timeout(600)
@pytest.mark.parametrize('dtype,rtol_loss,rtol_grad', [(torch.float32, 1e-5, 1e-5), (torch.float64, 1e-9, 1e-7)])
def test_c2_full_size_loss_and_gradient_vs_oracle(room, dtype, rtol_loss, rtol_grad):
    """BASELINE config 2 at full size (10 x 200 k points, N = 2 M, K = 10, ScaledPolynomial, normalised min-eigenvalue
    loss): loss, masked-point count and dL/dw of the HIP path (fp32 inputs, q32 points, block tables) against the CPU
    oracle in fp64 on the same inputs -- the north-star tolerance 1e-5.  The neighbour table is the GPU's (bit-exact
    against cKDTree in the tests above), so the oracle spends its time on the path proper.  Also with fp64 device data
    (the reference's default float type), where the bar is 1e-9."""
    from depth_correction_amd.pipeline import build_sequence
    scans, poses = room
    plan, info = build_sequence(scans, poses, k=10, dtype=dtype)
    dev = plan.device
    out = torch.zeros(2 + 4 + 12 * plan.n_scans, dtype=torch.float64, device=dev)
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
    plan.eval_native(w, e, plan.poses12(info['poses']), out)
    o = npy(out)
    oc = [dict(vps=c['vps'].double().cpu(), dirs=c['dirs'].double().cpu(), depth=c['depth'].double().cpu(),
               inc=c['inc_angles'].double().cpu(), mask=c['mask'].cpu()) for c in info['clouds']]
    wo = torch.tensor([[1e-3, 2e-3]], dtype=torch.float64, requires_grad=True)
    threads = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        lo, _ = O.eval_sequence(oc, info['poses'].cpu(), wo, e.cpu().reshape(1, -1), info['neighbors'].long().cpu(),
                                info['mask'].cpu(), reduction='sum')
        lo.backward()
    finally:
        torch.set_num_threads(threads)
    assert o[1] == float(info['mask'].sum().item()) and o[1] > 1.0e6
    np.testing.assert_allclose(o[0], lo.item(), rtol=rtol_loss)
    g = npy(wo.grad).ravel()
    np.testing.assert_allclose(o[2:4], g, rtol=rtol_grad, atol=0.1 * rtol_grad * np.abs(g).max())


def test_c1_full_size_features_vs_oracle(room):
    """BASELINE config 1 at full size (one 200 k-point scan, K = 10, covariance + eigen-decomposition forward through
    DepthCloud's operator, fp32 inputs): eigenvalues within 1e-5 relative (plus LAPACK's absolute floor), covariances
    and incidence angles against the fp64 oracle on the same fp32 points and the same (bit-exact) neighbour table."""
    from depth_correction_amd import pipeline
    scans, _ = room
    loc = pipeline.local_features(scans[0], k=10, dtype=torch.float32, eigenvalue_ratio_bounds=None)
    x = loc['points'].double().cpu()
    _, ind = O.knn_ckdtree(x.numpy(), 10)
    assert np.array_equal(ind, npy(loc['neighbors']))
    ref = O.features(x, torch.as_tensor(ind), loc['dirs'].double().cpu())
    ev, rv = npy(loc['eigvals']).astype(np.float64), npy(ref['eigvals'])
    # the kernel stores fp32 eigenvalues: half an fp32 ulp on top of the 1e-5 bar
    assert_eigvals = np.abs(ev - rv) <= 1e-5 * np.abs(rv) + 1e-12 * np.abs(rv).max(-1, keepdims=True) + 6e-8 * np.abs(rv)
    assert assert_eigvals.all(), 'worst rel %.3g' % (np.abs(ev - rv) / np.abs(rv))[~assert_eigvals].max()
    # incidence angles where the normal is well defined (lambda0 separated from lambda1)
    ok = (rv[:, 1] - rv[:, 0]) > 1e-3 * rv[:, 2]
    assert ok.mean() > 0.9
    np.testing.assert_allclose(npy(loc['inc_angles']).ravel()[ok], npy(ref['inc_angles']).ravel()[ok], atol=2e-4)


@pytest.mark.timeout(600)
def test_c4_full_size_icp_loss_and_gradients_vs_oracle():
    """BASELINE config 4 shape at full size: KITTI-like scans of 64 x 2048 rays, depth 5-25 m and 0.2 m voxel
    pre-filters, radius neighbourhoods (nn_r = 0.4), model + per-pose corrections, point-to-plane ICP over all
    consecutive pairs.  The whole pipeline runs on the GPU; the oracle (fp64, CPU) takes the GPU's local features
    (normals of collinear ring neighbourhoods are arbitrary eigenvectors) and the GPU's correspondences
    (bit-exact vs cKDTree, tested above) and repeats model -> transform -> point_to_plane_dist and its gradients."""
    from depth_correction_amd.config import Config, Loss, PoseCorrection
    from depth_correction_amd.dataset import KittiLikeDataset
    from depth_correction_amd.loss import icp_loss, icp_correspondences
    from depth_correction_amd.model import ScaledPolynomial
    from depth_correction_amd.preproc import filtered_cloud, local_feature_cloud
    from depth_correction_amd.transform import xyz_axis_angle_to_matrix
    dev = 'cuda:0'
    cfg = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0,
                 max_depth=25.0, vp_dispersion_bounds=[])
    ds = KittiLikeDataset(n_poses=4)
    raw = [(c, p) for c, p in ds]
    assert all(len(c) > 120_000 for c, _ in raw)
    clouds = [local_feature_cloud(filtered_cloud(c, cfg), cfg) for c, _ in raw]
    assert all(5_000 < len(c) < 131_072 for c in clouds), [len(c) for c in clouds]
    poses0 = torch.as_tensor(np.stack([p for _, p in raw]), dtype=torch.float64, device=dev)
    pts = [c.transform(T).to_points() for c, T in zip(clouds, poses0)]
    masks = []
    for j in range(len(clouds) - 1):
        m1, i2, _ = icp_correspondences(pts[j], pts[j + 1], 0.3)
        masks.append((m1, i2))
    model = ScaledPolynomial(w=[1e-3, -1e-3], exponent=[2.0, 4.0], device=dev)
    pd = torch.zeros((len(clouds), 6), dtype=torch.float64, device=dev)
    pd[1:] = torch.tensor([0.01, -0.02, 0.005, 0.002, -0.001, 0.003], dtype=torch.float64, device=dev)
    pd.requires_grad_(True)
    poses = torch.matmul(poses0, xyz_axis_angle_to_matrix(pd))
    loss, _ = icp_loss([clouds], [poses], model, masks=[masks], icp_point_to_plane=True, icp_inlier_ratio=0.3)
    loss.backward()
    # ---- oracle on the same local clouds / correspondences
    w = torch.tensor([[1e-3, -1e-3]], dtype=torch.float64, requires_grad=True)
    e = torch.tensor([[2.0, 4.0]], dtype=torch.float64)
    pdo = pd.detach().cpu().clone().requires_grad_(True)
    To = torch.matmul(poses0.cpu(), O.xyz_axis_angle_to_matrix(pdo))
    opts, onrm = [], []
    for c, Ts in zip(clouds, To):
        d = O.model_apply(c.depth.double().cpu(), c.inc_angles.double().cpu(), c.mask.cpu(), w, e)
        v, r, n = O.transform_cloud(c.vps.double().cpu().expand(len(c), 3), c.dirs.double().cpu(), Ts,
                                    normals=c.normals.double().cpu())
        opts.append(O.points_from(v, r, d)), onrm.append(n)
    lo = O.point_to_plane(opts, onrm, [(m1.cpu(), i2.cpu()) for m1, i2 in masks])
    lo.backward()
    # Bars: north_star's 1e-5 relative, plus the floor the fp32 cast of loss.py:436-437 sets.  Both sides round the fp64 world points
    # to fp32 before the distances are formed (and treat the cast as the identity in the backward pass), so they differ only where a
    # coordinate sits within one fp64 ulp of an fp32 rounding boundary and the two evaluation orders of R x + t land on different
    # sides: probability 2^-29 per coordinate, 6 coordinates per correspondence, m ~ 1.4e5 correspondences -> ~2e-3 such
    # coordinates expected.  One flip moves its term by at most ulp32(32 m) = 1.9e-6 m times the pair weight 0.5 / (m n_pairs)
    # ~ 4e-6: 7e-12 on a loss of ~1e-2, and a gradient entry by at most that times |dx/dtheta| <= 32 m: 2e-10.  atol = 1e-9.
    np.testing.assert_allclose(loss.item(), lo.item(), rtol=1e-5, atol=1e-11)
    np.testing.assert_allclose(npy(model.w.grad).ravel(), npy(w.grad).ravel(), rtol=1e-5, atol=1e-9)
    ref = npy(pdo.grad)
    np.testing.assert_allclose(npy(pd.grad), ref, rtol=1e-5, atol=1e-9)


def test_global_mask_and_incidence_angles_at_full_size_vs_oracle(room):
    """The set-up outputs the per-iteration tests above take from the GPU, now against the oracle's OWN values at the C2
    size (N = 2 M): local incidence angles and local planarity masks of all ten 200 k-point scans, and the global mask
    (valid-neighbour count + eigenvalue-ratio bounds on the 2 M-point cloud, preproc.py:122-164).  fp32 device data vs
    the fp64 oracle on the same fp32 inputs: a mask entry may differ only where an eigenvalue ratio lies within 1e-4
    relative of a bound (the eigenvalues themselves agree to 1e-5)."""
    from depth_correction_amd.pipeline import build_sequence, DEFAULT_RATIO_BOUNDS
    scans, poses = room
    plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32)
    threads = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        def near_bound(ev):
            near = np.zeros(len(ev), bool)
            with np.errstate(divide='ignore', invalid='ignore'):
                for i, j, lo, hi in DEFAULT_RATIO_BOUNDS:
                    r = ev[:, int(i)] / ev[:, int(j)]
                    for b in (lo, hi):
                        near |= np.abs(r - b) <= 1e-4 * max(abs(b), 1e-3)
            return near

        omasks = []
        for c in info['clouds']:
            x = (c['depth'].double() * c['dirs'].double()).cpu()                     # the same fp32 inputs, in fp64
            nbr = c['neighbors'].long().cpu()                                         # bit-exact vs cKDTree (tests above)
            f = O.features(x, nbr, c['dirs'].double().cpu())
            np.testing.assert_allclose(npy(c['inc_angles']).ravel(), npy(f['inc_angles']).ravel(), rtol=0, atol=2e-4)
            assert np.abs(npy(c['inc_angles']).ravel() - npy(f['inc_angles']).ravel()).mean() < 2e-6
            m = npy(O.local_mask(f['eigvals'], None, DEFAULT_RATIO_BOUNDS))
            diff = m != npy(c['mask'])
            assert not np.any(diff & ~near_bound(npy(f['eigvals']))) and diff.mean() < 1e-3
            omasks.append(torch.as_tensor(m))
        x0 = info['points0'].double().cpu()
        nbr = info['neighbors'].long().cpu()
        f0 = O.features(x0, nbr, torch.cat([c['dirs'] for c in info['clouds']]).double().cpu())
        om = npy(O.global_mask(torch.cat([c['mask'] for c in info['clouds']]).cpu(), nbr, f0['eigvals'], min_valid_neighbors=5,
                               eigenvalue_ratio_bounds=DEFAULT_RATIO_BOUNDS))
    finally:
        torch.set_num_threads(threads)
    gm = npy(info['mask'])
    diff = gm != om
    assert not np.any(diff & ~near_bound(npy(f0['eigvals']))), int((diff & ~near_bound(npy(f0['eigvals']))).sum())
    assert diff.mean() < 1e-3 and 0.5 < gm.mean() < 0.95


def test_q32_overflow_is_reported_and_large_maps_use_fp64():
    """The fixed-point format is sized for 4x the extent of the initial map: an evaluation whose poses carry points beyond
    it raises the plan's status flag and returns a NaN loss instead of silently saturated coordinates; point_format='auto'
    leaves q32 for maps whose extent would push its resolution past the 1e-5 parity bar and keeps fp64 points, with which
    a 600 m KITTI-360-shaped sequence matches the oracle."""
    from depth_correction_amd.dataset import KittiLikeDataset
    from depth_correction_amd.pipeline import build_sequence
    dev = 'cuda:0'
    ds = KittiLikeDataset(n_poses=4, n_rings=32, n_azimuth=1024)
    scans = [np.stack([c[f] for f in 'xyz'], 1).astype(np.float32) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)

    def evaluate(plan, T):
        out = torch.zeros(2 + 4 + 12 * plan.n_scans, dtype=torch.float64, device=dev)
        plan.eval_native(w, e, plan.poses12(torch.as_tensor(T, dtype=torch.float64, device=dev)), out)
        return npy(out)

    plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32, point_format='q32')
    assert plan.point_format == 'q32'
    ok = evaluate(plan, poses)
    assert np.isfinite(ok[0]) and not plan.overflowed()
    far = poses.copy()
    far[:, 0, 3] += 5000.0                                       # far outside 4x the map's extent
    bad = evaluate(plan, far)
    assert np.isnan(bad[0]) and plan.overflowed()

    # the same scans spread over 600 m (poses 200 m apart): q32 would have a 5e-7 m grid -> auto keeps fp64 points
    wide = poses.copy()
    wide[:, 0, 3] = 200.0 * np.arange(len(wide))
    plan2, info2 = build_sequence(scans, wide, k=10, dtype=torch.float32)
    assert plan2.point_format == 'f64' and plan2.qfmt is None
    o = evaluate(plan2, wide)
    oc = [dict(vps=c['vps'].double().cpu(), dirs=c['dirs'].double().cpu(), depth=c['depth'].double().cpu(),
               inc=c['inc_angles'].double().cpu(), mask=c['mask'].cpu()) for c in info2['clouds']]
    wo = torch.tensor([[1e-3, 2e-3]], dtype=torch.float64, requires_grad=True)
    lo, _ = O.eval_sequence(oc, torch.as_tensor(wide), wo, e.cpu().reshape(1, -1), info2['neighbors'].long().cpu(),
                            info2['mask'].cpu(), reduction='sum')
    lo.backward()
    np.testing.assert_allclose(o[0], lo.item(), rtol=1e-5)
    g = npy(wo.grad).ravel()
    np.testing.assert_allclose(o[2:4], g, rtol=1e-5, atol=1e-6 * np.abs(g).max())
    # and a compact map still takes q32
    plan3, _ = build_sequence(scans, poses, k=10, dtype=torch.float32)
    assert plan3.point_format == ('q32' if plan3.qfmt is not None else 'f64')


@pytest.mark.timeout(900)
def test_c2_full_size_chained_steps_equal_ordinary_steps_and_the_oracle_adam_loop(room):
    """The path bench.py TIMES, at the size it is timed at (N = 2 M: 7 816 + 8 blocks, far more than are resident at once,
    so blocks of a chained launch really do wait for weights that its leading blocks publish): (a) 25 chained steps
    (SequenceTrainer(chained=True) = dc_sequence_step_chained, one flush in the middle) equal 25 ordinary steps -- sums to
    1e-11, weights to 1e-10; (b) the first 3 steps' loss, dL/dw and updated weights equal the oracle's eval_sequence +
    torch.optim.Adam loop (train.py:300-312) to the north-star 1e-5."""
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import SequenceTrainer
    scans, poses = room
    plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32)
    w0, e0, lr = [1e-3, 2e-3], [2.0, 4.0], 1e-3
    plain = SequenceTrainer([plan], w0, e0, [info['poses']], lr=lr)
    ref, ref_w = [], []
    for _ in range(25):
        ref.append(npy(plain.step()).copy())
        ref_w.append(npy(plain.w).copy())
    chain = SequenceTrainer([plan], w0, e0, [info['poses']], lr=lr, chained=True)
    got = []
    for it in range(25):
        prev = npy(chain.step()).copy()
        assert chain.chained
        if it not in (0, 12):                                  # nothing pending at the start and right after the flush
            got.append(prev)
        if it in (11, 24):
            got.append(npy(chain.flush()).copy())
            np.testing.assert_allclose(npy(chain.w), ref_w[it], rtol=1e-10)
    torch.cuda.synchronize()
    assert len(got) == 25 and chain.t == 25 and not plan.chain_timed_out() and not plan.overflowed()
    for a, b in zip(got, ref):
        assert a[1] == b[1] and a[1] > 1.0e6
        np.testing.assert_allclose(a[0], b[0], rtol=1e-11)
        np.testing.assert_allclose(a[2:], b[2:], rtol=1e-9, atol=1e-12 * np.abs(b[2:]).max())
    np.testing.assert_allclose(npy(chain.w), npy(plain.w), rtol=1e-10)
    assert abs(npy(chain.w)[0] - w0[0]) > 1e-4                 # 25 Adam steps of 1e-3 moved the weights

    # (b) the oracle's loop on the same inputs: 3 iterations of eval_sequence (mean) -> backward -> torch.optim.Adam
    oc = [dict(vps=c['vps'].double().cpu(), dirs=c['dirs'].double().cpu(), depth=c['depth'].double().cpu(),
               inc=c['inc_angles'].double().cpu(), mask=c['mask'].cpu()) for c in info['clouds']]
    wo = torch.nn.Parameter(torch.tensor([w0], dtype=torch.float64))
    eo = torch.tensor([e0], dtype=torch.float64)
    opt = torch.optim.Adam([wo], lr=lr)
    nbr, mask, T = info['neighbors'].long().cpu(), info['mask'].cpu(), info['poses'].cpu()
    threads = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        for it in range(3):
            opt.zero_grad()
            lo, _ = O.eval_sequence(oc, T, wo, eo, nbr, mask, reduction='mean')
            lo.backward()
            cnt = got[it][1]
            np.testing.assert_allclose(got[it][0] / cnt, lo.item(), rtol=1e-5)
            g = npy(wo.grad).ravel()
            np.testing.assert_allclose(got[it][2:] / cnt, g, rtol=1e-5, atol=1e-6 * np.abs(g).max())
            opt.step()
            # Adam normalises the gradient (lr * m / sqrt(v)): the first updates are +-lr whatever the magnitude, so the bar
            # on the weights is the bar on the gradient's direction
            np.testing.assert_allclose(ref_w[it], npy(wo).ravel(), rtol=1e-5)
    finally:
        torch.set_num_threads(threads)


def test_c2_full_size_pose_gradients_vs_oracle(room):
    """Pose-mode evaluation at N = 2 M (what train() runs with pose corrections: train.py:300-312, eval.py:68-82): dL/d[R|t] of
    all ten scans and, through the pose chain, dL/d pose_deltas against the oracle's autograd (eval_sequence with
    T = T0 . xyz_axis_angle_to_matrix(delta) at delta = 0), together with loss and dL/dw -- the sums the grouped per-scan
    reduction of the backward produces (plan.scan_seg).  Bar: 1e-5 of each scan's largest entry (the entries of one scan's
    3 x 4 gradient differ by orders of magnitude; the small ones are differences of large sums)."""
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.transform import corrected_poses
    scans, poses = room
    plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32)
    assert plan.scan_seg is not None and plan.scan_seg.shape == ((plan.n + 255) // 256, 2 * plan.n_scans + 1)
    dev = plan.device
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
    deltas = torch.zeros((plan.n_scans, 6), dtype=torch.float64, device=dev, requires_grad=True)
    T = corrected_poses(info['poses'], deltas)
    out = torch.zeros(2 + 4 + 12 * plan.n_scans, dtype=torch.float64, device=dev)
    plan.eval_native(w, e, plan.poses12(T.detach()), out, want_pose=True)
    gT = out[6:].reshape(plan.n_scans, 3, 4)
    T.backward(torch.cat([gT, torch.zeros((plan.n_scans, 1, 4), dtype=torch.float64, device=dev)], dim=1))
    got = npy(out)
    got_d = npy(deltas.grad)

    oc = [dict(vps=c['vps'].double().cpu(), dirs=c['dirs'].double().cpu(), depth=c['depth'].double().cpu(),
               inc=c['inc_angles'].double().cpu(), mask=c['mask'].cpu()) for c in info['clouds']]
    wo = torch.tensor([[1e-3, 2e-3]], dtype=torch.float64, requires_grad=True)
    eo = torch.tensor([[2.0, 4.0]], dtype=torch.float64)
    To = info['poses'].cpu().clone().requires_grad_(True)
    do = torch.zeros((plan.n_scans, 6), dtype=torch.float64, requires_grad=True)
    threads = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        lo, _ = O.eval_sequence(oc, To, wo, eo, info['neighbors'].long().cpu(), info['mask'].cpu(), pose_deltas=do, reduction='sum')
        lo.backward()
    finally:
        torch.set_num_threads(threads)
    np.testing.assert_allclose(got[0], lo.item(), rtol=1e-5)
    gw = npy(wo.grad).ravel()
    np.testing.assert_allclose(got[2:4], gw, rtol=1e-5, atol=1e-6 * np.abs(gw).max())
    ref_T = npy(To.grad)[:, :3, :]                                  # dL/d(T0): at delta = 0 the applied pose IS T0
    ref_d = npy(do.grad)
    got_T = got[6:].reshape(plan.n_scans, 3, 4)
    for s in range(plan.n_scans):
        np.testing.assert_allclose(got_T[s], ref_T[s], rtol=1e-5, atol=1e-5 * np.abs(ref_T[s]).max(), err_msg='scan %d [R|t]' % s)
        np.testing.assert_allclose(got_d[s], ref_d[s], rtol=1e-5, atol=1e-5 * np.abs(ref_d[s]).max(), err_msg='scan %d deltas' % s)
    assert np.abs(ref_d).max() > 0 and np.abs(got_T[:, :, 3].sum(0)).max() <= 1e-6 * np.abs(got_T[:, :, 3]).sum()


@pytest.mark.parametrize('k, n_terms, loss, normalization, sqrt', [
    (10, 2, 'min_eigval_loss', True, False),
    (10, 1, 'min_eigval_loss', False, True),
    (4, 2, 'trace_loss', False, False),
    (16, 1, 'min_eigval_loss', True, True),
])
def test_pose_kernel_equals_three_kernel_path(room, k, n_terms, loss, normalization, sqrt):
    """The one-launch pose evaluation (consistency_step_pose_kernel: rows from the local basis, gradients of the staged rows summed
    in LDS as integers, per-scan sums in the world frame) against the general path (dc_points_fwd + forward + backward over the
    transposed table, dc_set_option(7, 1)) on the same plan: loss, count, dL/dw and dL/d[R|t] of every scan, for perturbed poses,
    a ragged last block and every table width the kernel is built for.  Both compute the sweeps in float32 on the q32 grid, in
    different orders: 1e-6 of each scan's largest entry."""
    from depth_correction_amd import _native as nv
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import KernelTimer
    scans, poses = room
    plan, info = build_sequence([s[:30_011] for s in scans[:4]], poses[:4], k=k, dtype=torch.float32, loss=loss,
                                normalization=normalization, sqrt=sqrt, min_valid_neighbors=min(5, k - 1))
    dev = plan.device
    rng = np.random.default_rng(k)
    w = torch.tensor([1e-3, 2e-3][:n_terms], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0][:n_terms], dtype=torch.float64, device=dev)
    from depth_correction_amd.transform import corrected_poses
    deltas = torch.as_tensor(rng.normal(size=(plan.n_scans, 6)) * 2e-3, dtype=torch.float64, device=dev)
    P = plan.poses12(corrected_poses(info['poses'], deltas))
    res = []
    for three in (0, 1):
        nv.check(nv.lib().dc_set_option(7, three), 'dc_set_option')
        try:
            out = torch.zeros(2 + 2 * n_terms + 12 * plan.n_scans, dtype=torch.float64, device=dev)
            with KernelTimer(every=1) as kt:
                plan.eval_native(w, e, P, out, want_pose=True)
                torch.cuda.synchronize()
                names = kt.kernels()
            assert ('consistency_step_pose_kernel<%d, %d>' % (k, n_terms) in names['consistency_fwd']) == (three == 0), names
            res.append(npy(out).copy())
        finally:
            nv.check(nv.lib().dc_set_option(7, 0), 'dc_set_option')
    a, b = res
    assert a[1] == b[1] and a[1] > 1000
    np.testing.assert_allclose(a[0], b[0], rtol=2e-8)           # (a few points land on the next q32 grid value: see test_fused_loss_2m_properties)
    np.testing.assert_allclose(a[2:2 + n_terms], b[2:2 + n_terms], rtol=1e-6, atol=1e-7 * np.abs(b[2:2 + n_terms]).max())
    assert not a[2 + n_terms:2 + 2 * n_terms].any()
    ga, gb = a[2 + 2 * n_terms:].reshape(-1, 3, 4), b[2 + 2 * n_terms:].reshape(-1, 3, 4)
    assert np.abs(gb).max() > 0
    for s_ in range(plan.n_scans):
        np.testing.assert_allclose(ga[s_], gb[s_], rtol=1e-6, atol=1e-6 * np.abs(gb[s_]).max(), err_msg='scan %d' % s_)
    # bit-reproducible: the integer sums do not depend on the order the wavefronts' atomics land in
    out2 = torch.zeros_like(out)
    plan.eval_native(w, e, P, out2, want_pose=True)
    plan.eval_native(w, e, P, out, want_pose=True)
    assert torch.equal(out, out2)


def test_ball_neighbourhood_plan_grouped_by_row_length(room):
    """Ball neighbourhoods (nn_r): SequencePlan(degree_group=True) orders every block's points by (mask, row length) so that rows of
    similar length share wavefronts of the ragged one-pass kernel -- a pure re-ordering: the same loss, count and dL/dw as the
    (mask, scan) grouping through the chained step, and the same pose gradients (which then take the un-grouped backward: no scan
    ranges in such a plan)."""
    from depth_correction_amd.filters import filter_grid
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import KernelTimer
    scans, poses = room
    rng = np.random.default_rng(3)
    kept = [filter_grid(s[:60_000], 0.15, keep='random', rng=rng) for s in scans[:4]]
    res, blocks = {}, {}
    for by_degree, heavy in ((True, True), (False, True), (True, False)):
        plan, info = build_sequence(kept, poses[:4], k=None, r=0.3, dtype=torch.float32, degree_group=by_degree, heavy_first=heavy)
        assert (plan.scan_seg is None) == by_degree and plan.fwd_table is not None
        nbf = plan.n // 256
        blocks[by_degree, heavy] = torch.sort(plan.order[:nbf * 256].reshape(nbf, 256), dim=1).values.cpu().numpy()
        dev = plan.device
        w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
        e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
        P = plan.poses12(info['poses'])
        out = torch.zeros(2 + 4 + 12 * plan.n_scans, dtype=torch.float64, device=dev)
        with KernelTimer(every=1) as kt:
            plan.eval_native(w, e, P, out, want_grad=True, want_pose=False)
            torch.cuda.synchronize()
            assert kt.kernels()['consistency_fwd'].startswith('consistency_step_ragged_q32_kernel'), kt.kernels()
        model_only = npy(out).copy()
        plan.eval_native(w, e, P, out, want_grad=True, want_pose=True)
        res[by_degree, heavy] = (model_only, npy(out).copy())
    # heavy_first: the same blocks of 256 points (as sets), numbered so that the grid starts with the longest rows; the blocks
    # after the first ones of the eight XCDs hold shorter and shorter longest rows
    a, b = blocks[True, True], blocks[True, False]
    assert a.shape == b.shape and not np.array_equal(a, b)
    assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
    deg = npy((info['neighbors'] >= 0).sum(1))
    nb = (len(deg) + 255) // 256
    per = (nb + 7) // 8
    longest = [deg[a[(g % 8) * per + g // 8]].max() for g in range(per * 8) if (g % 8) * per + g // 8 < len(a)]
    assert all(x >= y for x, y in zip(longest, longest[1:]))
    np.testing.assert_allclose(res[True, True][0], res[True, False][0], rtol=1e-9)
    (m1, p1), (m0, p0) = res[True, True], res[False, True]
    assert m1[1] == m0[1] > 1000
    np.testing.assert_allclose(m1[0], m0[0], rtol=1e-11)
    np.testing.assert_allclose(m1[2:4], m0[2:4], rtol=1e-6)
    np.testing.assert_allclose(p1[:2], p0[:2], rtol=1e-9)
    g1, g0 = p1[6:].reshape(-1, 3, 4), p0[6:].reshape(-1, 3, 4)
    for s_ in range(4):
        np.testing.assert_allclose(g1[s_], g0[s_], rtol=1e-6, atol=1e-6 * np.abs(g0[s_]).max())


def test_ball_pose_backward_with_large_lds_tiles(room):
    """Pose gradients on ball neighbourhoods of r = 0.4 m (0.2 m voxels: 185 neighbours per point, ~1 500 centres reference a block):
    the backward's record tile exceeds a workgroup's default LDS and takes the run kernel with a larger dynamic allocation
    (consistency_bwd_impl) -- the same loss and gradients as the un-staged kernels (dc_set_option(0, 1): gathers from global memory)."""
    from depth_correction_amd import _native as nv
    from depth_correction_amd.filters import filter_grid
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import KernelTimer
    scans, poses = room
    rng = np.random.default_rng(135)
    kept = [filter_grid(s, 0.2, keep='random', rng=rng) for s in scans]
    plan, info = build_sequence(kept, poses, k=None, r=0.4, dtype=torch.float32, degree_group=False)
    dev = plan.device
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
    P = plan.poses12(info['poses'])
    res = []
    for no_tab in (0, 1):
        nv.check(nv.lib().dc_set_option(0, no_tab), 'dc_set_option')
        try:
            out = torch.zeros(2 + 4 + 12 * plan.n_scans, dtype=torch.float64, device=dev)
            with KernelTimer(every=1) as kt:
                plan.eval_native(w, e, P, out, want_grad=True, want_pose=True)
                torch.cuda.synchronize()
                name = kt.kernels()['consistency_bwd']
            assert name.startswith('consistency_bwd_runs_kernel') == (no_tab == 0), name
            res.append(npy(out).copy())
        finally:
            nv.check(nv.lib().dc_set_option(0, 0), 'dc_set_option')
    assert plan.bwd_table.max_rows * 32 > 44 * 1024                    # the case this test is about
    a, b = res
    assert a[1] == b[1] > 10000
    np.testing.assert_allclose(a[0], b[0], rtol=1e-11)
    np.testing.assert_allclose(a[2:4], b[2:4], rtol=1e-7)
    ga, gb = a[6:].reshape(-1, 3, 4), b[6:].reshape(-1, 3, 4)
    for s_ in range(plan.n_scans):
        np.testing.assert_allclose(ga[s_], gb[s_], rtol=1e-6, atol=1e-7 * np.abs(gb[s_]).max())


def test_chained_wait_that_expires_is_reported_not_just_nan(room):
    """A chained launch whose blocks give up waiting for their weights (forced: zero polls) must not pass as a number and
    must not look like a q32 overflow: NaN sums AND bit 1 of the status word (SequencePlan.chain_timed_out)."""
    from depth_correction_amd import _native as nv
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import SequenceTrainer
    scans, poses = room
    plan, info = build_sequence([s[:50_000] for s in scans[:3]], poses[:3], k=10, dtype=torch.float32)
    tr = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3, chained=True)
    for _ in range(3):
        tr.step()
    ok = npy(tr.flush()).copy()
    assert np.isfinite(ok).all() and plan.status_bits() == 0
    nv.check(nv.lib().dc_set_option(5, 0), 'dc_set_option')
    try:
        tr.step()
        bad = npy(tr.flush()).copy()
    finally:
        nv.check(nv.lib().dc_set_option(5, -1), 'dc_set_option')
    torch.cuda.synchronize()
    assert np.isnan(bad[0]) and plan.chain_timed_out() and not plan.overflowed()
    assert plan.status_bits() == plan.STATUS_CHAIN_TIMEOUT
    plan.clear_status()
    # the sequence is usable again (fresh trainer: the poisoned one carries NaN weights)
    tr2 = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3, chained=True)
    tr2.step()
    again = npy(tr2.flush()).copy()
    assert np.isfinite(again).all() and not plan.chain_timed_out()


def test_chained_steps_beside_a_busy_stream_finish_with_the_same_bits(room):
    """The blocks of a chained launch wait (bounded) for the weights its leading blocks publish.  The leading blocks are dispatched
    first and need nothing from the waiting ones, so the launch completes whatever else occupies the CUs -- here a second stream
    of matrix products (as another rank's collective or any other kernel would): the steps finish beside it, nothing times out,
    and sums and weights are bit-identical to the quiet run (fixed-order sums: timing does not enter)."""
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import SequenceTrainer
    scans, poses = room
    plan, info = build_sequence([s[:50_000] for s in scans[:3]], poses[:3], k=10, dtype=torch.float32)

    def run(busy):
        tr = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3, chained=True)
        side = torch.cuda.Stream()
        a = torch.randn((4096, 4096), device='cuda:0')
        got = []
        if busy:
            with torch.cuda.stream(side):
                for _ in range(40):
                    a = torch.tanh(a @ a) * 0.5
        for _ in range(30):
            got.append(tr.step().clone())
            assert tr.chained
        got.append(tr.flush().clone())
        torch.cuda.synchronize()
        return [npy(v) for v in got], npy(tr.w).copy()

    quiet, wq = run(False)
    beside, wb = run(True)
    assert all(np.array_equal(x, y) for x, y in zip(quiet, beside)) and np.array_equal(wq, wb)
    assert np.isfinite(quiet[-1]).all() and plan.status_bits() == 0
