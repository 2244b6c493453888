"""Edge cases of the domain on the GPU: empty and tiny clouds, k larger than the cloud, ragged radius neighbourhoods,
neighbourhoods with one / zero valid members, duplicate points, non-finite coordinates, sizes around the block and
window boundaries, the widest k the builder supports."""
import ctypes

import numpy as np
import pytest
import torch

import dc_oracle as O
from helpers import t, npy, assert_eigvals_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cloud(n, seed=0, flat=0.02):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-2, 2, size=(n, 3)) * [1, 1, flat]).astype(np.float32).astype(np.float64)


def test_empty_inputs():
    from depth_correction_amd import ops
    x = torch.zeros((0, 3), dtype=torch.float64, device=DEV)
    d, i = ops.knn(x, 4)
    assert d.shape == (0, 4) and i.shape == (0, 4)
    nbr = torch.zeros((0, 4), dtype=torch.int32, device=DEV)
    f = ops.features_fwd(x, nbr, want=('mean', 'cov', 'eigvals'))
    assert f['eigvals'].shape == (0, 3)
    out = ops.consistency_fwd(x, nbr)
    assert npy(out['sums']).tolist() == [0.0, 0.0]
    p, s = ops.knn_transpose(nbr)
    assert npy(p).tolist() == [0]
    assert ops.radius_neighbors(x, 0.5).shape[0] == 0
    assert ops.spatial_order(x).numel() == 0


@pytest.mark.parametrize('n,k', [(1, 1), (1, 4), (3, 5), (7, 7), (40, 64)])
def test_k_larger_than_cloud(n, k):
    """cKDTree pads with index n (-> -1) and inf when fewer than k points exist (nearest_neighbors.py:48-49)."""
    from depth_correction_amd import ops
    pts = _cloud(n, seed=n)
    d, i = ops.knn(t(pts, DEV), k)
    dref, iref = O.knn_ckdtree(pts, k)
    if k == 1:
        dref, iref = dref.reshape(-1, 1), iref.reshape(-1, 1)          # cKDTree squeezes k = 1
    assert np.array_equal(npy(i), iref) and np.array_equal(npy(d), dref)


@pytest.mark.parametrize('n', [255, 256, 257, 1023, 1025, 5000])
def test_sizes_around_block_and_window_edges(n):
    from depth_correction_amd import ops
    pts = _cloud(n, seed=n)
    x = t(pts, DEV)
    _, idx = ops.knn(x, 6)
    assert np.array_equal(npy(idx), O.knn_ckdtree(pts, 6)[1])
    ref = O.closed_form_backward(pts, npy(idx).astype(np.int64), None, normalization=True, reduction='sum')
    for q32 in (False, True):
        if q32:
            qf = ops.QFormat.for_extent(pts.min(0), pts.max(0))
            xq = torch.round((x - torch.tensor(qf.origin, device=DEV)) / qf.scale).to(torch.int32)
            xs = torch.cat([xq, torch.zeros((n, 1), dtype=torch.int32, device=DEV)], 1).contiguous()
        else:
            qf, xs = None, torch.cat([x, torch.zeros((n, 1), dtype=x.dtype, device=DEV)], 1).contiguous()
        fw = ops.consistency_fwd(xs, idx, qfmt=qf, want_pointwise=True)
        cp, cs = ops.knn_transpose(idx)
        gp, _ = ops.consistency_bwd(xs, fw['rec'], cp, cs, want_grad_points=True, qfmt=qf)
        tol = 1e-9 if not q32 else 1e-4
        np.testing.assert_allclose(npy(fw['sums'])[0], ref['loss'], rtol=tol)
        np.testing.assert_allclose(npy(gp)[:, :3], ref['grad_points'], rtol=10 * tol, atol=tol * np.abs(ref['grad_points']).max())


def test_ragged_radius_neighbourhoods():
    """Radius search: variable neighbour counts, -1 padding, isolated points keep only themselves; features use
    weight 0 for the padding exactly like the reference (depth_cloud.py:213,291-295)."""
    from depth_correction_amd import ops
    pts = np.concatenate([_cloud(600, 1), np.array([[50.0, 50.0, 50.0], [-40.0, 0.0, 9.0]])])
    x = t(pts, DEV)
    idx = ops.radius_neighbors(x, 0.3)
    ref = O.radius_ckdtree(pts, 0.3)
    assert np.array_equal(npy(idx), ref)
    assert npy(idx)[-1].tolist()[:2] == [len(pts) - 1, -1] and (ref[-2] >= 0).sum() == 1
    f = ops.features_fwd(x, idx, want=('mean', 'cov', 'eigvals'), want_saved=True)
    fo = O.features(t(pts), t(ref), torch.zeros((len(pts), 3), dtype=torch.float64))
    many = (ref >= 0).sum(1) >= 3
    np.testing.assert_allclose(npy(f['mean']), npy(fo['mean']), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(npy(f['cov']), npy(fo['cov']), rtol=1e-9, atol=1e-16)
    assert_eigvals_close(npy(f['eigvals'])[many], npy(fo['eigvals'])[many], 1e-9)
    assert np.array_equal(npy(f['nvalid']), (ref >= 0).sum(1))
    # one valid neighbour: W - 1 = 0 is clamped to 1e-6, the covariance is exactly zero (utils.py:145-147)
    assert np.all(npy(f['cov'])[-1] == 0) and np.all(npy(f['eigvals'])[-1] == 0)


def test_neighbourhood_without_valid_members_is_nan_like_the_reference():
    from depth_correction_amd import ops
    pts = _cloud(50, 3)
    nbr = O.knn_ckdtree(pts, 4)[1]
    nbr[7] = -1                                                      # W = 0 -> 0/0 (depth_cloud.py:292-293)
    x = t(pts, DEV)
    f = ops.features_fwd(x, t(nbr, DEV).int(), want=('mean', 'eigvals'))
    valid = nbr.copy()
    valid[7] = 7                       # the reference itself cannot finish here: LAPACK rejects the NaN covariance
    fo = O.features(t(pts), t(valid), torch.zeros((50, 3), dtype=torch.float64))
    assert np.isnan(npy(f['mean'])[7]).all() and np.isnan(npy(f['eigvals'])[7]).all()
    ok = np.arange(50) != 7
    np.testing.assert_allclose(npy(f['mean'])[ok], npy(fo['mean'])[ok], rtol=1e-12, atol=1e-13)
    mask = torch.ones(50, dtype=torch.bool, device=DEV)
    mask[7] = False
    xs = torch.cat([x, torch.zeros((50, 1), dtype=x.dtype, device=DEV)], 1).contiguous()
    out = ops.consistency_fwd(xs, t(nbr, DEV).int(), mask=mask)
    assert np.isfinite(npy(out['sums'])).all() and npy(out['sums'])[1] == 49


def test_duplicate_points_are_deterministic_and_valid():
    """Exact ties are outside the cKDTree contract (tree-traversal order); here they resolve by index."""
    from depth_correction_amd import ops
    base = _cloud(300, 5)
    pts = np.concatenate([base, base[:100], base[:50]])
    x = t(pts, DEV)
    d1, i1 = ops.knn(x, 8)
    d2, i2 = ops.knn(x, 8)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)
    dref, _ = O.knn_ckdtree(pts, 8)
    assert np.array_equal(npy(d1), dref)                              # the distance multiset is still the exact k-NN
    dd = np.linalg.norm(pts[npy(i1)] - pts[:, None, :], axis=-1)
    np.testing.assert_allclose(dd, npy(d1), rtol=1e-14, atol=0)
    i = npy(i1)
    tie = npy(d1)[:, 1:] == npy(d1)[:, :-1]
    assert np.all(i[:, 1:][tie] > i[:, :-1][tie])                     # ties ordered by ascending index


def test_non_finite_points_do_not_poison_the_rest():
    from depth_correction_amd import ops
    pts = _cloud(400, 9)
    pts[13] = np.nan
    pts[77, 1] = np.inf
    x = t(pts, DEV)
    d, i = ops.knn(x, 5)
    good = np.isfinite(pts).all(1)
    sub = pts[good]
    remap = np.flatnonzero(good)
    dref, iref = O.knn_ckdtree(sub, 5)
    rows = npy(i)[good]
    assert np.array_equal(rows, remap[iref]) and np.array_equal(npy(d)[good], dref)


@pytest.mark.parametrize('k', [2, 16, 17, 33, 64])
def test_wide_and_narrow_k(k):
    from depth_correction_amd import ops
    pts = _cloud(3000, k)
    d, i = ops.knn(t(pts, DEV), k)
    dref, iref = O.knn_ckdtree(pts, k)
    assert np.array_equal(npy(i), iref) and np.array_equal(npy(d), dref)
    x4 = torch.cat([t(pts, DEV), torch.zeros((3000, 1), dtype=torch.float64, device=DEV)], 1).contiguous()
    kind = 'trace_loss' if k == 2 else 'min_eigval_loss'        # two points: lambda0 is pure round-off
    fw = ops.consistency_fwd(x4, i, loss=kind, normalization=False)
    ref = O.closed_form_backward(pts, iref, None, kind=kind, normalization=False, reduction='sum')
    np.testing.assert_allclose(npy(fw['sums'])[0], ref['loss'], rtol=1e-9)
    with pytest.raises(ValueError):
        ops.knn(t(pts, DEV), 65)


def test_operand_checks_refuse_bad_shapes():
    from depth_correction_amd import ops
    x = t(_cloud(10), DEV)
    with pytest.raises(ValueError):
        ops.features_fwd(x, torch.zeros((9, 3), dtype=torch.int32, device=DEV))
    with pytest.raises(TypeError):
        ops.features_fwd(x, torch.zeros((10, 3), dtype=torch.int64, device=DEV))
    with pytest.raises(RuntimeError):
        ops.knn(x.cpu(), 3)
    with pytest.raises(RuntimeError):
        ops.knn(x.t().contiguous().t(), 3)                             # non-contiguous view


def _staged_vs_gather(x4, nbr, qf=None, mask=None):
    """Forward + backward through block tables (LDS-staged) and without: everything must agree bit for bit."""
    from depth_correction_amd import ops
    cp, cs = ops.knn_transpose(nbr, n_dst=x4.shape[0])
    ft, bt = ops.block_table(nbr=nbr), ops.block_table(csr=(cp, cs))
    assert ft is not None and bt is not None
    outs = []
    for tabs in ((None, None), (ft, bt)):
        fw = ops.consistency_fwd(x4, nbr, mask=mask, want_pointwise=True, want_eigvals=True, qfmt=qf, table=tabs[0])
        gp, _ = ops.consistency_bwd(x4, fw['rec'], cp, cs, want_grad_points=True, qfmt=qf, table=tabs[1])
        outs.append([npy(fw[f]) for f in ('sums', 'pointwise', 'eigvals', 'rec')] + [npy(gp)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
    return outs[1]


@pytest.mark.parametrize('n,k', [(1, 1), (2, 2), (255, 6), (256, 6), (257, 6), (1025, 10), (3000, 16), (3000, 17),
                                 (3000, 33), (2000, 64)])
def test_block_tables_sizes_and_wide_k(n, k):
    """Block boundaries, a single point, and neighbour lists longer than the 16 positions the staged kernels keep in
    registers (k = 17, 33, 64; in-degrees above 16 in the backward)."""
    from depth_correction_amd import ops
    pts = _cloud(n, seed=n + k)
    x = t(pts, DEV)
    _, idx = ops.knn(x, k)
    for q32 in (False, True):
        if q32:
            qf = ops.QFormat.for_extent(pts.min(0), pts.max(0))
            xq = torch.round((x - torch.tensor(qf.origin, device=DEV)) / qf.scale).to(torch.int32)
            xs = torch.cat([xq, torch.zeros((n, 1), dtype=torch.int32, device=DEV)], 1).contiguous()
        else:
            qf, xs = None, torch.cat([x, torch.zeros((n, 1), dtype=x.dtype, device=DEV)], 1).contiguous()
        _staged_vs_gather(xs, idx, qf)


def test_block_tables_ragged_lists_and_hubs():
    """Radius neighbourhoods (empty slots, isolated points, rows without any valid member) and a hub point that sits in
    hundreds of neighbourhoods (an in-degree far above the block's other points)."""
    from depth_correction_amd import ops
    rng = np.random.default_rng(5)
    pts = np.concatenate([_cloud(1500, 2), np.array([[50.0, 50.0, 50.0], [-40.0, 0.0, 9.0]])])
    x = t(pts, DEV)
    idx = ops.radius_neighbors(x, 0.25)
    idx[7] = -1                                                      # a centre without any valid neighbour
    hub = rng.choice(len(pts), 400, replace=False)
    idx[hub, idx.shape[1] - 1] = 3                                   # point 3 gains ~400 incoming edges
    xs = torch.cat([x, torch.zeros((len(pts), 1), dtype=x.dtype, device=DEV)], 1).contiguous()
    mask = t(rng.random(len(pts)) < 0.7, DEV)
    out = _staged_vs_gather(xs, idx.contiguous(), mask=mask)
    assert np.isfinite(out[4][3]).all() and np.abs(out[4][3]).max() > 0


def test_block_table_of_empty_cloud():
    from depth_correction_amd import ops
    nbr = torch.zeros((0, 4), dtype=torch.int32, device=DEV)
    tab = ops.block_table(nbr=nbr)
    assert tab.max_rows == 0 and npy(tab.blk_ptr).tolist() == [0] and npy(tab.slot_ptr).tolist() == [0]
    x = torch.zeros((0, 4), dtype=torch.float64, device=DEV)
    assert npy(ops.consistency_fwd(x, nbr, table=tab)['sums']).tolist() == [0.0, 0.0]


@pytest.mark.timeout(120)
def test_knn_with_far_outliers_is_exact_and_does_not_stall():
    """A few points far from the bulk (range outliers): the grid is sized for the bulk (bounding box cut to
    mean +- 6 sigma) and a query that would need thousands of empty shells switches to a scan of all points.
    Results stay bit-exact, for self queries and for cross-cloud queries far outside the cloud."""
    from depth_correction_amd import ops
    rng = np.random.default_rng(11)
    bulk = _cloud(20000, 3)
    far = np.array([[1.0e3, 0.0, 0.0], [-2.5e4, 3.0e4, 10.0], [5.0, -7.0e5, 2.0e5], [1.0e3, 0.01, 0.0]])
    pts = np.concatenate([bulk, far])[rng.permutation(len(bulk) + len(far))]
    x = t(pts, DEV)
    d, i = ops.knn(x, 10)
    dref, iref = O.knn_ckdtree(pts, 10)
    assert np.array_equal(npy(i), iref) and np.array_equal(npy(d), dref)
    q = np.array([[4.0e4, 1.0, -3.0], [0.0, 0.0, 9.0e6], [0.1, 0.2, 0.0]])
    dq, iq = ops.knn(x, 3, query=t(q, DEV))
    from scipy.spatial import cKDTree
    dr, ir = cKDTree(pts).query(q, 3)
    assert np.array_equal(npy(iq), ir) and np.array_equal(npy(dq), dr)


def _local_scans(sizes, dtype, seed):
    """Synthetic sensor-frame scans of a slightly noisy floor patch (dirs, depth, incidence angle, local mask) + poses."""
    rng = np.random.default_rng(seed)
    clouds, poses = [], []
    for s, n in enumerate(sizes):
        xy = rng.uniform(-1.5, 1.5, size=(n, 2))
        p = np.concatenate([xy, -1.2 + 0.01 * rng.normal(size=(n, 1))], axis=1)
        depth = np.linalg.norm(p, axis=1, keepdims=True)
        dirs = p / depth
        inc = np.arccos(np.abs(dirs[:, 2:3]))
        T = np.eye(4)
        T[:3, 3] = [0.2 * s, -0.1 * s, 0.01 * s]
        clouds.append(dict(vps=torch.zeros((n, 3), dtype=dtype, device=DEV), dirs=t(dirs, DEV, dtype), depth=t(depth, DEV, dtype),
                           inc_angles=t(inc, DEV, dtype), mask=t(rng.random(n) < 0.8, DEV)))
        poses.append(T)
    return clouds, t(np.stack(poses), DEV)


@pytest.mark.parametrize('sizes,k', [((1, 1), 2), ((100, 155), 4), ((128, 128), 10), ((200, 57), 10), ((300, 213), 6),
                                     ((700, 325), 16), ((40, 900, 84), 12), ((600, 500), 33)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_sequence_plan_small_and_odd_sizes(sizes, k, dtype):
    """Whole-sequence evaluations (basis form and general path) on sequences of 2-1025 points: partial blocks, exactly one
    block, scans of very different sizes, every supported way of choosing the forward kernel (k = 4 / 10 / 16 compiled,
    2 / 6 / 12 / 33 run-time slots -- 33 beyond the 16 positions kept in registers); a plan whose global mask is empty yields count 0, loss 0 and zero gradients."""
    from depth_correction_amd import ops, _native as nv
    from depth_correction_amd.plan import SequencePlan
    clouds, poses = _local_scans(sizes, dtype, seed=sum(sizes) + k)
    n = sum(sizes)
    x = torch.cat([(c['dirs'] * c['depth']).double() @ T[:3, :3].T + T[:3, 3] for c, T in zip(clouds, poses)])
    _, nbr = ops.knn(x, k)
    mask = t(np.random.default_rng(1).random(n) < 0.7, DEV)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=DEV)
    w = torch.tensor([2e-3, -1e-3], dtype=torch.float64, device=DEV)
    for m in (mask, torch.zeros_like(mask)):
        plan = SequencePlan(clouds, poses, nbr, m)
        outs = {}
        for general in (0, 1):
            nv.check(nv.lib().dc_set_option(3, general), 'dc_set_option')
            try:
                out = torch.full((2 + 4 + 12 * plan.n_scans,), 3.0, dtype=torch.float64, device=DEV)
                outs[general] = npy(plan.eval_native(w, e, plan.poses12(poses), out))
            finally:
                nv.check(nv.lib().dc_set_option(3, 0), 'dc_set_option')
        a, b = outs[0], outs[1]
        assert a[1] == b[1] == float(m.sum()) and np.all(a[4:] == 0) and np.all(b[4:] == 0)
        if not bool(m.any()):
            assert np.all(a[:4] == 0) and np.all(b[:4] == 0)
            continue
        f64 = dtype == torch.float64
        np.testing.assert_allclose(a[0], b[0], rtol=1e-11 if f64 else 1e-5)
        # (fp64 clouds: the loss to 1e-11; the gradient to 3e-6 -- the basis form's second sweep reads float32 copies of u and c,
        #  6e-8 on each term of a sum of both signs over a few hundred points that average nothing away: 8.7e-7 measured at
        #  sizes (128, 128), K = 10; the full-size cases hold 1e-7)
        np.testing.assert_allclose(a[2:4], b[2:4], rtol=3e-6 if f64 else 1e-3, atol=(1e-8 if f64 else 1e-5) * np.abs(b[2:4]).max())


@pytest.mark.timeout(180)
def test_knn_row_per_query_kernel_corner_cases():
    """knn_group_kernel (16 lanes per query, the default for k <= 16) against cKDTree and against the two lane-per-query builds
    (dc_knn_set_shell_budget(102): with the tail kernel, (-1): to the end) where its special paths run:
      * points repeated 40 times -- more than 16 finalists share the k-th rounded distance: the exact row-wide selection;
      * a regular lattice -- many EQUAL distances: finalists ranked by (fp64 distance, index), cKDTree's tie order by index;
      * k = 1, 4, 10, 16, a radius limit that leaves rows short, a cloud smaller than k, one dense cell with 600 points
        (pool refilled many times), NaN rows."""
    from depth_correction_amd import ops, _native as nv
    rng = np.random.default_rng(3)

    def three_ways(x, k, r=None, query=None):
        out = []
        for budget in (2, 102, -1):
            nv.check(nv.lib().dc_knn_set_shell_budget(budget), 'budget')
            try:
                d, i = ops.knn(x, k, r=r, query=query)
            finally:
                nv.check(nv.lib().dc_knn_set_shell_budget(1000), 'budget')
            out.append((npy(d), npy(i)))
        for d, i in out[1:]:
            assert np.array_equal(out[0][1], i) and np.array_equal(out[0][0], d)
        # without a distance table knn_group_kernel has nowhere to hand its best lists over: knn_tail_kernel starts the pending queries
        # from nothing -- the same table
        _, i_nd = ops.knn(x, k, r=r, query=query, want_dist=False)
        assert np.array_equal(out[0][1], npy(i_nd))
        return out[0]

    # repeated points: every query has 40 candidates at distance 0 and 40 more at each next distance
    base = rng.uniform(-1, 1, size=(300, 3)) * [1, 1, 0.02]
    rep = np.repeat(base, 40, axis=0)[rng.permutation(300 * 40)]
    d, i = three_ways(t(rep, DEV), 10)
    dref, iref = O.knn_ckdtree(rep, 10)
    assert np.array_equal(d, dref)                                       # distances are unique as a multiset
    assert (d == 0).all()                                                # ten of the 40 copies
    same = (rep[i] == rep[:, None, :]).all(axis=2)
    assert same.all()
    assert (np.diff(i, axis=1) > 0).all()                                # equal distances: ascending index, the builders' order
    # lattice: equal distances everywhere
    ax = np.arange(24, dtype=np.float64) * 0.25
    lat = np.stack(np.meshgrid(ax, ax, ax[:6], indexing='ij'), axis=-1).reshape(-1, 3)[rng.permutation(24 * 24 * 6)]
    for k in (1, 4, 10, 16):
        d, i = three_ways(t(lat, DEV), k)
        dref, _ = O.knn_ckdtree(lat, k)
        assert np.array_equal(d, dref.reshape(d.shape))
        dd = np.linalg.norm(lat[i] - lat[:, None, :], axis=2)
        assert np.allclose(dd, d, rtol=0, atol=1e-12)
        tie = np.diff(d, axis=1) == 0
        assert (np.diff(i, axis=1)[tie] > 0).all()
    # radius limit + short rows, float32 input
    pts = (rng.uniform(-2, 2, size=(6000, 3)) * [1, 1, 0.05]).astype(np.float32)
    d, i = three_ways(t(pts, DEV), 8, r=0.08)
    dref, iref = O.knn_ckdtree(pts.astype(np.float64), 8, r=0.08)
    assert np.array_equal(i, iref) and np.array_equal(d, dref) and (i == -1).any()
    # fewer points than k; cross-cloud queries
    tiny = rng.uniform(-1, 1, size=(7, 3))
    d, i = three_ways(t(tiny, DEV), 10)
    assert (i[:, 7:] == -1).all() and np.isinf(d[:, 7:]).all() and (np.sort(i[:, :7], axis=1) == np.arange(7)).all()
    d, i = three_ways(t(pts.astype(np.float64), DEV), 16, query=t(rng.uniform(-2.5, 2.5, size=(999, 3)) * [1, 1, 0.1], DEV))
    # one very dense cell: the pool runs full over and over
    dense = np.concatenate([rng.normal(scale=1e-3, size=(600, 3)), rng.uniform(-1, 1, size=(4000, 3)) * [1, 1, 0.02]])
    d, i = three_ways(t(dense, DEV), 10)
    dref, iref = O.knn_ckdtree(dense, 10)
    assert np.array_equal(i, iref) and np.array_equal(d, dref)
    # the grid level a query searches at (fine where its own coarse cell is crowded) never changes the answer
    mixed = np.concatenate([rng.normal(scale=0.02, size=(3000, 3)) + [0.5, 0.5, 0.0], rng.uniform(-2, 2, size=(5000, 3)) * [1, 1, 0.03]])
    want = None
    for fine_min in (0, 1, 4, 16, 64):
        nv.check(nv.lib().dc_knn_set_fine_cell_count(fine_min), 'fine')
        try:
            d, i = ops.knn(t(mixed, DEV), 10)
            dq, iq = ops.knn(t(mixed, DEV), 7, query=t(mixed[::3] + 1e-3, DEV))
        finally:
            nv.check(nv.lib().dc_knn_set_fine_cell_count(14), 'fine')
        got = (npy(d), npy(i), npy(dq), npy(iq))
        if want is None:
            want = got
            dref, iref = O.knn_ckdtree(mixed, 10)
            assert np.array_equal(got[1], iref) and np.array_equal(got[0], dref)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    # NaN rows neither find nor are found
    bad = pts.astype(np.float64).copy()
    bad[::97] = np.nan
    d, i = three_ways(t(bad, DEV), 5)
    assert (i[::97] == -1).all() and not np.isin(i, np.arange(0, len(bad), 97)).any()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('shape', ['volume', 'plane', 'line', 'clusters', 'offset', 'tiny', 'identical'])
def test_knn_row_per_query_kernel_shapes_and_sizes(shape):
    """knn_group_kernel over cloud shapes (volumetric, planar, a line, tight clusters with empty space between them: every
    stage-count and tail hand-over) x sizes that are not multiples of 16 x k = 2..15: indices and distances equal cKDTree's and
    the lane-per-query build's."""
    from depth_correction_amd import ops, _native as nv
    rng = np.random.default_rng({'volume': 1, 'plane': 2, 'line': 3, 'clusters': 4, 'offset': 5, 'tiny': 6, 'identical': 7}[shape])
    for n in (1, 15, 17, 1000, 30011):
        if shape == 'volume':
            pts = rng.uniform(-1, 1, size=(n, 3))
        elif shape == 'plane':
            pts = rng.uniform(-5, 5, size=(n, 3)) * [1, 1, 1e-3]
        elif shape == 'line':
            pts = np.outer(rng.uniform(0, 100, size=n), [1.0, 0.5, 0.25]) + 1e-4 * rng.normal(size=(n, 3))
        elif shape == 'clusters':
            centres = rng.uniform(-50, 50, size=(7, 3))
            pts = centres[rng.integers(0, 7, size=n)] + 0.01 * rng.normal(size=(n, 3))
        elif shape == 'offset':                      # a cloud far from the origin: absolute coordinates ~1e6, extent a few metres
            pts = rng.uniform(-3, 3, size=(n, 3)) * [1, 1, 0.05] + [1.0e6, -2.0e6, 5.0e5]
        elif shape == 'tiny':                        # extent 1e-7: every distance is a handful of ulps of the coordinates
            pts = 1.0 + 1e-7 * rng.uniform(size=(n, 3))
        else:                                        # one point repeated (one cell holds everything)
            if n > 1000:
                continue
            pts = np.tile(np.array([[0.25, -1.5, 3.0]]), (n, 1))
        x = t(pts, DEV)
        for k in (2, 5, 9, 15):
            d, i = ops.knn(x, k)
            dref, iref = O.knn_ckdtree(pts, k)
            assert np.array_equal(npy(d), dref.reshape(npy(d).shape)), (shape, n, k)
            if shape == 'identical':                 # equal distances: cKDTree's order is its traversal's, the builders' is by index
                ii = npy(i)
                assert (ii[:, :min(k, n)] == np.arange(min(k, n))).all() and (ii[:, n:] == -1).all()
            else:
                assert np.array_equal(npy(i), iref.reshape(npy(i).shape)), (shape, n, k)
        nv.check(nv.lib().dc_knn_set_shell_budget(102), 'budget')
        try:
            d2, i2 = ops.knn(x, 9)
        finally:
            nv.check(nv.lib().dc_knn_set_shell_budget(1000), 'budget')
        d1, i1 = ops.knn(x, 9)
        assert torch.equal(i1, i2) and torch.equal(d1, d2)


def test_round4_entry_points_reject_bad_arguments():
    """Status codes of the C ABI (include/dc_hip.h: 0 ok, < 0 invalid argument, nothing launched): the entry points added in round 4
    refuse what they cannot serve instead of faulting -- layouts, counts and sizes outside their ranges, missing arrays, in-place
    calls that would race."""
    from depth_correction_amd import _native as nv
    lib, ptr = nv.lib(), nv.ptr
    dev = torch.device(DEV)
    f64 = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
    i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
    S, P = 3, 2
    sums, w, m, v = f64(2 + 2 * P + 12 * S), f64(P), f64(P), f64(P)
    T0, d, dm, dv, T, P12 = f64(S, 16), f64(S, 6), f64(S, 6), f64(S, 6), f64(S, 16), f64(S, 12)
    step = torch.zeros((), dtype=torch.int64, device=dev)
    args = lambda **kw: [kw.get('sums', ptr(sums)), kw.get('layout', 0), kw.get('nt', P), kw.get('ns', S), ptr(w), ptr(m), ptr(v), ptr(T0),
                         ptr(d), ptr(dm), ptr(dv), kw.get('nd', S), 1, ptr(step), 1e-3, 1e-3, kw.get('b1', 0.9), 0.999, 1e-8, ptr(T),
                         kw.get('rec', None), kw.get('rows', 0), ptr(T), ptr(P12), None, kw.get('extra', None), kw.get('n_extra', 0),
                         nv.stream_ptr()]
    assert lib.dc_pose_train_finish(*args()) == 0
    for bad in (dict(layout=2), dict(sums=None), dict(nt=0), dict(nt=99), dict(ns=0), dict(nd=2), dict(b1=1.0),
                dict(rec=ptr(f64(4, 200)), rows=0), dict(n_extra=4), dict(extra=ptr(f64(4)), n_extra=-1)):
        assert lib.dc_pose_train_finish(*args(**bad)) == nv.DC_ERR_ARG, bad
    outs = (ctypes.c_void_p * 17)(*([sums.data_ptr()] * 17))
    tot = f64(2 + P)
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    assert lib.dc_pose_train_combine(cast(outs), 2, 0, P, ptr(tot), nv.stream_ptr()) == 0
    for n_seq, layout, nt in ((0, 0, P), (17, 0, P), (2, 3, P), (2, 0, 0)):
        assert lib.dc_pose_train_combine(cast(outs), n_seq, layout, nt, ptr(tot), nv.stream_ptr()) == nv.DC_ERR_ARG
    # the two-group form (training and validation sequences of an iteration): either group may be empty
    tot2 = f64(2, 2 + P)
    sums[:4] = torch.tensor([3.0, 2.0, 0.5, -0.25], dtype=torch.float64, device=dev)
    assert lib.dc_pose_train_combine2(cast(outs), 2, cast(outs), 0, 0, P, ptr(tot2), nv.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert tot2.cpu().tolist() == [[6.0, 4.0, 1.0, -0.5], [0.0, 0.0, 0.0, 0.0]]
    for n_a, n_b in ((17, 0), (2, 17), (-1, 0)):
        assert lib.dc_pose_train_combine2(cast(outs), n_a, cast(outs), n_b, 0, P, ptr(tot2), nv.stream_ptr()) == nv.DC_ERR_ARG
    pts = torch.zeros((10, 3), dtype=torch.float32, device=dev)
    sp = torch.tensor([0, 4, 10], dtype=torch.int64, device=dev)
    xs, info = f64(10, 3), i32(1)
    nb = lib.dc_scan_lattice_workspace_bytes(2)
    ws = torch.zeros((nb,), dtype=torch.uint8, device=dev)
    assert lib.dc_scan_lattice_shift(ptr(pts), nv.DC_F32, 10, ptr(sp), 2, ptr(xs), ptr(info), ptr(ws), nb, nv.stream_ptr()) == 0
    assert lib.dc_scan_lattice_shift(ptr(pts), nv.DC_F32, 10, ptr(sp), 65, ptr(xs), ptr(info), ptr(ws), nb, nv.stream_ptr()) == nv.DC_ERR_ARG
    assert lib.dc_scan_lattice_shift(ptr(pts), nv.DC_F32, 10, ptr(sp), 2, ptr(xs), ptr(info), ptr(ws), 8, nv.stream_ptr()) == nv.DC_ERR_WORKSPACE
    assert lib.dc_scan_lattice_shift(ptr(pts), 7, 10, ptr(sp), 2, ptr(xs), ptr(info), ptr(ws), nb, nv.stream_ptr()) == nv.DC_ERR_DTYPE
    nbr = i32(10, 4)
    assert lib.dc_scan_lattice_localize(ptr(nbr), 10, 0, ptr(sp), 2, ptr(info), nv.stream_ptr()) == nv.DC_ERR_ARG
    out6 = f64(6)
    nb6 = lib.dc_points_extent_workspace_bytes()
    ws6 = torch.zeros((nb6,), dtype=torch.uint8, device=dev)
    assert lib.dc_points_extent(ptr(pts), 5, nv.DC_F32, 10, ptr(out6), ptr(ws6), nb6, nv.stream_ptr()) == nv.DC_ERR_ARG
    assert lib.dc_points_extent(ptr(pts), 3, nv.DC_F32, 10, ptr(out6), ptr(ws6), 16, nv.stream_ptr()) == nv.DC_ERR_WORKSPACE
    order = torch.arange(10, dtype=torch.int64, device=dev)
    rank = i32(10)
    assert lib.dc_table_permute(ptr(nbr), 10, 4, ptr(order), ptr(rank), ptr(nbr), nv.stream_ptr()) == nv.DC_ERR_ARG      # in place
    assert lib.dc_scan_ids(None, 2, 10, ptr(rank), nv.stream_ptr()) == nv.DC_ERR_ARG
    torch.cuda.synchronize()
