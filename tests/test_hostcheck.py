"""The per-point math of the kernels (csrc/dc_eig3.h, dc_pointmath.h) compiled for the host (libdc_hostcheck.so, a
TEST-ONLY build) against LAPACK and the oracle.  No GPU needed; pins the arithmetic before it goes on the device."""
import ctypes
import os

import numpy as np
import pytest
import torch

import dc_oracle as O
from helpers import t, npy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'depth_correction_amd', 'lib', 'libdc_hostcheck.so')


@pytest.fixture(scope='module')
def host():
    if not os.path.exists(LIB):
        import __graft_entry__ as ge
        ge.build()
    return ctypes.CDLL(LIB)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _eig(host, C, solver='dc_host_eig3'):
    c6 = np.ascontiguousarray(np.stack([C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2]], 1))
    lam, vec = np.zeros((len(C), 3)), np.zeros((len(C), 9))
    getattr(host, solver)(_p(c6), ctypes.c_long(len(C)), _p(lam), _p(vec))
    return lam, vec.reshape(-1, 3, 3)


def _spd(rng, lams):
    Q, _ = np.linalg.qr(rng.normal(size=(len(lams), 3, 3)))
    C = np.einsum('nij,nj,nkj->nik', Q, lams, Q)
    return 0.5 * (C + C.transpose(0, 2, 1))


@pytest.mark.parametrize('solver', ['dc_host_eig3', 'dc_host_eig3_v2'])
@pytest.mark.parametrize('case', ['generic', 'planar', 'needle', 'double_lo', 'double_hi', 'isotropic', 'near_isotropic', 'edge',
                                  'sign_switch', 'tiny', 'huge'])
def test_eig3_matches_lapack(host, case, solver):
    """Every eigenvalue within LAPACK's own absolute accuracy (a few eps * |C|), for clustered small eigenvalues too;
    the reference's known-answer test asks for 1e-6 / 1e-5 (loss.py:731-735)."""
    rng = np.random.default_rng(0)
    n = 5000
    u = rng.uniform
    lams = {'generic': u(0, 1, (n, 3)),
            'planar': np.stack([10 ** u(-10, -3, n), u(0.3, 1, n), u(0.3, 1, n)], 1),
            'needle': np.stack([10 ** u(-10, -4, n), 10 ** u(-10, -4, n), u(0.3, 1, n)], 1),
            'double_lo': np.stack([np.full(n, 0.2), np.full(n, 0.2), u(0.3, 1, n)], 1),
            'double_hi': np.stack([u(0.01, 0.2, n), np.full(n, 0.5), np.full(n, 0.5)], 1),
            'isotropic': np.full((n, 3), 0.37),
            # anisotropy from round-off level up to 1e-6 of the scale; spectra with det(B) on either side of zero (where the
            # solvers switch the eigenvalue they isolate); edge-like spectra
            'near_isotropic': 0.37 * (1.0 + 10 ** u(-16, -6, (n, 1)) * u(-1, 1, (n, 3))),
            'edge': np.stack([10 ** u(-8, -3, n), 10 ** u(-3, -0.5, n), u(0.3, 1, n)], 1),
            'sign_switch': np.stack([0.5 - u(0.1, 0.4, n), 0.5 + u(-1e-7, 1e-7, n), 0.5 + u(0.1, 0.4, n)], 1),
            'tiny': u(0, 1, (n, 3)) * 1e-14, 'huge': u(0, 1, (n, 3)) * 1e12}[case]
    if case == 'sign_switch':
        lams[:, 2] = 1.0 - lams[:, 0]                                  # symmetric about the middle one: det(B) ~ 0
    lams = np.sort(lams, axis=1)
    C = _spd(rng, lams)
    lam, V = _eig(host, C, solver)
    ref = np.linalg.eigh(C)[0]
    scale = np.abs(ref).max(1, keepdims=True)
    assert np.all(np.diff(lam, axis=1) >= 0)
    assert np.abs(lam - ref).max() <= 1e-14 * scale.max() and (np.abs(lam - ref) / scale).max() < 5e-15
    resid = np.linalg.norm(np.einsum('nij,nkj->nki', C, V) - lam[:, :, None] * V, axis=2) / scale
    assert resid.max() < 1e-14
    assert np.abs(np.einsum('nki,nli->nkl', V, V) - np.eye(3)).max() < 1e-14


def _eig_smallest(host, C, solver='dc_host_eig3_smallest'):
    c6 = np.ascontiguousarray(np.stack([C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2]], 1))
    lam0, v0, tr = np.zeros(len(C)), np.zeros((len(C), 3)), np.zeros(len(C))
    getattr(host, solver)(_p(c6), ctypes.c_long(len(C)), _p(lam0), _p(v0), _p(tr))
    return lam0, v0, tr


@pytest.mark.parametrize('solver', ['dc_host_eig3_smallest', 'dc_host_eig3_smallest_r2', 'dc_host_eig3_smallest_v2'])
@pytest.mark.parametrize('case', ['generic', 'planar', 'needle', 'edge', 'threshold', 'threshold_unit', 'double_hi', 'tiny', 'huge'])
def test_eig3_smallest_matches_lapack(host, case, solver):
    """The hot-path solver (smallest eigenpair + trace only, dc_eig3.h eig3_smallest): eigenvalue within a few
    eps * |C| of LAPACK's for every family, including spectra on either side of the switch between the direct path and
    the isolate-largest-then-deflate path (kDeflateHalf); eigenvector residual small wherever lam0 is separated."""
    rng = np.random.default_rng(1)
    n = 20000
    u = rng.uniform
    if case in ('threshold', 'threshold_unit'):
        # scaled spectra 2 cos(ang + 2 pi k / 3) with cos(3 ang) swept across the switch at 0.9 (eig3_smallest: deflation;
        # eig3_smallest_unit: second Newton step) and across 0.999 (eig3_smallest_unit: deflation)
        ang = np.arccos(u(0.85, 0.95, n) if case == 'threshold' else u(0.99, 0.99999, n)) / 3
        beta = np.stack([2 * np.cos(ang + 2 * np.pi / 3), 2 * np.cos(ang - 2 * np.pi / 3), 2 * np.cos(ang)], 1)
        lams = 1.0 + 0.3 * beta
    else:
        lams = {'generic': u(0, 1, (n, 3)),
                'planar': np.stack([10 ** u(-10, -3, n), u(0.3, 1, n), u(0.3, 1, n)], 1),
                'needle': np.stack([10 ** u(-10, -4, n), 10 ** u(-10, -4, n), u(0.3, 1, n)], 1),
                'edge': np.stack([10 ** u(-8, -3, n), 10 ** u(-3, -0.5, n), u(0.3, 1, n)], 1),
                'double_hi': np.stack([u(0.01, 0.2, n), np.full(n, 0.5), np.full(n, 0.5)], 1),
                'tiny': u(0, 1, (n, 3)) * 1e-14, 'huge': u(0, 1, (n, 3)) * 1e12}[case]
    lams = np.sort(lams, axis=1)
    C = _spd(rng, lams)
    lam0, v0, tr = _eig_smallest(host, C, solver)
    ref, refV = np.linalg.eigh(C)
    scale = np.abs(ref).max(1)
    assert (np.abs(lam0 - ref[:, 0]) / scale).max() < 1e-14
    np.testing.assert_allclose(tr, np.trace(C, axis1=1, axis2=2), rtol=1e-14)
    assert np.abs(np.linalg.norm(v0, axis=1) - 1).max() < 1e-14
    sep = (ref[:, 1] - ref[:, 0]) / scale > 1e-3                       # eigenvector defined (not a cluster)
    resid = np.linalg.norm(np.einsum('nij,nj->ni', C, v0) - lam0[:, None] * v0, axis=1) / scale
    align = np.abs(np.einsum('ni,ni->n', v0, refV[:, :, 0]))
    if sep.any():
        gap = (ref[sep, 1] - ref[sep, 0]) / scale[sep]
        assert (resid[sep] * gap).max() < 1e-13
        assert ((1 - align[sep]) * gap ** 2).max() < 1e-13


@pytest.mark.parametrize('solver', ['dc_host_eig3_smallest', 'dc_host_eig3_smallest_r2', 'dc_host_eig3_smallest_v2'])
def test_eig3_smallest_degenerate_inputs(host, solver):
    lam0, v0, tr = _eig_smallest(host, np.zeros((2, 3, 3)), solver)
    assert np.all(lam0 == 0) and np.all(tr == 0) and np.allclose(v0, [[1, 0, 0]] * 2)
    lam0, _, _ = _eig_smallest(host, np.full((1, 3, 3), np.nan), solver)
    assert np.all(np.isnan(lam0))
    lam0, v0, tr = _eig_smallest(host, np.diag([3.0, 1.0, 2.0])[None], solver)
    assert abs(lam0[0] - 1.0) < 1e-15 and abs(tr[0] - 6.0) < 1e-15 and np.allclose(np.abs(v0), [[0, 1, 0]])
    lam0, v0, tr = _eig_smallest(host, 0.37 * np.eye(3)[None], solver)          # isotropic: any unit vector
    assert abs(lam0[0] - 0.37) < 1e-15 and abs(np.linalg.norm(v0) - 1) < 1e-15


@pytest.mark.parametrize('solver', ['dc_host_eig3', 'dc_host_eig3_v2'])
def test_eig3_degenerate_inputs(host, solver):
    lam, V = _eig(host, np.zeros((2, 3, 3)), solver)
    assert np.all(lam == 0) and np.allclose(V, np.eye(3))
    lam, _ = _eig(host, np.full((1, 3, 3), np.nan), solver)
    assert np.all(np.isnan(lam))
    lam, V = _eig(host, np.diag([3.0, 1.0, 2.0])[None], solver)
    assert np.allclose(lam, [[1.0, 2.0, 3.0]])
    assert np.allclose(np.abs(V[0]), [[0, 1, 0], [0, 0, 1], [1, 0, 0]])
    lam, V = _eig(host, np.diag([0.0, 0.0, 2.0])[None], solver)         # rank one
    assert np.allclose(lam, [[0.0, 0.0, 2.0]]) and np.allclose(np.abs(V[0, 2]), [0, 0, 1])


@pytest.mark.parametrize('kind,norm,sqrt', [(0, 1, 0), (0, 0, 0), (0, 1, 1), (1, 0, 0), (1, 0, 1)])
def test_neighbourhood_math_vs_oracle(host, golden, kind, norm, sqrt):
    """mean / covariance / eigenvalues / pointwise loss / backward coefficients of the kernels' inline functions on the
    golden room cloud: forward vs the oracle's features, backward vs the closed form."""
    g = golden('room_k10')
    x = np.ascontiguousarray(g['g_points'])
    nbr = np.ascontiguousarray(g['g_neighbors'].astype(np.int32))
    n, k = nbr.shape
    out = {f: np.zeros((n, d)) for f, d in dict(mean=3, cov6=6, lam=3, v0=3).items()}
    loss, c1, c2 = np.zeros(n), np.zeros(n), np.zeros(n)
    host.dc_host_neighbourhoods(_p(x), _p(nbr), ctypes.c_long(n), k, ctypes.c_double(0.0), kind, norm, sqrt, _p(out['mean']),
                                _p(out['cov6']), _p(out['lam']), _p(out['v0']), _p(loss), _p(c1), _p(c2))
    np.testing.assert_allclose(out['mean'], g['g_mean'], rtol=1e-12, atol=1e-13)
    C = g['g_cov']
    ref6 = np.stack([C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2]], 1)
    np.testing.assert_allclose(out['cov6'], ref6, rtol=1e-9, atol=1e-16)
    np.testing.assert_allclose(out['lam'], g['g_eigvals'], rtol=1e-9, atol=1e-15)
    name = 'trace_loss' if kind else 'min_eigval_loss'
    cf = O.closed_form_backward(x, g['g_neighbors'], None, kind=name, normalization=bool(norm), sqrt=bool(sqrt),
                                reduction='sum')
    np.testing.assert_allclose(loss, cf['pointwise'], rtol=1e-9, atol=1e-15)
    # gradient assembled from the coefficients, as the backward kernel does
    d = x[nbr] - out['mean'][:, None, :]
    contrib = c1[:, None, None] * (d @ out['v0'][:, :, None]) * out['v0'][:, None, :] - c2[:, None, None] * d
    gp = np.zeros_like(x)
    np.add.at(gp, nbr.reshape(-1), contrib.reshape(-1, 3))
    np.testing.assert_allclose(gp, cf['grad_points'], rtol=1e-7, atol=1e-9 * np.abs(cf['grad_points']).max())


def test_model_and_incidence_helpers(host):
    host.dc_host_model_depth.restype = ctypes.c_double
    w, e = np.array([-0.06, 0.02, 0.01]), np.array([2.0, 4.0, 1.5])
    for kind, name in ((1, 'Polynomial'), (2, 'ScaledPolynomial')):
        for inc in (0.0, 0.3, 1.2):
            ref = O.model_apply(torch.tensor([[7.0]]), torch.tensor([[inc]], dtype=torch.float64), None,
                                torch.tensor(w[None]), torch.tensor(e[None]), name).item()
            got = host.dc_host_model_depth(kind, 3, _p(w), _p(e), ctypes.c_double(7.0), ctypes.c_double(inc), 1)
            assert abs(got - ref) < 1e-14
    assert host.dc_host_model_depth(2, 3, _p(w), _p(e), ctypes.c_double(7.0), ctypes.c_double(0.5), 0) == 7.0
    normal, inc = np.zeros(3), np.zeros(1)
    d, v = np.array([0.0, 0.6, 0.8]), np.array([0.0, 0.0, 1.0])
    host.dc_host_normal_inc(_p(d), _p(v), _p(normal), _p(inc))
    assert np.allclose(normal, [0, 0, -1]) and abs(inc[0] - np.arccos(0.8)) < 1e-15
    host.dc_host_normal_inc(_p(np.array([1.0, 0, 0])), _p(v), _p(normal), _p(inc))
    assert np.all(normal == 0) and abs(inc[0] - np.pi / 2) < 1e-15          # sign(0) = 0 (depth_cloud.py:401-407)
