"""Shared helpers of the parity tests (test infrastructure)."""
import numpy as np
import torch


def t(x, dev=None, dtype=None):
    x = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None and x.dtype.is_floating_point:
        x = x.to(dtype)
    return x.to(dev) if dev is not None else x


def npy(x):
    return x.detach().cpu().numpy()


def scans_from_golden(g, dtype=torch.float64):
    """Per-scan local inputs of a fixture as CPU tensors (vps are zeros: the sensor frame)."""
    scans = []
    for s in range(int(g['n_scans'])):
        dirs = t(g['scan%d_dirs' % s], dtype=dtype)
        scans.append(dict(vps=torch.zeros_like(dirs), dirs=dirs, depth=t(g['scan%d_depth' % s], dtype=dtype),
                          inc=t(g['scan%d_inc_angles' % s], dtype=dtype), mask=t(g['scan%d_mask' % s])))
    return scans


def concat_scans(scans, dev):
    """Sequence layout of the kernels: local scans concatenated + scan ids."""
    from depth_correction_amd.ops import PointSet
    cat = lambda k: torch.cat([s[k] for s in scans]).contiguous().to(dev)
    sid = torch.cat([torch.full((len(s['dirs']),), i, dtype=torch.int32) for i, s in enumerate(scans)]).to(dev)
    return PointSet(cat('vps'), cat('dirs'), cat('depth'), cat('inc'), cat('mask'), sid)


def poses12(poses, dev):
    return torch.as_tensor(poses, dtype=torch.float64)[..., :3, :].reshape(-1, 12).contiguous().to(dev)


def rel_err(a, b, floor=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def assert_eigvals_close(lam, ref, rtol, what=''):
    """|dlam| <= rtol |lam| + 1e-12 |C|: relative bar plus LAPACK's own absolute noise floor (eps * norm)."""
    lam, ref = np.asarray(lam, np.float64), np.asarray(ref, np.float64)
    tol = rtol * np.abs(ref) + 1e-12 * np.abs(ref).max(axis=-1, keepdims=True)
    bad = np.abs(lam - ref) > tol
    assert not bad.any(), '%s: %d eigenvalues off, worst rel %.3g' % (what, bad.sum(), (np.abs(lam - ref) / np.abs(ref))[bad].max())
