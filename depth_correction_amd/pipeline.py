"""Set-up phase of one sequence on the GPU (what train.py:94-215 does once before the optimisation loop):
local feature clouds -> global cloud -> global neighbourhoods -> global mask -> fused SequencePlan.

Array-level functions used by ``bench.py`` and by the reference-API layer (``preproc.py``).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._native import check, dtype_code, lib, ptr, stream_ptr
from .ops import _ws
from .plan import SequencePlan

__all__ = ['local_features', 'local_features_batch', 'global_neighborhoods', 'global_mask', 'build_sequence', 'on_streams', 'DEFAULT_RATIO_BOUNDS']

# config.py:218 default eigenvalue_ratio_bounds
DEFAULT_RATIO_BOUNDS = [[0, 1, 0.0, 0.25], [1, 2, 0.25, 1.0]]


def _eig_masks(mask, eigvals, eigenvalue_bounds=None, eigenvalue_ratio_bounds=None):
    for e, lo, hi in (eigenvalue_bounds or []):                       # filters.py:196-221
        ops.mask_bounds(mask, eigvals, int(e), lo=lo, hi=hi)
    for i, j, lo, hi in (eigenvalue_ratio_bounds or []):              # filters.py:224-254
        ops.mask_bounds(mask, eigvals, int(i), eigvals, int(j), lo, hi)
    return mask


def local_features(xyz, k=None, r=None, vps=None, eigenvalue_bounds=None,
                   eigenvalue_ratio_bounds=DEFAULT_RATIO_BOUNDS, dtype=None, device='cuda:0'):
    """local_feature_cloud (preproc.py:35-64) on raw sensor-frame points: DepthCloud.from_points
    (depth_cloud.py:592-638) + update_all (:435-441) + the eigenvalue masks (:53-62)."""
    pts = torch.as_tensor(np.ascontiguousarray(xyz) if isinstance(xyz, np.ndarray) else xyz, device=device)
    if dtype is not None:
        pts = pts.to(dtype)
    pts = pts.contiguous()
    vps_t = torch.zeros_like(pts) if vps is None else torch.as_tensor(vps, dtype=pts.dtype, device=pts.device).expand_as(pts).contiguous()
    # rays of the points (DepthCloud.from_points, depth_cloud.py:592-638) by the kernel the package's from_points uses: depth = |p - vp|,
    # dirs = (p - vp) / depth, zero-depth rays left un-normalised
    _, dirs, depth, _ = ops.cloud_from_points(pts, vps=None if vps is None else vps_t)
    ps = ops.PointSet(vps_t, dirs, depth)
    x = ops.points_fwd(ps)                                            # update_points
    if k:
        _, nbr = ops.knn(x, k, r=r, want_dist=False)                  # update_neighbors
    else:
        nbr = ops.radius_neighbors(x, r)
    f = ops.features_fwd(x, nbr, dirs=dirs, want=('eigvals', 'normals', 'inc_angles'))
    mask = torch.ones((len(pts),), dtype=torch.bool, device=pts.device)
    _eig_masks(mask, f['eigvals'], eigenvalue_bounds, eigenvalue_ratio_bounds)
    return dict(vps=vps_t, dirs=dirs, depth=depth, inc_angles=f['inc_angles'], mask=mask, normals=f['normals'],
                eigvals=f['eigvals'], neighbors=nbr, points=x)


def local_features_batch(scans, k, r=None, eigenvalue_bounds=None, eigenvalue_ratio_bounds=DEFAULT_RATIO_BOUNDS, dtype=None,
                         device='cuda:0'):
    """local_features of several scans (viewpoints at the sensor origin, k nearest neighbours) in ONE pass: one k-NN build, one
    feature launch, one set of masks over the concatenated scans instead of one each (ten 200 k-point scans: 4.2 -> ~2 ms).

    The scans are set side by side on a lattice (dc_scan_lattice_shift: scan s shifted by an integer offset per axis, two box
    widths apart) so that no neighbourhood crosses scans.  The result is the per-scan result bit for bit when (a) every shifted
    coordinate is exact in fp64 (then every difference, distance and tie is what it was: the k-NN orders equal distances by index)
    and (b) every neighbour of every point lies in its own scan; both are checked on the device, and None is returned when either
    fails (the caller then runs the scans one by one)."""
    pts = ops.cat_rows([torch.as_tensor(np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x, device=device) for x in scans])
    if dtype is not None and pts.dtype != dtype:
        pts = pts.to(dtype)
    pts = pts.contiguous()
    dev = pts.device
    sizes = [len(x) for x in scans]
    if min(sizes) <= k:
        return None
    vps_t = torch.zeros_like(pts)
    _, dirs, depth, _ = ops.cloud_from_points(pts)                           # as local_features: the rays of the points
    x = ops.points_fwd(ops.PointSet(vps_t, dirs, depth))                    # update_points
    if len(scans) > 64:
        return None
    scan_ptr = torch.as_tensor(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64), device=dev)
    info = torch.zeros((1,), dtype=torch.int32, device=dev)
    xs = torch.empty((len(pts), 3), dtype=torch.float64, device=dev)
    nbytes = lib().dc_scan_lattice_workspace_bytes(len(scans))
    check(lib().dc_scan_lattice_shift(ptr(x), dtype_code(x), len(pts), ptr(scan_ptr), len(scans), ptr(xs), ptr(info), ptr(_ws(nbytes, dev)),
                                      nbytes, stream_ptr()), 'dc_scan_lattice_shift')
    _, nbr = ops.knn(xs, k, r=r, want_dist=False)                            # update_neighbors, all scans at once
    nbr_global = nbr.clone()
    check(lib().dc_scan_lattice_localize(ptr(nbr), len(pts), k, ptr(scan_ptr), len(scans), ptr(info), stream_ptr()), 'dc_scan_lattice_localize')
    if int(info.item()) != 0:
        return None
    f = ops.features_fwd(x, nbr_global, dirs=dirs, want=('eigvals', 'normals', 'inc_angles'))
    mask = torch.ones((len(pts),), dtype=torch.bool, device=dev)
    _eig_masks(mask, f['eigvals'], eigenvalue_bounds, eigenvalue_ratio_bounds)
    nbr_local = nbr
    out, a = [], 0
    for n_ in sizes:
        sl = slice(a, a + n_)
        out.append(dict(vps=vps_t[sl], dirs=dirs[sl], depth=depth[sl], inc_angles=f['inc_angles'][sl], mask=mask[sl],
                        normals=f['normals'][sl], eigvals=f['eigvals'][sl], neighbors=nbr_local[sl], points=x[sl]))
        a += n_
    return out


def global_cloud_arrays(clouds, poses):
    """preproc.global_cloud (preproc.py:80-119) without a model: (points, vps, dirs) of the concatenated cloud."""
    dev = clouds[0]['dirs'].device
    sizes = [len(c['dirs']) for c in clouds]
    scan_id = ops.scan_ids(sizes, dev)
    cat = lambda f: ops.cat_rows([c[f].reshape(len(c['dirs']), -1) for c in clouds])
    ps = ops.PointSet(cat('vps'), cat('dirs'), cat('depth'), None, None, scan_id)
    P = torch.as_tensor(poses, device=dev).to(torch.float64)[:, :3, :].reshape(len(clouds), 12).contiguous()
    return ops.points_fwd(ps, P, want_parts=True)


def global_neighborhoods(points, k=None, r=None):
    """establish_neighborhoods (preproc.py:168-185) for the ball type: (neighbors i32 [N,K], weights implied)."""
    if k:
        return ops.knn(points, k, r=r, want_dist=False)[1]
    return ops.radius_neighbors(points, r)


def global_mask(local_mask, points, vps, dirs, nbr, min_valid_neighbors=5, eigenvalue_bounds=None,
                eigenvalue_ratio_bounds=DEFAULT_RATIO_BOUNDS, dir_dispersion_bounds=None, vp_dispersion_bounds=None):
    """global_cloud_mask (preproc.py:122-164)."""
    mask = torch.ones((len(points),), dtype=torch.bool, device=points.device) if local_mask is None else local_mask.clone()
    if min_valid_neighbors:
        ops.mask_bounds(mask, ops.valid_count(nbr).to(torch.float64), lo=min_valid_neighbors)
    if eigenvalue_bounds or eigenvalue_ratio_bounds:
        ev = ops.features_fwd(points, nbr, want=('eigvals',))['eigvals']
        _eig_masks(mask, ev, eigenvalue_bounds, eigenvalue_ratio_bounds)
    if dir_dispersion_bounds:
        ops.mask_bounds(mask, ops.dispersion(dirs, nbr), lo=dir_dispersion_bounds[0], hi=dir_dispersion_bounds[1])
    if vp_dispersion_bounds:
        ops.mask_bounds(mask, ops.dispersion(vps, nbr), lo=vp_dispersion_bounds[0], hi=vp_dispersion_bounds[1])
    return mask


def on_streams(jobs, device, n_streams=4):
    """Run independent jobs (callables issuing GPU work on torch's current stream) round robin on ``n_streams`` side streams
    and join them with the caller's stream; results in job order.  Tensors the jobs allocate belong to the side streams'
    pools; the join makes every later use on the caller's stream ordered after their producers."""
    jobs = list(jobs)
    if n_streams <= 1 or len(jobs) <= 1 or not torch.cuda.is_available():
        return [job() for job in jobs]
    dev = torch.device(device)
    main = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(min(n_streams, len(jobs)))]
    for s_ in streams:
        s_.wait_stream(main)
    out = []
    for i, job in enumerate(jobs):
        with torch.cuda.stream(streams[i % len(streams)]):
            out.append(job())
    for s_ in streams:
        main.wait_stream(s_)
    # the results were allocated in the side streams' pools: tell the allocator that the caller's stream uses them from here
    # on, or a later free would hand the blocks back to a side stream with nothing ordering the reuse after these readers
    for r in out:
        for t in _tensors_of(r):
            t.record_stream(main)
    return out


def _tensors_of(obj, depth=0):
    """The torch tensors reachable from a job's result: tensors, containers of them, objects holding them as attributes
    (DepthCloud fields)."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors_of(o, depth + 1)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors_of(o, depth + 1)
    elif depth < 2 and hasattr(obj, '__dict__'):
        for o in vars(obj).values():
            yield from _tensors_of(o, depth + 1)


class _Stages(object):
    """Wall-clock stage timer of the set-up phase (synchronises the device at every mark; only when asked for)."""

    def __init__(self, on, device):
        import time
        self.on, self.device, self.t, self.out, self._time = on, device, None, {}, time
        if on:
            torch.cuda.synchronize(device)
            self.t = time.perf_counter()

    def mark(self, name):
        if self.on:
            torch.cuda.synchronize(self.device)
            now = self._time.perf_counter()
            self.out[name] = self.out.get(name, 0.0) + (now - self.t) * 1e3
            self.t = now


def build_sequence(scans_xyz, poses, k=10, r=None, dtype=torch.float32, device='cuda:0', min_valid_neighbors=5,
                   eigenvalue_ratio_bounds=DEFAULT_RATIO_BOUNDS, vp_dispersion_bounds=None, model_kind='ScaledPolynomial',
                   loss='min_eigval_loss', normalization=True, sqrt=False, spatial_sort=True, point_format='auto',
                   active_only=False, degree_sort=False, block_tables=True, bwd_layout='runs', stage_times=False, basis=True,
                   local_streams=4, scan_group=True, mask_first=False, degree_group=None, batch_local=True, heavy_first=None,
                   wave_pack=None):
    """Everything train.py does before its loop for one sequence; returns (plan, info).  ``stage_times``: info['setup_ms']
    = wall-clock milliseconds per stage (device synchronised between stages)."""
    st = _Stages(stage_times, device)
    if all(isinstance(xyz, np.ndarray) and xyz.ndim == 2 and xyz.dtype == scans_xyz[0].dtype and xyz.shape[1] == scans_xyz[0].shape[1]
           for xyz in scans_xyz):
        # all scans into ONE device array (a copy from the host each): the per-scan clouds are then row ranges of it and nothing
        # concatenates them again (pipeline.local_features_batch, ops.cat_rows)
        sizes_ = [len(xyz) for xyz in scans_xyz]
        buf = torch.empty((sum(sizes_), scans_xyz[0].shape[1]), dtype=torch.from_numpy(scans_xyz[0][:0]).dtype, device=device)
        uploaded, a_ = [], 0
        for xyz, n_ in zip(scans_xyz, sizes_):
            buf[a_:a_ + n_].copy_(torch.from_numpy(np.ascontiguousarray(xyz)))
            uploaded.append(buf[a_:a_ + n_])
            a_ += n_
    else:
        uploaded = [torch.as_tensor(np.ascontiguousarray(xyz) if isinstance(xyz, np.ndarray) else xyz, device=device) for xyz in scans_xyz]
    st.mark('upload')
    # the scans are independent and one 200 k-point scan does not fill the chip (782 blocks for > 1024 resident): their
    # pipelines go to a few streams side by side
    clouds = None
    if k and batch_local and len(uploaded) > 1:
        clouds = local_features_batch(uploaded, k, r=r, eigenvalue_ratio_bounds=eigenvalue_ratio_bounds, dtype=dtype, device=device)
    if clouds is None:
        clouds = on_streams([lambda xyz=xyz: local_features(xyz, k=k, r=r, eigenvalue_ratio_bounds=eigenvalue_ratio_bounds, dtype=dtype,
                                                              device=device) for xyz in uploaded], device, n_streams=local_streams)
    st.mark('local_feature_clouds')
    poses_t = torch.as_tensor(np.asarray(poses), dtype=torch.float64, device=device)
    x0, vps0, dirs0, _ = global_cloud_arrays(clouds, poses_t)
    st.mark('global_cloud')
    nbr = global_neighborhoods(x0, k=k, r=r)
    st.mark('global_neighborhoods')
    lmask = torch.cat([c['mask'] for c in clouds])
    mask = global_mask(lmask, x0, vps0, dirs0, nbr, min_valid_neighbors=min_valid_neighbors,
                       eigenvalue_ratio_bounds=eigenvalue_ratio_bounds, vp_dispersion_bounds=vp_dispersion_bounds)
    st.mark('global_mask')
    plan = SequencePlan(clouds, poses_t, nbr, mask, model_kind=model_kind, loss=loss, normalization=normalization,
                        sqrt=sqrt, spatial_sort=spatial_sort, point_format=point_format, active_only=active_only,
                        degree_sort=degree_sort, block_tables=block_tables, bwd_layout=bwd_layout, stages=st, basis=basis,
                        scan_group=scan_group, mask_first=mask_first,
                        degree_group=(r is not None) if degree_group is None else degree_group,
                        heavy_first=(r is not None) if heavy_first is None else heavy_first, wave_pack=wave_pack)
    st.mark('plan_other')
    return plan, dict(clouds=clouds, poses=poses_t, neighbors=nbr, mask=mask, points0=x0, setup_ms=st.out)
