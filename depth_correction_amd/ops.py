"""Operator layer: one Python function per C-ABI entry point (include/dc_hip.h), torch tensors in and out.

No autograd here (see ``autograd.py``) and no CPU implementation: every function requires GPU tensors
and the HIP library.  Shapes follow the reference's tensors (``[N,3]`` points, ``[N,K]`` neighbours ...).
"""
from __future__ import annotations

import ctypes

import torch

from . import _native as nv
from ._native import lib, check, need, ptr, stream_ptr, dtype_code, on_device

__all__ = [
    'knn', 'radius_neighbors', 'knn_transpose', 'BlockTable', 'block_table', 'table_to_csr', 'spatial_order', 'points_fwd', 'points_bwd', 'features_fwd',
    'features_bwd', 'consistency_fwd', 'consistency_bwd', 'mask_bounds', 'valid_count', 'dispersion', 'p2plane_pair', 'p2point_pair',
    'IcpSequence', 'shadow_mask', 'shadow_filter', 'correct_depth', 'cloud_from_points', 'mask_bounds_all', 'compact_rows', 'to_points', 'valid_weights', 'scan_prefilter',
    'as_index32', 'scan_ids', 'points_extent', 'gather_rows', 'cat_rows',
]


def _ws(nbytes, device):
    return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)


def as_index32(neighbors):
    """Reference neighbour tensors are int64 (nearest_neighbors.py:78); the kernels take int32."""
    if neighbors.dtype == torch.int32:
        return neighbors.contiguous()
    assert neighbors.dtype == torch.int64
    return neighbors.to(torch.int32).contiguous()


# ------------------------------------------------------------------------------------------------
# neighbourhood builder
# ------------------------------------------------------------------------------------------------
@on_device
def knn(points, k, r=None, query=None, cell_hint=0.0, want_dist=True, want_index64=False):
    """k-NN of ``query`` (default: ``points`` itself) in ``points``: (dist f64 [M,k] | None, idx i32 [M,k]); with
    ``want_index64`` a third result, the same table as int64 written by the same kernels (dc_knn_build_i64)."""
    need(points, (None, 3), name='points')
    n = points.shape[0]
    if not (1 <= k <= 64):
        raise ValueError('k must be in 1..64, got %r' % (k,))
    if query is not None:
        need(query, (None, 3), dtype=points.dtype, name='query', device=points.device)
    m = n if query is None else query.shape[0]
    idx = torch.empty((m, k), dtype=torch.int32, device=points.device)
    # (the kernels write every entry of both tables -- inf beside a missing neighbour --; only the empty cloud needs the fill)
    dist = (torch.empty if n else torch.full)((m, k), *(() if n else (float('inf'),)), dtype=torch.float64, device=points.device) if want_dist else None
    if n == 0:
        idx.fill_(-1)
        return (dist, idx, idx.long()) if want_index64 else (dist, idx)
    idx64 = torch.empty((m, k), dtype=torch.int64, device=points.device) if want_index64 else None
    nbytes = lib().dc_knn_workspace_bytes(n, 0 if query is None else m)
    ws = _ws(nbytes, points.device)
    check(lib().dc_knn_build_i64(ptr(points), 3, dtype_code(points), n, ptr(query), 3, 0 if query is None else m, k,
                                 float(r) if r else 0.0, float(cell_hint), ptr(idx), ptr(idx64), ptr(dist), ptr(ws), nbytes,
                                 stream_ptr()), 'dc_knn_build_i64')
    return (dist, idx, idx64) if want_index64 else (dist, idx)


@on_device
def nn1_corr(dist, idx, ratio):
    """(mask bool [n], idx[mask] int32 [m], threshold fp64 0-dim): the inlier correspondences of a scan pair from the 1-NN distances
    (dc_nn1_corr: np.quantile by a radix select on the device, train.py:186-193).  One synchronisation: the survivors' count."""
    need(dist, (None,), dtype=torch.float64, name='dist')
    n = dist.shape[0]
    need(idx, (n,), dtype=torch.int32, name='idx', device=dist.device)
    mask = torch.empty((n,), dtype=torch.uint8, device=dist.device)
    out = torch.empty((max(n, 1),), dtype=torch.int32, device=dist.device)
    count = torch.zeros((1,), dtype=torch.int64, device=dist.device)
    th = torch.empty((1,), dtype=torch.float64, device=dist.device)
    nbytes = lib().dc_nn1_corr_workspace_bytes(n)
    ws = _ws(nbytes, dist.device)
    check(lib().dc_nn1_corr(ptr(dist), ptr(idx), n, float(ratio), ptr(mask), ptr(out), ptr(count), ptr(th), ptr(ws), nbytes, stream_ptr()),
          'dc_nn1_corr')
    return mask.view(torch.bool), out[:int(count.item())], th[0]


@on_device
def gather_rows(src, order):
    """src[order] for a contiguous per-point array [n, ...] and an int64 permutation (dc_gather_rows: one kernel family for every
    dtype and row shape)."""
    row_bytes = src.element_size() * (src[0].numel() if src.shape[0] else 1)
    if not (src.is_contiguous() and order.dtype == torch.int64 and order.is_contiguous() and (row_bytes == 1 or row_bytes % 4 == 0)):
        return src[order].contiguous()
    out = torch.empty((order.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    check(lib().dc_gather_rows(ptr(src), row_bytes, ptr(order), order.shape[0], ptr(out), stream_ptr()), 'dc_gather_rows')
    return out


def cat_rows(parts):
    """torch.cat(parts) -- without a copy when the parts are consecutive row ranges of ONE contiguous allocation (the clouds of
    pipeline.local_features_batch are slices of the arrays it built for all scans at once): a view over all of them."""
    parts = list(parts)
    first = parts[0]
    if len(parts) > 1 and first.dim() >= 1 and all(p.is_contiguous() and p.dtype == first.dtype and p.device == first.device
                                                   and p.shape[1:] == first.shape[1:] for p in parts):
        row = first.element_size() * (first[0].numel() if first.shape[0] else 0)
        store = first.untyped_storage().data_ptr()
        at, ok = first.data_ptr(), row > 0
        for p_ in parts:
            ok = ok and p_.untyped_storage().data_ptr() == store and p_.data_ptr() == at
            at += p_.shape[0] * row
        if ok:
            n = sum(p_.shape[0] for p_ in parts)
            return torch.as_strided(first, (n,) + tuple(first.shape[1:]), first.stride())
    return torch.cat(parts).contiguous()


def scan_ids(sizes, device):
    """int32 [sum(sizes)]: the scan of every row of scans concatenated in order (torch.repeat_interleave(arange(S), sizes) in one
    small launch: dc_scan_ids)."""
    import numpy as _np
    n = int(sum(sizes))
    out = torch.empty((n,), dtype=torch.int32, device=device)
    if n == 0:
        return out
    scan_ptr = torch.as_tensor(_np.concatenate([[0], _np.cumsum(sizes)]).astype(_np.int64), device=device)
    with torch.cuda.device(out.device):
        check(lib().dc_scan_ids(ptr(scan_ptr), len(sizes), n, ptr(out), stream_ptr()), 'dc_scan_ids')
    return out


@on_device
def points_extent(points):
    """(lo, hi): per-axis minimum and maximum of a cloud [n, 3 | 4] (float32 / float64) as Python lists -- ONE synchronisation
    (dc_points_extent; the extent the q32 point format is sized for)."""
    need(points, (None, None), name='points')
    n, stride = points.shape
    if n == 0:
        return [float('inf')] * 3, [float('-inf')] * 3
    out = torch.empty((6,), dtype=torch.float64, device=points.device)
    nbytes = lib().dc_points_extent_workspace_bytes()
    check(lib().dc_points_extent(ptr(points), stride, dtype_code(points), n, ptr(out), ptr(_ws(nbytes, points.device)), nbytes,
                                 stream_ptr()), 'dc_points_extent')
    v = out.tolist()
    return v[:3], v[3:]


@on_device
def radius_neighbors(points, r, query=None):
    """All neighbours within ``r`` (inclusive) of every point -- or of every row of ``query``, another cloud -- ascending
    index, padded with -1: idx i32 [N | M, Kmax]."""
    need(points, (None, 3), name='points')
    n = points.shape[0]
    if query is not None:
        need(query, (None, 3), dtype=points.dtype, name='query', device=points.device)
        m = query.shape[0]
        if m == 0:
            return torch.empty((0, 0), dtype=torch.int32, device=points.device)
        nbytes = lib().dc_knn_workspace_bytes(n, m)
        ws = _ws(nbytes, points.device)
        count = torch.empty((m,), dtype=torch.int32, device=points.device)
        kmax = torch.zeros((1,), dtype=torch.int32, device=points.device)
        check(lib().dc_radius_count_query(ptr(points), 3, dtype_code(points), n, ptr(query), 3, m, float(r), ptr(count), ptr(kmax),
                                          ptr(ws), nbytes, stream_ptr()), 'dc_radius_count_query')
        km = max(int(kmax.item()), 1)
        if m * km * 4 > (16 << 30):
            raise MemoryError('radius neighbourhoods too dense for a padded [M, Kmax] table (%d x %d)' % (m, km))
        idx = torch.empty((m, km), dtype=torch.int32, device=points.device)
        check(lib().dc_radius_fill_query(n, m, float(r), km, ptr(idx), ptr(ws), nbytes, stream_ptr()), 'dc_radius_fill_query')
        return idx
    if n == 0:
        return torch.empty((0, 0), dtype=torch.int32, device=points.device)
    nbytes = lib().dc_knn_workspace_bytes(n, 0)
    ws = _ws(nbytes, points.device)
    count = torch.empty((n,), dtype=torch.int32, device=points.device)
    kmax = torch.zeros((1,), dtype=torch.int32, device=points.device)
    check(lib().dc_radius_count(ptr(points), 3, dtype_code(points), n, float(r), ptr(count), ptr(kmax), ptr(ws), nbytes,
                                stream_ptr()), 'dc_radius_count')
    km = int(kmax.item())            # set-up phase: the padded width is needed on the host
    if n * max(km, 1) * 4 > (16 << 30):
        raise MemoryError('radius neighbourhoods too dense: %d points x %d neighbours do not fit a padded [N, Kmax] table; '
                          'voxel-filter the cloud first (cfg.grid_res) as the reference pipeline does' % (n, km))
    idx = torch.empty((n, max(km, 1)), dtype=torch.int32, device=points.device)
    check(lib().dc_radius_fill(n, float(r), max(km, 1), ptr(idx), ptr(ws), nbytes, stream_ptr()), 'dc_radius_fill')
    return idx


@on_device
def knn_transpose(nbr, n_dst=None):
    """(csr_ptr i32 [n_dst+1], csr_src i32 [rows*K]) -- for every point the rows whose neighbourhood contains it.
    ``n_dst`` (default: the number of rows) is the number of points the indices refer to."""
    need(nbr, (None, None), dtype=torch.int32, name='neighbors')
    n, k = nbr.shape
    n_dst = n if n_dst is None else int(n_dst)
    csr_ptr = torch.empty((n_dst + 1,), dtype=torch.int32, device=nbr.device)
    csr_src = torch.empty((max(n * k, 1),), dtype=torch.int32, device=nbr.device)
    nbytes = lib().dc_knn_transpose_workspace_bytes(n, k)
    ws = _ws(nbytes, nbr.device)
    check(lib().dc_knn_transpose(ptr(nbr), n, k, n_dst, ptr(csr_ptr), ptr(csr_src), ptr(ws), nbytes, stream_ptr()),
          'dc_knn_transpose')
    return csr_ptr, csr_src


class BlockTable:
    """Block table of a reference list (dcBlockTable): per block of 256 rows the distinct rows it references and the
    16-bit LDS positions (16 x index in that list) of every reference -- slot-major (``slot_ptr``; the forward's
    [rows, K] table) or as per-row runs padded to four (``run_ptr``; the backward's incoming-edge lists).  Lets the
    fused kernels gather from LDS."""

    def __init__(self, blk_ptr, blk_ids, slot_ptr, loc, max_rows, n_rows, run_ptr=None, own_base=None, packed=0, row_ptr=None):
        self.blk_ptr, self.blk_ids, self.slot_ptr, self.loc, self.run_ptr = blk_ptr, blk_ids, slot_ptr, loc, run_ptr
        self.own_base, self.packed, self.row_ptr = own_base, int(packed), row_ptr
        self.max_rows, self.n_rows = int(max_rows), int(n_rows)
        self.layout = nv.DC_TABLE_SLOTS if run_ptr is None else nv.DC_TABLE_RUNS
        self.device = blk_ptr.device
        self.desc = nv.BlockTableDesc(ptr(blk_ptr), ptr(blk_ids), ptr(slot_ptr), ptr(loc), self.max_rows, self.layout, ptr(run_ptr),
                                      ptr(own_base), self.packed, 0, ptr(row_ptr))

    def ref(self):
        return ctypes.cast(ctypes.pointer(self.desc), ctypes.c_void_p)

    @property
    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in (self.blk_ptr, self.blk_ids, self.slot_ptr, self.loc, self.run_ptr)
                   if t is not None)


def _table_ref(table, n_rows):
    if table is None:
        return None
    assert isinstance(table, BlockTable) and table.n_rows == n_rows, 'block table built for another row count'
    return table.ref()


@on_device
def block_table(nbr=None, csr=None, layout=None, own_rows=True):
    """BlockTable of a neighbour table ``nbr`` int32 [rows, K] (forward; slot-major) or of CSR lists ``csr`` = (ptr, ids)
    (backward: knn_transpose's output; ``layout`` 'runs' (default) or 'slots').  ``own_rows``: the table's rows and ids
    are the same points (not a compact centre list), so the position of each block's own rows in its list is recorded.
    Returns None when a block references 4095 or more distinct rows."""
    if nbr is not None:
        need(nbr, (None, None), dtype=torch.int32, name='neighbors')
        n_rows, k = nbr.shape
        row_ptr, ids, n_refs, dev = None, nbr, n_rows * k, nbr.device
        layout = layout or 'slots'
        if layout != 'slots':
            raise ValueError('a neighbour table [rows, K] is stored slot-major')
    else:
        row_ptr, ids = csr
        need(row_ptr, (None,), dtype=torch.int32, name='csr_ptr')
        need(ids, (None,), dtype=torch.int32, name='csr_src', device=row_ptr.device)
        n_rows, k, n_refs, dev = row_ptr.shape[0] - 1, 0, ids.shape[0], row_ptr.device
        layout = layout or 'runs'
    assert layout in ('slots', 'runs')
    nb = (n_rows + 255) // 256
    blk_ptr = torch.empty((nb + 1,), dtype=torch.int32, device=dev)
    blk_ids = torch.empty((max(n_refs, 1),), dtype=torch.int32, device=dev)
    info = torch.empty((4,), dtype=torch.int32, device=dev)
    if nbr is not None:
        nbytes = lib().dc_block_table_slots_workspace_bytes(n_rows, k)      # (the LDS build: a tenth of the radix build's)
    else:
        nbytes = lib().dc_block_table_workspace_bytes(max(n_refs, n_rows + 1))
    ws = _ws(nbytes, dev)
    slot_ptr = run_ptr = None
    if layout == 'runs':
        run_ptr = torch.empty((n_rows + 1,), dtype=torch.int32, device=dev)
        loc = torch.empty((lib().dc_block_table_run_capacity(n_rows, n_refs) * 4,), dtype=torch.uint16, device=dev)
        check(lib().dc_block_table_build_runs(ptr(row_ptr), ptr(ids), n_rows, n_refs, ptr(run_ptr), ptr(blk_ptr), ptr(blk_ids),
                                              ptr(loc), ptr(info), ptr(ws), nbytes, stream_ptr()), 'dc_block_table_build_runs')
    else:
        slot_ptr = torch.empty((nb + 1,), dtype=torch.int32, device=dev)
        if row_ptr is None:
            # a table [rows, K]: every block has K slots, known without asking the device
            import numpy as _np
            slot_ptr = torch.as_tensor(_np.arange(nb + 1, dtype=_np.int32) * _np.int32(k), device=dev)      # (a copy from the host: no kernel)
            n_slot_rows = nb * k
        else:
            cnt = torch.empty((max(nb, 1),), dtype=torch.int32, device=dev)
            check(lib().dc_block_table_slots(ptr(row_ptr), n_rows, k, ptr(cnt), ptr(slot_ptr), stream_ptr()), 'dc_block_table_slots')
            n_slot_rows = int(slot_ptr[-1])                          # one synchronisation, at set-up time
        # (eight slot rows of slack: the ragged kernels request the next trip's positions before they know it is the last one)
        loc = torch.empty(((max(n_slot_rows, 1) + 8) * 256,), dtype=torch.uint16, device=dev)
        check(lib().dc_block_table_build(ptr(row_ptr), ptr(ids), n_rows, k, n_refs, ptr(slot_ptr), n_slot_rows, ptr(blk_ptr),
                                         ptr(blk_ids), ptr(loc), ptr(info), ptr(ws), nbytes, stream_ptr()), 'dc_block_table_build')
    total, max_rows, overflow, _ = info.tolist()
    if overflow:
        return None
    own_base = None
    if layout == 'slots' and own_rows and (row_ptr is None or own_rows == 'csr'):
        # rows and ids of a k-NN table share one index space: where does every block find its own rows in its list?
        own_base = torch.empty((max(nb, 1),), dtype=torch.int32, device=dev)
        check(lib().dc_block_table_own_base(ptr(blk_ptr), ptr(blk_ids), n_rows, ptr(own_base), stream_ptr()),
              'dc_block_table_own_base')
    packed = 1 if (row_ptr is not None and layout == 'slots') else 0
    # (forward tables from CSR lists keep the offsets: the ragged one-pass kernel takes the rows' lengths from them; it needs every
    # block's own rows in its list -- own_base >= 0 throughout -- because its padding slots read the lane's own row)
    own_all = packed and own_base is not None and nb > 0 and bool((own_base >= 0).all())
    return BlockTable(blk_ptr, blk_ids[:max(total, 1)].clone(), slot_ptr, loc, max_rows, n_rows, run_ptr=run_ptr,
                      own_base=own_base, packed=packed, row_ptr=row_ptr if own_all else None)


def table_to_csr(nbr):
    """(ptr int32 [rows + 1], ids int32 [E]) of a padded neighbour table [rows, Kmax] with -1 for missing entries (radius
    neighbourhoods, nearest_neighbors.py:69-73): its valid entries row by row, in their order."""
    need(nbr, (None, None), dtype=torch.int32, name='neighbors')
    valid = nbr >= 0
    ptr_ = torch.zeros((nbr.shape[0] + 1,), dtype=torch.int32, device=nbr.device)
    ptr_[1:] = valid.sum(dim=1).cumsum(dim=0).to(torch.int32)
    return ptr_, nbr[valid].contiguous()


@on_device
def spatial_order(points):
    need(points, (None, 3), name='points')
    n = points.shape[0]
    order = torch.empty((n,), dtype=torch.int32, device=points.device)
    nbytes = lib().dc_spatial_order_workspace_bytes(n)
    ws = _ws(nbytes, points.device)
    check(lib().dc_spatial_order(ptr(points), 3, dtype_code(points), n, ptr(order), ptr(ws), nbytes, stream_ptr()),
          'dc_spatial_order')
    return order


# ------------------------------------------------------------------------------------------------
# points (model + pose + ray end point)
# ------------------------------------------------------------------------------------------------
class PointSet:
    """Validated per-point inputs of one sequence (local scans concatenated)."""

    def __init__(self, vps, dirs, depth, inc=None, lmask=None, scan_id=None):
        need(dirs, (None, 3), name='dirs')
        n = dirs.shape[0]
        dev, dt = dirs.device, dirs.dtype
        if vps is not None:                  # None: every viewpoint is the sensor origin
            need(vps, (n, 3), dtype=dt, name='vps', device=dev)
        depth = depth.reshape(-1)
        need(depth, (n,), dtype=dt, name='depth', device=dev)
        if inc is not None:
            inc = inc.reshape(-1)
            need(inc, (n,), dtype=dt, name='inc_angles', device=dev)
        if lmask is not None:
            need(lmask, (n,), dtype=torch.bool, name='mask', device=dev)
        if scan_id is not None:
            need(scan_id, (n,), dtype=torch.int32, name='scan_id', device=dev)
        self.vps, self.dirs, self.depth, self.inc, self.lmask, self.scan_id = vps, dirs, depth, inc, lmask, scan_id
        self.n, self.device, self.dtype = n, dev, dt


def _model_args(model_kind, w, e, ps):
    kind = nv.MODEL_KINDS[model_kind] if not isinstance(model_kind, int) else model_kind
    if kind == 0:
        return 0, 0, None, None
    need(w, (None,), dtype=torch.float64, name='w', device=ps.device)
    need(e, (w.shape[0],), dtype=torch.float64, name='exponent', device=ps.device)
    if not (1 <= w.shape[0] <= nv.MAX_MODEL_TERMS):
        raise ValueError('model must have 1..%d terms' % nv.MAX_MODEL_TERMS)
    if ps.inc is None:
        raise ValueError('the model needs incidence angles')
    return kind, w.shape[0], w, e


def _pose_args(poses, ps):
    if poses is None:
        if ps.scan_id is not None:
            raise ValueError('scan_id given without poses')
        return None, 0
    need(poses, (None, 12), dtype=torch.float64, name='poses[S,12]', device=ps.device)
    return poses, poses.shape[0]


class QFormat:
    """Fixed-point parameters of the DC_Q32 point format: x = origin + q * scale (q int32)."""

    def __init__(self, origin, scale):
        self.origin = [float(v) for v in origin]
        self.scale = float(scale)
        self._c = (ctypes.c_double * 4)(*self.origin, self.scale)

    # Resolution up to which 32-bit fixed point keeps neighbourhood eigenvalues within 1e-5 of the fp64 result when the
    # surfaces are as smooth as a millimetre-noise lidar sees them (lambda0 ~ 1e-6 m^2): the rounding of K = 10 points
    # perturbs lambda0 by about 0.2 * sqrt(lambda0) * scale, i.e. 1e-5 * lambda0 at scale = 5e-8 m (a map of ~50 m with
    # the 4x pose margin).  Larger maps keep their points in fp64 instead (SequencePlan, point_format='auto').
    MAX_AUTO_SCALE = 2.0 ** -24

    @staticmethod
    def for_extent(lo, hi, margin=4.0):
        """Power-of-two resolution covering `margin` x the half extent of the box [lo, hi] about its centre."""
        import math
        origin = [(a + b) / 2 for a, b in zip(lo, hi)]
        half = max(max((b - a) / 2 for a, b in zip(lo, hi)), 1e-3)
        return QFormat(origin, 2.0 ** (math.ceil(math.log2(half * margin)) - 31))


def _fmt_args(points_dtype, qfmt):
    """(point_fmt code, qparams pointer) for a points / rec buffer."""
    if qfmt is not None:
        return nv.DC_Q32, qfmt._c
    return (nv.DC_F32 if points_dtype == torch.float32 else nv.DC_F64), None


def _check_points(points, qfmt, name='points'):
    need(points, (None, None), name=name)
    if qfmt is not None:
        need(points, (None, 4), dtype=torch.int32, name=name + ' (q32)')
    elif points.shape[1] not in (3, 4) or not points.dtype.is_floating_point:
        raise ValueError('%s must be float [N,3] or [N,4]' % name)


@on_device
def points_fwd(ps, poses=None, model_kind=None, w=None, e=None, stride=3, want_parts=False, qfmt=None, out=None, status=None):
    """x = pose(vps) + model(depth) * pose(dirs); optionally also (vps', dirs', depth').
    With ``qfmt`` the points are written as int32 fixed-point rows [N,4] (DC_Q32); ``status`` (int32 [1], optional)
    then gets bit 0 set when a coordinate does not fit the format's extent or is NaN."""
    kind, nt, w, e = _model_args(model_kind, w, e, ps)
    poses, ns = _pose_args(poses, ps)
    if qfmt is not None:
        stride = 4
        if ps.dtype != torch.float32:
            raise TypeError('the q32 point format goes with float32 inputs')
    x = out if out is not None else torch.empty((ps.n, stride), dtype=torch.int32 if qfmt is not None else ps.dtype,
                                                device=ps.device)
    _check_points(x, qfmt, 'points_out')
    assert x.shape == (ps.n, stride) and x.device == ps.device
    fmt, qptr = _fmt_args(ps.dtype, qfmt)
    parts = (torch.empty_like(ps.dirs), torch.empty_like(ps.dirs), torch.empty_like(ps.depth)) if want_parts \
        else (None, None, None)
    check(lib().dc_points_fwd(ptr(ps.vps), ptr(ps.dirs), ptr(ps.depth), ptr(ps.inc), ptr(ps.lmask), ptr(ps.scan_id),
                              ptr(poses), ns, kind, nt, ptr(w), ptr(e), ps.n, dtype_code(ps.dirs), fmt, qptr, stride,
                              ptr(x), ptr(parts[0]), ptr(parts[1]), ptr(parts[2]), ptr(status), stream_ptr()), 'dc_points_fwd')
    return (x,) + parts if want_parts else x


def _grads_split(g, nt, ns):
    return g[:nt], g[nt:2 * nt], g[2 * nt:].reshape(ns, 3, 4)


@on_device
def points_bwd(grad_points, ps, poses=None, model_kind=None, w=None, e=None, want_exponent=False, want_pose=False,
               perm=None, out=None):
    """(dL/dw [P], dL/dexponent [P], dL/d[R|t] [S,3,4]) for a given dL/dpoints; ``perm`` int32 [N]: point i uses
    row perm[i] of ``grad_points``."""
    kind, nt, w, e = _model_args(model_kind, w, e, ps)
    poses, ns = _pose_args(poses, ps)
    need(grad_points, (ps.n, None), dtype=ps.dtype, name='grad_points', device=ps.device)
    if perm is not None:
        need(perm, (ps.n,), dtype=torch.int32, name='perm', device=ps.device)
    stride = grad_points.shape[1]
    nacc = 2 * nt + 12 * ns
    if out is None:
        out = torch.zeros((max(nacc, 1),), dtype=torch.float64, device=ps.device)
    else:
        need(out, (nacc,), dtype=torch.float64, name='grads_out', device=ps.device)
    if nacc == 0:
        return _grads_split(out[:0], 0, 0)
    rows = lib().dc_partial_rows(ps.n)
    part = torch.empty((rows * nacc,), dtype=torch.float64, device=ps.device)
    check(lib().dc_points_bwd(ptr(grad_points), ptr(perm), stride, dtype_code(grad_points), ps.n, ptr(ps.vps), ptr(ps.dirs),
                              ptr(ps.depth), ptr(ps.inc), ptr(ps.lmask), ptr(ps.scan_id), ptr(poses), ns, kind, nt,
                              ptr(w), ptr(e), int(want_exponent), int(want_pose), ptr(part), ptr(out), stream_ptr()),
          'dc_points_bwd')
    return _grads_split(out, nt, ns)


# ------------------------------------------------------------------------------------------------
# neighbourhood features
# ------------------------------------------------------------------------------------------------
@on_device
def features_fwd(points, nbr, dirs=None, mean_weights=None, scale=None, want=('mean', 'cov', 'eigvals', 'eigvecs',
                                                                             'normals', 'inc_angles'),
                 want_saved=False, want_weights=False):
    need(points, (None, None), name='points')
    n, stride = points.shape
    assert stride in (3, 4)
    dev, dt = points.device, points.dtype
    need(nbr, (n, None), dtype=torch.int32, name='neighbors', device=dev)
    k = nbr.shape[1]
    if dirs is not None:
        need(dirs, (n, 3), dtype=dt, name='dirs', device=dev)
    if mean_weights is not None:
        mean_weights = mean_weights.reshape(n, k)
        need(mean_weights, (n, k), dtype=dt, name='weights', device=dev)
    shapes = dict(mean=(n, 3), cov=(n, 3, 3), eigvals=(n, 3), eigvecs=(n, 3, 3), normals=(n, 3), inc_angles=(n, 1))
    out = {f: (torch.empty(shapes[f], dtype=dt, device=dev) if f in want else None) for f in shapes}
    if (out['normals'] is not None or out['inc_angles'] is not None) and dirs is None:
        raise ValueError('normals / incidence angles need dirs')
    nvalid = torch.empty((n,), dtype=torch.int32, device=dev) if want_saved else None
    cmean = torch.empty((n, 3), dtype=dt, device=dev) if want_saved else None
    invd = torch.empty((n,), dtype=dt, device=dev) if want_saved else None
    wout = torch.empty((n, k), dtype=dt, device=dev) if want_weights else None
    check(lib().dc_features_fwd(ptr(points), stride, dtype_code(points), ptr(nbr), n, k, ptr(mean_weights),
                                float(scale) if scale else 0.0, ptr(dirs), ptr(out['mean']), ptr(out['cov']),
                                ptr(out['eigvals']), ptr(out['eigvecs']), ptr(out['normals']), ptr(out['inc_angles']),
                                ptr(nvalid), ptr(wout), ptr(cmean), ptr(invd), stream_ptr()), 'dc_features_fwd')
    out.update(nvalid=nvalid, cmean=cmean, invd=invd, weights=wout)
    return out


@on_device
def features_bwd(points, csr_ptr, csr_src, cmean, invd, nvalid, eigvecs=None, grad_mean=None, grad_cov=None,
                 grad_eigvals=None):
    need(points, (None, None), name='points')
    n, stride = points.shape
    dev, dt = points.device, points.dtype
    need(csr_ptr, (n + 1,), dtype=torch.int32, name='csr_ptr', device=dev)
    need(csr_src, (None,), dtype=torch.int32, name='csr_src', device=dev)
    need(cmean, (n, 3), dtype=dt, name='cmean', device=dev)
    need(invd, (n,), dtype=dt, name='invd', device=dev)
    need(nvalid, (n,), dtype=torch.int32, name='nvalid', device=dev)
    for t, shp, nm in ((eigvecs, (n, 3, 3), 'eigvecs'), (grad_mean, (n, 3), 'grad_mean'), (grad_cov, (n, 3, 3), 'grad_cov'),
                       (grad_eigvals, (n, 3), 'grad_eigvals')):
        if t is not None:
            need(t, shp, dtype=dt, name=nm, device=dev)
    grec = torch.empty((n, 12), dtype=dt, device=dev)
    gp = torch.empty((n, stride), dtype=dt, device=dev)
    check(lib().dc_features_bwd(ptr(points), stride, dtype_code(points), ptr(csr_ptr), ptr(csr_src), n, ptr(cmean),
                                ptr(invd), ptr(nvalid), ptr(eigvecs), ptr(grad_mean), ptr(grad_cov), ptr(grad_eigvals),
                                ptr(grec), ptr(gp), stream_ptr()), 'dc_features_bwd')
    return gp


# ------------------------------------------------------------------------------------------------
# fused consistency loss
# ------------------------------------------------------------------------------------------------
@on_device
def consistency_fwd(points, nbr, mask=None, offset=None, loss='min_eigval_loss', normalization=True, sqrt=False,
                    rec=None, want_pointwise=False, want_eigvals=False, partials=None, sums=None, qfmt=None,
                    centre_idx=None, table=None, raw_pointwise=False):
    """raw_pointwise: the pointwise output holds the loss before relu / sqrt (what the quantile-inlier gating of
    loss.py:256-277 compares; see consistency_gate)."""
    _check_points(points, qfmt)
    n_points, stride = points.shape
    dev = points.device
    dt = torch.float32 if qfmt is not None else points.dtype
    if centre_idx is not None:               # compact centre list: rows of nbr / rec / outputs follow it
        need(centre_idx, (None,), dtype=torch.int32, name='centre_idx', device=dev)
        n = centre_idx.shape[0]
        assert n == 0 or (int(centre_idx.min()) >= 0 and int(centre_idx.max()) < n_points)
    else:
        n = n_points
    need(nbr, (n, None), dtype=torch.int32, name='neighbors', device=dev)
    k = nbr.shape[1]
    if mask is not None:
        need(mask, (n,), dtype=torch.bool, name='mask', device=dev)
    if offset is not None:
        need(offset, (n,), dtype=dt, name='offset', device=dev)
    if rec is None:
        rec = torch.empty((n, 8), dtype=points.dtype, device=dev)
    else:
        need(rec, (n, 8), dtype=points.dtype, name='rec', device=dev)
    fmt, qptr = _fmt_args(dt, qfmt)
    pw = torch.empty((n,), dtype=dt, device=dev) if want_pointwise else None
    ev = torch.empty((n, 3), dtype=dt, device=dev) if want_eigvals else None
    rows = lib().dc_partial_rows(n)
    if partials is None:
        partials = torch.empty((rows * 2,), dtype=torch.float64, device=dev)
    else:
        need(partials, (None,), dtype=torch.float64, name='partials', device=dev)
        assert partials.numel() >= rows * 2
    if sums is None:
        sums = torch.empty((2,), dtype=torch.float64, device=dev)
    check(lib().dc_consistency_fwd(ptr(points), stride, fmt if qfmt is None else nv.DC_F32, fmt, qptr, ptr(nbr),
                                   ptr(centre_idx), _table_ref(table, n), n, k,
                                   ptr(mask), ptr(offset), nv.LOSS_KINDS[loss] | (nv.LOSS_RAW_POINTWISE if raw_pointwise else 0),
                                   int(bool(normalization)), int(bool(sqrt)), ptr(rec), ptr(pw),
                                   ptr(ev), ptr(partials), ptr(sums), stream_ptr()), 'dc_consistency_fwd')
    return dict(sums=sums, rec=rec, pointwise=pw, eigvals=ev)


@on_device
def consistency_gate(raw, rec, threshold, mask=None, sqrt=False, partials=None, sums=None, qfmt=None):
    """Quantile-inlier gating of a fused forward (loss.py:256-277): masked centres whose raw loss exceeds the device scalar
    ``threshold`` stop contributing -- their records' coefficients are zeroed in place -- and ``sums`` becomes
    {sum of the inliers' loss, number of inliers}."""
    n = raw.shape[0]
    dev = raw.device
    need(rec, (n, 8), name='rec', device=dev)
    need(threshold, (), dtype=torch.float64, name='threshold', device=dev)
    if mask is not None:
        need(mask, (n,), dtype=torch.bool, name='mask', device=dev)
    fmt, _ = _fmt_args(raw.dtype, qfmt)
    rows = lib().dc_partial_rows(n)
    if partials is None:
        partials = torch.empty((rows * 2,), dtype=torch.float64, device=dev)
    if sums is None:
        sums = torch.empty((2,), dtype=torch.float64, device=dev)
    check(lib().dc_consistency_gate(ptr(raw), nv.DC_F32 if raw.dtype == torch.float32 else nv.DC_F64, fmt, ptr(mask), n,
                                    ptr(threshold), int(bool(sqrt)), ptr(rec), ptr(partials), ptr(sums), stream_ptr()),
          'dc_consistency_gate')
    return sums


def degree_lane_perm(csr_ptr, block=256):
    """u8 [block * ceil(n / block)]: per block of consecutive points the lane -> point map by ascending in-degree."""
    n = csr_ptr.shape[0] - 1
    nb = (n + block - 1) // block
    deg = torch.full((nb * block,), 2 ** 30, dtype=torch.int32, device=csr_ptr.device)
    deg[:n] = csr_ptr[1:] - csr_ptr[:-1]
    return torch.argsort(deg.reshape(nb, block), dim=1, stable=True).to(torch.uint8).reshape(-1).contiguous()


@on_device
def consistency_bwd(points, rec, csr_ptr, csr_src, ps=None, poses=None, model_kind=None, w=None, e=None,
                    want_exponent=False, want_pose=False, want_grad_points=False, partials=None, grads=None, qfmt=None,
                    lane_perm=None, table=None):
    _check_points(points, qfmt)
    n, stride = points.shape
    dev = points.device
    dt = torch.float32 if qfmt is not None else points.dtype
    fmt, qptr = _fmt_args(dt, qfmt)
    dcode = nv.DC_F32 if qfmt is not None else fmt
    need(rec, (None, 8), dtype=points.dtype, name='rec', device=dev)        # one row per centre (<= n with a centre list)
    need(csr_ptr, (n + 1,), dtype=torch.int32, name='csr_ptr', device=dev)
    need(csr_src, (None,), dtype=torch.int32, name='csr_src', device=dev)
    gp = torch.empty((n, stride), dtype=dt, device=dev) if want_grad_points else None
    if lane_perm is not None:
        need(lane_perm, (((n + 255) // 256) * 256,), dtype=torch.uint8, name='lane_perm', device=dev)
    if ps is None:
        assert want_grad_points
        check(lib().dc_consistency_bwd(ptr(points), stride, dcode, fmt, qptr, ptr(rec), ptr(csr_ptr), ptr(csr_src),
                                       ptr(lane_perm), _table_ref(table, n), n, None, None, None, None, None, None, None, 0, 0, 0,
                                       None, None, 0, 0, ptr(gp), None,
                                       None, stream_ptr()), 'dc_consistency_bwd')
        return gp, None
    assert ps.n == n and ps.dtype == dt and ps.device == dev
    kind, nt, w, e = _model_args(model_kind, w, e, ps)
    poses, ns = _pose_args(poses, ps)
    nacc = 2 * nt + 12 * ns
    rows = lib().dc_partial_rows(n)
    if partials is None:
        partials = torch.empty((max(rows * nacc, 1),), dtype=torch.float64, device=dev)
    else:
        assert partials.numel() >= rows * nacc
    if grads is None:
        grads = torch.zeros((max(nacc, 1),), dtype=torch.float64, device=dev)
    check(lib().dc_consistency_bwd(ptr(points), stride, dcode, fmt, qptr, ptr(rec), ptr(csr_ptr), ptr(csr_src),
                                   ptr(lane_perm), _table_ref(table, n), n, ptr(ps.vps), ptr(ps.dirs), ptr(ps.depth), ptr(ps.inc), ptr(ps.lmask), ptr(ps.scan_id),
                                   ptr(poses), ns, kind, nt, ptr(w), ptr(e), int(want_exponent), int(want_pose), ptr(gp),
                                   ptr(partials), ptr(grads), stream_ptr()), 'dc_consistency_bwd')
    return gp, _grads_split(grads, nt, ns)


# ------------------------------------------------------------------------------------------------
# masks
# ------------------------------------------------------------------------------------------------
def _bound(v, default):
    if v is None:
        return default
    v = float(v)
    return default if v != v else v        # config files encode "unbounded" as nan (config.py:41-43)


@on_device
def mask_bounds(mask, num, num_index=0, den=None, den_index=0, lo=None, hi=None):
    """mask &= lo <= num[:, num_index] (/ den[:, den_index]) <= hi   (in place, returns mask)."""
    n = mask.shape[0]
    need(mask, (n,), dtype=torch.bool, name='mask')
    num2 = num.reshape(n, -1)
    need(num2, (n, None), name='values', device=mask.device)
    den2 = None
    if den is not None:
        den2 = den.reshape(n, -1)
        need(den2, (n, None), dtype=num2.dtype, name='denominator', device=mask.device)
    check(lib().dc_mask_bounds(ptr(num2), num2.shape[1], num_index, ptr(den2), den2.shape[1] if den2 is not None else 0,
                               den_index, dtype_code(num2), n, _bound(lo, float('-inf')), _bound(hi, float('inf')),
                               ptr(mask), stream_ptr()), 'dc_mask_bounds')
    return mask


@on_device
def mask_bounds_all(values, bounds, mask=None):
    """bool [N]: every ``(num_index, den_index | None, lo, hi)`` of ``bounds`` on the columns of ``values`` [N, C] in ONE pass
    (dc_mask_bounds_multi), ANDed into ``mask`` when given (in place), else into a new mask."""
    n = values.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.bool, device=values.device) if mask is None else mask
    v2 = values.reshape(n, -1)
    need(v2, (n, None), name='values')
    init = mask is None
    if init:
        mask = torch.empty((n,), dtype=torch.bool, device=v2.device)
    need(mask, (n,), dtype=torch.bool, name='mask', device=v2.device)
    nb = len(bounds)
    if nb > 8:
        raise ValueError('at most 8 bounds per pass')
    num = (ctypes.c_int32 * max(nb, 1))(*[int(b[0]) for b in bounds])
    den = (ctypes.c_int32 * max(nb, 1))(*[-1 if b[1] is None else int(b[1]) for b in bounds])
    lo = (ctypes.c_double * max(nb, 1))(*[_bound(b[2], float('-inf')) for b in bounds])
    hi = (ctypes.c_double * max(nb, 1))(*[_bound(b[3], float('inf')) for b in bounds])
    check(lib().dc_mask_bounds_multi(ptr(v2), v2.shape[1], dtype_code(v2), n, nb, num, den, lo, hi, 1 if init else 0, ptr(mask),
                                     stream_ptr()), 'dc_mask_bounds_multi')
    return mask


@on_device
def compact_rows(mask, fields, want_index=False):
    """``[f[mask] for f in fields]`` (and the kept row numbers, int32) for up to 8 contiguous arrays of N rows on the device: two
    launches and ONE synchronisation (the number of kept rows) for all of them (dc_compact_rows; depth_cloud.py:126-134)."""
    n = mask.shape[0]
    need(mask, (n,), dtype=torch.bool, name='mask')
    dev = mask.device
    if len(fields) > 8:
        raise ValueError('at most 8 fields per call')
    srcs, outs, rb = [], [], []
    for f in fields:
        if f.shape[0] != n or f.device != dev:
            raise ValueError('every field needs %d rows on %s' % (n, dev))
        f = f.contiguous()
        srcs.append(f)
        outs.append(torch.empty_like(f))
        rb.append(f.element_size() * (f.numel() // n if n else 1))
    index = torch.empty((n,), dtype=torch.int32, device=dev) if want_index else None
    count = torch.empty((1,), dtype=torch.int64, device=dev)
    nf = len(srcs)
    a_src = (ctypes.c_void_p * max(nf, 1))(*[ptr(f) for f in srcs])
    a_dst = (ctypes.c_void_p * max(nf, 1))(*[ptr(f) for f in outs])
    a_rb = (ctypes.c_int32 * max(nf, 1))(*rb)
    nbytes = lib().dc_compact_rows_workspace_bytes(n)
    ws = _ws(nbytes, dev)
    check(lib().dc_compact_rows(ptr(mask), n, nf, a_src, a_dst, a_rb, ptr(index), ptr(count), ptr(ws), nbytes, stream_ptr()),
          'dc_compact_rows')
    m = int(count.item())
    outs = [o[:m] for o in outs]
    return (outs, index[:m]) if want_index else outs


@on_device
def scan_prefilter(points, vps, dtype, r, lo, hi):
    """Raw rows [N, >=3] (and viewpoints [N,3] | None) on the device -> (vps, dirs, depth [M,1], points) of the rays the scan-shadow
    filter keeps: from_points, to_points, dc_shadow_filter and cloud[mask] launched by ONE call (dc_scan_prefilter), one
    synchronisation (M)."""
    need(points, (None, None), name='points')
    n, stride = points.shape
    if stride < 3:
        raise ValueError('points need at least 3 columns')
    dev = points.device
    dtype = points.dtype if dtype is None else dtype
    if vps is not None:
        need(vps, (n, 3), dtype=points.dtype, name='vps', device=dev)
    outs = [torch.empty((n, c), dtype=dtype, device=dev) for c in (3, 3, 1, 3)]
    count = torch.empty((1,), dtype=torch.int64, device=dev)
    nbytes = lib().dc_scan_prefilter_workspace_bytes(n)
    ws = _ws(nbytes, dev)
    check(lib().dc_scan_prefilter(ptr(points), stride, dtype_code(points), ptr(vps), n, nv.DC_F32 if dtype == torch.float32 else nv.DC_F64,
                                  float(r), float(lo), float(hi), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), ptr(count),
                                  ptr(ws), nbytes, stream_ptr()), 'dc_scan_prefilter')
    m = int(count.item())
    return tuple(o[:m] for o in outs)


@on_device
def to_points(vps, dirs, depth):
    """``vps + depth * dirs`` in one kernel, bit-equal to the two tensor operations (dc_to_points; no autograd)."""
    n = dirs.shape[0]
    need(dirs, (n, 3), name='dirs')
    need(depth, (n, 1), dtype=dirs.dtype, name='depth', device=dirs.device)
    vps = vps.reshape(-1, 3)
    need(vps, (None, 3), dtype=dirs.dtype, name='vps', device=dirs.device)
    if vps.shape[0] not in (1, n):
        raise ValueError('vps must have 1 or %d rows' % n)
    out = torch.empty_like(dirs)
    check(lib().dc_to_points(ptr(vps), vps.shape[0], ptr(dirs), ptr(depth), dtype_code(dirs), n, ptr(out), stream_ptr()), 'dc_to_points')
    return out


@on_device
def valid_weights(nbr):
    """float32 [N, K, 1]: 1 where the table holds a neighbour (``valid_neighbor_mask().float()[..., None]``)."""
    need(nbr, (None, None), dtype=torch.int32, name='neighbors')
    out = torch.empty(tuple(nbr.shape) + (1,), dtype=torch.float32, device=nbr.device)
    check(lib().dc_valid_weights(ptr(nbr), nbr.numel(), ptr(out), stream_ptr()), 'dc_valid_weights')
    return out


@on_device
def valid_count(nbr):
    need(nbr, (None, None), dtype=torch.int32, name='neighbors')
    cnt = torch.empty((nbr.shape[0],), dtype=torch.int32, device=nbr.device)
    check(lib().dc_valid_count(ptr(nbr), nbr.shape[0], nbr.shape[1], ptr(cnt), stream_ptr()), 'dc_valid_count')
    return cnt


@on_device
def dispersion(vec, nbr, weights=None):
    need(vec, (None, 3), name='vectors')
    n = vec.shape[0]
    need(nbr, (n, None), dtype=torch.int32, name='neighbors', device=vec.device)
    k = nbr.shape[1]
    if weights is not None:
        weights = weights.reshape(n, k)
        need(weights, (n, k), dtype=vec.dtype, name='weights', device=vec.device)
    out = torch.empty((n,), dtype=vec.dtype, device=vec.device)
    check(lib().dc_dispersion(ptr(vec), dtype_code(vec), ptr(nbr), ptr(weights), n, k, ptr(out), stream_ptr()),
          'dc_dispersion')
    return out


@on_device
def cloud_from_points(points, vps=None, dtype=None, ego_box=None, min_depth=None, max_depth=None, want_index=False, want_zero_vps=False):
    """Raw rows [N, >=3] on the device -> (vps [M,3] | None, dirs [M,3], depth [M,1], index int64 [M] | None) of the rows
    that survive the ego-box crop and the depth bounds, in their original order (dc_cloud_from_points: kitti360.py:101-105,
    filters.py:116-141, depth_cloud.py:592-638).  One synchronisation (the number of kept rows) when a filter is active."""
    need(points, (None, None), name='points')
    n, stride = points.shape
    if stride < 3:
        raise ValueError('points need at least 3 columns')
    dev = points.device
    dtype = points.dtype if dtype is None else dtype
    if vps is not None:
        need(vps, (n, 3), dtype=points.dtype, name='vps', device=dev)
    dirs = torch.empty((n, 3), dtype=dtype, device=dev)
    depth = torch.empty((n, 1), dtype=dtype, device=dev)
    filtered = bool(ego_box and ego_box > 0) or min_depth is not None or max_depth is not None
    # (want_zero_vps: the kernel that writes the fields also writes the zero viewpoints of a cloud without any)
    vps_out = torch.empty((n, 3), dtype=dtype, device=dev) if (vps is not None or want_zero_vps) else None
    index = torch.empty((n,), dtype=torch.int32, device=dev) if want_index else None
    count = torch.empty((1,), dtype=torch.int64, device=dev)
    nbytes = lib().dc_cloud_from_points_workspace_bytes(n)
    ws = _ws(nbytes, dev)
    nan = float('nan')
    check(lib().dc_cloud_from_points(ptr(points), stride, dtype_code(points), ptr(vps), n, float(ego_box or 0.0),
                                     nan if min_depth is None else float(min_depth), nan if max_depth is None else float(max_depth),
                                     nv.DC_F32 if dtype == torch.float32 else nv.DC_F64, ptr(dirs), ptr(depth), ptr(vps_out),
                                     ptr(index), ptr(count), ptr(ws), nbytes, stream_ptr()), 'dc_cloud_from_points')
    m = int(count.item()) if filtered else n
    return (None if vps_out is None else vps_out[:m], dirs[:m], depth[:m], None if index is None else index[:m].long())


@on_device
def correct_depth(depth, gamma, mask, w, exponent, op):
    """depth' [N,1]: the polynomial models outside the training loop (dc_correct_depth; op 0 d-b, 1 d+b, 2 d(1-b), 3 d/(1-b));
    ``mask`` bool [N] or None; ``w`` / ``exponent`` fp64 [1,P] on the device.  No autograd."""
    n = depth.shape[0]
    dev = depth.device
    need(depth, (n, 1), name='depth')
    need(gamma, (n, 1), dtype=depth.dtype, name='inc_angles', device=dev)
    w1, e1 = w.detach().reshape(-1).contiguous(), exponent.detach().reshape(-1).contiguous()
    need(w1, (None,), dtype=torch.float64, name='w', device=dev)
    need(e1, (w1.shape[0],), dtype=torch.float64, name='exponent', device=dev)
    if mask is not None:
        need(mask, (n,), dtype=torch.bool, name='mask', device=dev)
    out = torch.empty_like(depth)
    check(lib().dc_correct_depth(ptr(depth), ptr(gamma), ptr(mask), ptr(w1), ptr(e1), w1.shape[0], int(op), dtype_code(depth), n,
                                 ptr(out), stream_ptr()), 'dc_correct_depth')
    return out


@on_device
def shadow_filter(points, vps, dirs, r, lo, hi):
    """bool [N]: ``shadow_mask`` over the direction neighbourhoods of radius ``r`` (chord length) WITHOUT building their table
    (dc_shadow_filter: grid over ``dirs`` + one walk); the same mask as radius_neighbors(dirs, r) -> shadow_mask."""
    need(points, (None, 3), name='points')
    n = points.shape[0]
    dev, dt = points.device, points.dtype
    vps = vps.reshape(-1, 3)
    need(vps, (None, 3), dtype=dt, name='vps', device=dev)
    need(dirs, (n, 3), dtype=dt, name='dirs', device=dev)
    if vps.shape[0] not in (1, n):
        raise ValueError('vps must have 1 or %d rows' % n)
    mask = torch.empty((n,), dtype=torch.bool, device=dev)
    if n:
        nbytes = lib().dc_knn_workspace_bytes(n, 0)
        ws = _ws(nbytes, dev)
        check(lib().dc_shadow_filter(ptr(points), ptr(vps), vps.shape[0], ptr(dirs), dtype_code(points), n, float(r), float(lo), float(hi),
                                     ptr(mask), ptr(ws), nbytes, stream_ptr()), 'dc_shadow_filter')
    return mask


@on_device
def shadow_mask(points, vps, dir_nbr, lo, hi, fill):
    """bool [N]: every angle between (vps - x) and (x_j - x), j in dir_nbr[i], within [lo, hi] (filters.py:257-309)."""
    need(points, (None, 3), name='points')
    n = points.shape[0]
    dev, dt = points.device, points.dtype
    vps = vps.reshape(-1, 3)
    need(vps, (None, 3), dtype=dt, name='vps', device=dev)
    if vps.shape[0] not in (1, n):
        raise ValueError('vps must have 1 or %d rows' % n)
    need(dir_nbr, (n, None), dtype=torch.int32, name='dir_neighbors', device=dev)
    mask = torch.empty((n,), dtype=torch.bool, device=dev)
    if n and dir_nbr.shape[1]:
        check(lib().dc_shadow_mask(ptr(points), ptr(vps), vps.shape[0], dtype_code(points), ptr(dir_nbr), n, dir_nbr.shape[1],
                                   float(lo), float(hi), float(fill), ptr(mask), stream_ptr()), 'dc_shadow_mask')
    else:
        mask.fill_(True)
    return mask


@on_device
def voxel_filter(points, grid_res, seq=None, preserve_order=False):
    """Indices (int64, device) of one survivor per voxel with filter_grid's dict semantics; None if the voxel range
    does not fit the 63-bit key (caller falls back to the host algorithm)."""
    need(points, (None, 3), name='points')
    n = points.shape[0]
    dev = points.device
    if seq is not None:
        need(seq, (n,), dtype=torch.int32, name='seq', device=dev)
    out = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    meta = torch.zeros((2,), dtype=torch.int32, device=dev)
    nbytes = lib().dc_voxel_filter_workspace_bytes(n)
    ws = _ws(nbytes, dev)
    check(lib().dc_voxel_filter(ptr(points), 3, dtype_code(points), n, float(grid_res), ptr(seq), int(bool(preserve_order)),
                                ptr(out), ctypes.c_void_p(meta.data_ptr()), ctypes.c_void_p(meta.data_ptr() + 4), ptr(ws),
                                nbytes, stream_ptr()), 'dc_voxel_filter')
    count, status = meta.tolist()
    return None if status else out[:count].long()


# ------------------------------------------------------------------------------------------------
# point-to-plane ICP pair
# ------------------------------------------------------------------------------------------------
@on_device
def p2plane_pair(psa, normals_a, psb, normals_b, pose_a, pose_b, idx_a, idx_b, model_kind=None, w=None, e=None):
    """Sums of point-to-plane distances over the correspondences of one scan pair and their gradients:
    returns (sums f64 [2], dw [P], de [P], dTa [3,4], dTb [3,4]) for d(sum12 + sum21)."""
    dev, dt = psa.device, psa.dtype
    assert psb.device == dev and psb.dtype == dt
    need(normals_a, (psa.n, 3), dtype=dt, name='normals_a', device=dev)
    need(normals_b, (psb.n, 3), dtype=dt, name='normals_b', device=dev)
    need(pose_a, (12,), dtype=torch.float64, name='pose_a', device=dev)
    need(pose_b, (12,), dtype=torch.float64, name='pose_b', device=dev)
    need(idx_a, (None,), dtype=torch.int32, name='idx_a', device=dev)
    m = idx_a.shape[0]
    need(idx_b, (m,), dtype=torch.int32, name='idx_b', device=dev)
    kind, nt, w, e = _model_args(model_kind, w, e, psa)
    if kind != 0 and psb.inc is None:
        raise ValueError('the model needs incidence angles')
    part = torch.empty((lib().dc_p2plane_partial_count(m),), dtype=torch.float64, device=dev)
    out = torch.empty((2 + 2 * nt + 24,), dtype=torch.float64, device=dev)
    check(lib().dc_p2plane_pair(ptr(psa.vps), ptr(psa.dirs), ptr(psa.depth), ptr(psa.inc), ptr(psa.lmask),
                                ptr(normals_a), ptr(psb.vps), ptr(psb.dirs), ptr(psb.depth), ptr(psb.inc),
                                ptr(psb.lmask), ptr(normals_b), dtype_code(psa.dirs), ptr(pose_a), ptr(pose_b), kind, nt,
                                ptr(w), ptr(e), ptr(idx_a), ptr(idx_b), m, 1, 1, ptr(part), ptr(out), stream_ptr()),
          'dc_p2plane_pair')
    return out[:2], out[2:2 + nt], out[2 + nt:2 + 2 * nt], out[2 + 2 * nt:14 + 2 * nt].reshape(3, 4), \
        out[14 + 2 * nt:].reshape(3, 4)


@on_device
def p2point_pair(psa, psb, pose_a, pose_b, idx_a, idx_b, model_kind=None, w=None, e=None):
    """Sum of point-to-point distances over the correspondences of one scan pair and its gradients:
    returns (sums f64 [2] = {sum |xb - xa|, 0}, dw [P], de [P], dTa [3,4], dTb [3,4])."""
    dev, dt = psa.device, psa.dtype
    assert psb.device == dev and psb.dtype == dt
    need(pose_a, (12,), dtype=torch.float64, name='pose_a', device=dev)
    need(pose_b, (12,), dtype=torch.float64, name='pose_b', device=dev)
    need(idx_a, (None,), dtype=torch.int32, name='idx_a', device=dev)
    m = idx_a.shape[0]
    need(idx_b, (m,), dtype=torch.int32, name='idx_b', device=dev)
    kind, nt, w, e = _model_args(model_kind, w, e, psa)
    if kind != 0 and psb.inc is None:
        raise ValueError('the model needs incidence angles')
    part = torch.empty((lib().dc_p2plane_partial_count(m),), dtype=torch.float64, device=dev)
    out = torch.empty((2 + 2 * nt + 24,), dtype=torch.float64, device=dev)
    check(lib().dc_p2point_pair(ptr(psa.vps), ptr(psa.dirs), ptr(psa.depth), ptr(psa.inc), ptr(psa.lmask), ptr(psb.vps),
                                ptr(psb.dirs), ptr(psb.depth), ptr(psb.inc), ptr(psb.lmask), dtype_code(psa.dirs), ptr(pose_a),
                                ptr(pose_b), kind, nt, ptr(w), ptr(e), ptr(idx_a), ptr(idx_b), m, ptr(part), ptr(out),
                                stream_ptr()), 'dc_p2point_pair')
    return out[:2], out[2:2 + nt], out[2 + nt:2 + 2 * nt], out[2 + 2 * nt:14 + 2 * nt].reshape(3, 4), \
        out[14 + 2 * nt:].reshape(3, 4)


class IcpSequence:
    """Descriptors of one sequence of scans and its pair correspondences for dc_p2plane_sequence / dc_p2point_sequence,
    filled once; ``eval`` is then one host call per evaluation.  ``scans``: list of (PointSet, normals [n,3] | None);
    ``pairs``: list of (scan_a, scan_b, idx_a int32 [m], idx_b int32 [m]); pair weights follow icp_loss
    (loss.py:391-403): 0.5 / (m * n_pairs) for point to plane (two directed sums), 1 / (m * n_pairs) for point to
    point (``plane=False``, loss.py:553,563)."""

    @on_device
    def __init__(self, scans, pairs, with_model=True, plane=True):
        ps0 = scans[0][0]
        self.device, self.dtype, self.n_scans = ps0.device, ps0.dtype, len(scans)
        self.with_model, self.plane = bool(with_model), bool(plane)
        self._keep = []
        self.scan_desc = (nv.IcpScan * max(len(scans), 1))()
        for d, (ps, normals) in zip(self.scan_desc, scans):
            assert ps.device == self.device and ps.dtype == self.dtype
            if plane:
                need(normals, (ps.n, 3), dtype=self.dtype, name='normals', device=self.device)
            else:
                normals = None
            if with_model and ps.inc is None:
                raise ValueError('the model needs incidence angles')
            self._keep.append((ps, normals))
            d.vps, d.dirs, d.depth, d.inc = ptr(ps.vps), ptr(ps.dirs), ptr(ps.depth), ptr(ps.inc)
            d.lmask, d.normals = ptr(ps.lmask), ptr(normals)
        self.pair_desc = (nv.IcpPair * max(len(pairs), 1))()
        self.n_pairs, max_m = len(pairs), 0
        for d, (a, b, idx_a, idx_b) in zip(self.pair_desc, pairs):
            need(idx_a, (None,), dtype=torch.int32, name='idx_a', device=self.device)
            m = idx_a.shape[0]
            need(idx_b, (m,), dtype=torch.int32, name='idx_b', device=self.device)
            if not (0 <= a < self.n_scans and 0 <= b < self.n_scans and a != b):
                raise ValueError('pair (%d, %d) outside the sequence' % (a, b))
            self._keep.append((idx_a, idx_b))
            d.scan_a, d.scan_b, d.idx_a, d.idx_b, d.m = int(a), int(b), ptr(idx_a), ptr(idx_b), m
            d.weight = (0.5 if plane else 1.0) / (max(m, 1) * len(pairs))
            max_m = max(max_m, m)
        self.part = torch.empty((lib().dc_p2plane_sequence_partial_count(ctypes.cast(self.pair_desc, ctypes.c_void_p), self.n_pairs),),
                                dtype=torch.float64, device=self.device)
        self.ticket = torch.zeros((32,), dtype=torch.int32, device=self.device)      # dc_icp_sequence_step's (left zero by every launch)

    @on_device
    def eval(self, poses12, model_kind=None, w=None, e=None, out=None):
        """out fp64 [1 + 2P + 12 S] = {loss, dloss/dw, dloss/dexponent, dloss/d[R|t]}."""
        kind, nt, w, e = _model_args(model_kind, w, e, self._keep[0][0]) if self.n_scans else (0, 0, None, None)
        if kind != 0 and not self.with_model:
            raise ValueError('sequence was built without incidence angles')
        need(poses12, (self.n_scans, 12), dtype=torch.float64, name='poses[S,12]', device=self.device)
        n_out = 1 + 2 * nt + 12 * self.n_scans
        if out is None:
            out = torch.empty((n_out,), dtype=torch.float64, device=self.device)
        else:
            need(out, (n_out,), dtype=torch.float64, name='out', device=self.device)
        fn = lib().dc_p2plane_sequence if self.plane else lib().dc_p2point_sequence
        check(fn(ctypes.cast(self.scan_desc, ctypes.c_void_p), self.n_scans, ctypes.cast(self.pair_desc, ctypes.c_void_p),
                 self.n_pairs, 0 if self.dtype == torch.float32 else 1, ptr(poses12), kind, nt, ptr(w), ptr(e),
                 ptr(self.part), ptr(out), stream_ptr()), 'dc_p2plane_sequence' if self.plane else 'dc_p2point_sequence')
        return out

    @on_device
    def step(self, poses12, model_kind, w, e, out, fin=None):
        """``eval`` as ONE launch (dc_icp_sequence_step: the sums are finished by the block that takes the last ticket) and, with
        ``fin`` (a _native.PoseTrainStepDesc), the finishing step of a training iteration -- dc_pose_train_finish's work -- by
        that same block.  Returns False when the sequence cannot take it (more than sixteen pairs, no correspondence at all)."""
        kind, nt, w, e = _model_args(model_kind, w, e, self._keep[0][0]) if self.n_scans else (0, 0, None, None)
        if kind != 0 and not self.with_model:
            raise ValueError('sequence was built without incidence angles')
        need(poses12, (self.n_scans, 12), dtype=torch.float64, name='poses[S,12]', device=self.device)
        need(out, (1 + 2 * nt + 12 * self.n_scans,), dtype=torch.float64, name='out', device=self.device)
        rc = lib().dc_icp_sequence_step(1 if self.plane else 0, ctypes.cast(self.scan_desc, ctypes.c_void_p), self.n_scans,
                                        ctypes.cast(self.pair_desc, ctypes.c_void_p), self.n_pairs, 0 if self.dtype == torch.float32 else 1,
                                        ptr(poses12), kind, nt, ptr(w), ptr(e), ptr(self.part), ptr(out), ptr(self.ticket),
                                        None if fin is None else ctypes.byref(fin), stream_ptr())
        if rc == nv.DC_ERR_UNSUPPORTED:
            return False
        check(rc, 'dc_icp_sequence_step')
        return True
