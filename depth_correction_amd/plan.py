"""Fused per-sequence map-consistency evaluation (the MI355X hot path).

One ``SequencePlan`` holds everything that is constant over the optimisation of one sequence
(train.py:94-215 builds exactly these once: local feature clouds, global neighbourhoods, masks) laid out
for the GPU, and evaluates one training iteration (eval.py:85-112: model -> pose transform -> concatenate
-> neighbourhood features -> loss, plus its backward) in three kernels:

    dc_points_fwd       corrected, posed points of all scans            (K1-K3)
    dc_consistency_fwd  covariance, smallest eigenpair, loss, bwd record (K5-K15)
    dc_consistency_bwd  dL/dx gather + dL/dw, dL/dexponent, dL/dpose     (K18)

or -- whenever only the model weights are differentiated, i.e. poses and exponents are constants of the loop -- in ONE:
the basis form x = X0 + (sum_k w_k c_k) u (dc_points_basis, rebuilt when poses / exponents change) needs no pass over the
points, and for up to three weights the kernel that computes the loss also finishes dL/dw by a second sweep over each
centre's own neighbours (consistency_step_basis_kernel).

Layout decisions:
  * points are permuted once into Morton order of the initial global cloud, so the K neighbours of
    consecutive lanes share cache lines (the reference's scan-major order has no locality at all);
  * float32 clouds keep their intermediate points / means as 32-bit fixed point (DC_Q32), float64 clouds
    as float64 -- on-chip arithmetic is fp64 either way;
  * the backward runs over the transposed neighbour list (no atomics, bitwise reproducible).
Results are reported in the caller's point order; per-point outputs are un-permuted lazily.
"""
from __future__ import annotations

import ctypes

import os

import numpy as np

import torch

from . import ops
from . import _native as nv
from ._native import need, lib, check, ptr, stream_ptr, on_device

__all__ = ['SequencePlan', 'SequenceTrainer', 'consistency_loss', 'KernelTimer', 'PlanRegistry']


class SequencePlan:
    kMaskGroup = 2048          # super-block inside which masked-in points are put before masked-out ones (mask_first)

    @on_device
    def __init__(self, clouds, poses, neighbors, mask=None, model_kind='ScaledPolynomial', loss='min_eigval_loss',
                 normalization=True, sqrt=False, spatial_sort=True, point_format='auto', degree_sort=False,
                 active_only=False, block_tables=True, bwd_layout='runs', stages=None, basis=True, lazy_backward=True,
                 scan_group=True, mask_first=False, nan_policy=None, pose_kernel=True, degree_group=False, heavy_first=False,
                 wave_pack=None):
        """
        :param clouds: per-scan dicts / objects with vps [n,3], dirs [n,3], depth [n,1], inc_angles [n,1], mask [n]
                       (local feature clouds, sensor frame), GPU tensors of one dtype.
        :param poses: [S,4,4] initial scan poses (used for the layout only; every evaluation takes its own).
        :param neighbors: [N,K] int neighbour indices of the global cloud (scan-major order, -1 = missing).
        :param mask: [N] bool global mask (None = all points).
        :param block_tables: build the block tables that let both hot kernels gather from LDS (ops.block_table).
        :param bwd_layout: 'runs' (per-point runs padded to four positions) or 'slots' (slot-major, padded per block).
        :param lazy_backward: build the transposed neighbour lists / backward block table on first need instead of up front.
        :param basis: use the basis form x = X0 + (sum_k w_k c_k) u (dc_points_basis) whenever an evaluation asks for neither pose
                      nor exponent gradients: the basis rows are rebuilt only when the poses or the exponents change.
        :param pose_kernel: evaluations with pose gradients in one launch (consistency_step_pose_kernel) where the plan allows it.
        :param nan_policy: None, 'skip_nans' or 'only_finite' (loss.py:125-137): pointwise losses that are NaN / not finite are left
                           out of the sum, the count and the gradients; the count of an evaluation is then its own (out[1]).
        :param mask_first: additionally collect the masked-out points of every 2048-point stretch into blocks of their own, which
                           the one-pass kernels skip whole (24 % of the wavefronts at C2 instead of 11 %).  OFF by default:
                           measured at C2 the remaining blocks then draw their neighbours from a third more distinct rows --
                           larger LDS tiles, fewer resident blocks -- and the step got SLOWER (44 -> 55 us); the two-kernel
                           forms lose more; super-blocks of 512 with a 768-row static tile: 19 % skipped, 44.5 against 42.4 us.
                           Results are the same either way.
        :param wave_pack: form the wavefronts (64 points) from points that are ALL inside or ALL outside the loss mask -- 64 Morton-
                          consecutive points of either kind -- and lay the wavefronts out in Morton order of their first point; a block
                          is four consecutive wavefronts.  The one-pass kernels skip a wavefront none of whose centres is inside the
                          mask, and with whole-block grouping alone only floor(outside / 64) wavefronts of a block qualify (C2: 26 % of
                          the points are outside, 10.7 % of the wavefronts were skipped).  Default: on for [rows, K <= 10] tables with
                          a mask.
        :param degree_group: ball neighbourhoods without pose gradients: lanes of a block ordered by (mask, row length) instead of
                             (mask, scan).
        :param heavy_first: ball neighbourhoods: the blocks with the longest rows first in the grid (see below).
        :param scan_group: group the points of every 256-point block by scan (contiguous per-scan lane ranges for the pose-gradient
                           sums; the set of points per block, hence every other kernel's work, is unchanged).
        :param active_only: evaluate only the masked points as neighbourhood centres (the others contribute neither to
                            the loss nor to any gradient); per-point outputs then cover the masked points only.
        """
        mark = stages.mark if stages is not None else (lambda name: None)
        get = (lambda c, f: c[f]) if isinstance(clouds[0], dict) else getattr
        vps = ops.cat_rows([get(c, 'vps') for c in clouds])
        dirs = ops.cat_rows([get(c, 'dirs') for c in clouds])
        depth = ops.cat_rows([get(c, 'depth') for c in clouds]).reshape(-1)
        incs = [get(c, 'inc_angles') for c in clouds]
        inc = None if incs[0] is None else ops.cat_rows(incs).reshape(-1)
        lms = [get(c, 'mask') for c in clouds]
        dev = dirs.device
        sizes = [len(get(c, 'dirs')) for c in clouds]
        lmask = None if lms[0] is None else ops.cat_rows(lms)
        scan_id = ops.scan_ids(sizes, dev)
        self.n, self.n_scans, self.device, self.dtype = dirs.shape[0], len(clouds), dev, dirs.dtype
        self._poses_key = self._poses12 = self._poses_ref = None
        self.sizes = sizes
        self.model_kind, self.loss, self.normalization, self.sqrt = model_kind, loss, bool(normalization), bool(sqrt)
        assert nan_policy in (None, 'skip_nans', 'only_finite')
        self.nan_policy = nan_policy
        g = getattr(neighbors, '_dc_graph', None)          # the int32 table the reference-typed indices were made from
        nbr = g.nbr if g is not None and g.version == neighbors._version else ops.as_index32(neighbors)
        need(nbr, (self.n, None), dtype=torch.int32, name='neighbors', device=dev)
        self.k = nbr.shape[1]
        if mask is not None:
            need(mask, (self.n,), dtype=torch.bool, name='mask', device=dev)

        # ---- layout: Morton order of the initial global cloud ----------------------------------
        mark('plan_concat')
        ps0 = ops.PointSet(vps if bool(vps.any()) else None, dirs, depth, inc, lmask, scan_id)
        P0 = self.poses12(poses)
        x0 = ops.points_fwd(ps0, P0)
        mark('plan_points0')
        if spatial_sort and self.n > 1:
            order = ops.spatial_order(x0).long()
            mark('plan_morton_order')
            if degree_sort:
                # inside every block of 256 Morton-consecutive points, order by in-degree: the lanes of a wavefront of
                # the backward then walk incoming-edge lists of similar length (same cache lines per block as before)
                valid = nbr[nbr >= 0].long()
                deg = torch.bincount(valid, minlength=self.n)[order]
                nb = (self.n + 255) // 256
                pad = torch.full((nb * 256,), 2 ** 40, dtype=torch.int64, device=dev)
                pad[:self.n] = deg
                local = torch.argsort(pad.reshape(nb, 256), dim=1, stable=True)
                pos = (local + torch.arange(nb, device=dev)[:, None] * 256).reshape(-1)
                order = order[pos[pos < self.n]]
            self.scan_seg, self.skipped_wavefronts, self.blk_skip = None, 0.0, None
            if wave_pack is None:
                # (K = 16: a block's list under this layout can pass the 512 rows the pose kernel's tile holds -- left as it was)
                wave_pack = nbr.shape[1] <= 10 and not heavy_first and not degree_group
            if wave_pack and mask is not None and not degree_sort and not active_only and not mask_first and self.n >= 4 * 256:
                # wavefronts of one kind: the points inside the mask, in Morton order, cut into runs of 64, and likewise the points
                # outside; the runs merged by the Morton position of their first point.  Every wavefront but the two at the ends of
                # the lists is all-in or all-out; a block (four consecutive runs) still covers one stretch of the curve -- its
                # distinct-row list grows by a few per cent, not by the third that whole super-blocks of one kind cost (mask_first)
                m_o = mask[order]
                pos = torch.arange(self.n, device=dev)
                pa, pb = pos[m_o], pos[~m_o]
                na, nb_ = pa.numel(), pb.numel()
                big = torch.iinfo(torch.int64).max
                # (the last, partial run of either kind goes to the very end: every run before it has exactly 64 points)
                ka = pa[::64].clone()
                kb = pb[::64].clone()
                if na % 64 and ka.numel():
                    ka[-1] = big - 1
                if nb_ % 64 and kb.numel():
                    kb[-1] = big
                keys = torch.cat([ka, kb])
                runs = torch.argsort(keys, stable=True)                    # run r: kind A when r < len(ka)
                start = torch.cat([torch.arange(ka.numel(), device=dev) * 64, na + torch.arange(kb.numel(), device=dev) * 64])[runs]
                length = torch.cat([torch.full((ka.numel(),), 64, device=dev), torch.full((kb.numel(),), 64, device=dev)])
                if na % 64 and ka.numel():
                    length[ka.numel() - 1] = na % 64
                if nb_ % 64 and kb.numel():
                    length[-1] = nb_ % 64
                length = length[runs]
                off = torch.cumsum(length, 0) - length
                idx = torch.repeat_interleave(start - off, length) + pos      # position in cat([pa, pb]) of every output slot
                order = order[torch.cat([pa, pb])[idx]]
                self._wave_packed = True
                mark('plan_wave_pack')
            if mask_first and mask is not None and not degree_sort and not active_only:
                # inside every super-block of kMaskGroup Morton-consecutive points: the points inside the loss mask first, then those
                # outside (Morton order inside each part).  Most blocks of 256 are then all-in or all-out, and the one-pass kernels
                # skip the all-out ones (dcSequenceDesc.blk_skip): a centre outside the mask adds nothing to the loss, the count or
                # dL/dw.  Every point stays in the table -- masked-out points are neighbours like any other -- and every block
                # still draws its neighbours from one super-block's surroundings (the LDS tiles grow by a third, not by an order)
                sb = torch.arange(self.n, device=dev) // self.kMaskGroup
                order = order[torch.argsort(sb * 2 + (~mask[order]).long(), stable=True)]
                mark('plan_mask_first')
            # ball neighbourhoods, no pose gradients expected: (mask, row length) instead of (mask, scan) -- the lanes of a wavefront of
            # the ragged one-pass kernel run to its longest row, so rows of similar length share wavefronts (64 bins of the
            # longest row; 112 -> 106 us at r = 0.4 m).  No scan ranges then: pose gradients take the un-grouped backward
            by_degree = bool(degree_group) and scan_group and not degree_sort
            deg = None
            if heavy_first and scan_group and not degree_sort and self.n // 256 > 8:
                # rows of different lengths (ball neighbourhoods): the blocks with the longest rows go out first.  The grid is a
                # handful of blocks per CU (1 103 blocks with 32-49 KB tiles at r = 0.25-0.4 m) and a block lasts as long as its
                # longest row, so in Morton order the launch ended whenever the last heavy block to start did: 107.7 -> 92.0 us
                # (r = 0.4 m), 52.8 -> 47.9 (0.25 m), 386 -> 293 (grid 0.1 m).  Block b of the grid is logical block
                # (b % 8) * per + b / 8 (xcd_block_of); a last, partly filled block keeps its place.  Which points share a block
                # is unchanged.  (Fixed K, weight = wavefronts inside the mask: slower, 43.2 -> 45.7 us at C2 -- the blocks of
                # an XCD no longer share rows in its L2 -- so not done there.)
                deg = (nbr >= 0).sum(1)
                nbf, per = self.n // 256, ((self.n + 255) // 256 + 7) // 8
                b = torch.arange(per * 8, device=dev)
                logical = (b % 8) * per + b // 8
                logical = logical[logical < nbf]
                dblk = deg[order[:nbf * 256]].reshape(nbf, 256)
                heavy = torch.argsort(dblk.max(1).values * 1024 + dblk.sum(1) // 256, descending=True, stable=True)
                src = torch.empty_like(heavy)
                src[logical] = heavy
                order = torch.cat([order[:nbf * 256].reshape(nbf, 256)[src].reshape(-1), order[nbf * 256:]])
                mark('plan_heavy_first')
            if scan_group and not degree_sort and (self.n_scans <= 64 or by_degree):
                # inside every block of 256 Morton-consecutive points: the points inside the loss mask first, then those outside;
                # each group ordered by scan (stable: Morton order inside a (mask, scan) segment).  Which 256 points share a block
                # does not change -- the LDS-staged gathers do not care about the lane order -- but (a) a block's points of one scan
                # become contiguous lane ranges known from now on, which is what the pose-gradient sums of the backward need, and
                # (b) its masked-out points fill whole wavefronts at its end, which the one-pass kernels skip (they add nothing to
                # the loss, the count or dL/dw).  dcSequenceDesc.scan_seg: [blocks, 2 S + 1]
                nb, S = (self.n + 255) // 256, (64 if by_degree else self.n_scans)
                group_key = scan_id
                if by_degree:
                    deg = (nbr >= 0).sum(1) if deg is None else deg
                    group_key = (deg * 63 // deg.max().clamp(min=1)).to(torch.int32).contiguous()
                order32 = order.to(torch.int32).contiguous()
                grouped = torch.empty_like(order32)
                seg16 = torch.empty((nb, 2 * S + 1), dtype=torch.uint16, device=dev)
                m8 = None if mask is None else mask.view(torch.uint8) if mask.dtype == torch.bool else mask
                # (the skipped-wavefront count and the blocks without a point inside the mask come from the same kernel: as tensor
                #  expressions over seg16 they were eight more torch kernels to load in the first set-up of a process)
                skip8 = torch.empty((nb,), dtype=torch.uint8, device=dev) if mask is not None else None
                nskip = torch.zeros((1,), dtype=torch.int32, device=dev)
                check(lib().dc_block_group(ptr(order32), ptr(group_key), ptr(m8), self.n, S, ptr(grouped), ptr(seg16), ptr(skip8), ptr(nskip),
                                           stream_ptr()), 'dc_block_group')
                order = grouped.long()
                self.scan_seg = None if by_degree else seg16
                # share of the wavefronts (64 lanes) whose centres are all outside the mask: what the one-pass kernels skip
                self.skipped_wavefronts = float(nskip.item()) / max((self.n + 63) // 64, 1)
                self.blk_skip = skip8
                mark('plan_scan_groups')
            if nbr.numel():
                rank32 = torch.empty((self.n,), dtype=torch.int32, device=dev)
                nbr_in = nbr.to(torch.int32).contiguous()
                nbr = torch.empty_like(nbr_in)
                check(lib().dc_table_permute(ptr(nbr_in), self.n, nbr_in.shape[1], ptr(order.contiguous()), ptr(rank32), ptr(nbr),
                                             stream_ptr()), 'dc_table_permute')
                rank = rank32.long()
            else:
                rank = torch.empty_like(order)
                rank[order] = torch.arange(self.n, device=dev)
                nbr = nbr[order].to(torch.int32).contiguous()
            gat = lambda t_: None if t_ is None else ops.gather_rows(t_.contiguous(), order)
            vps, dirs, depth, inc, lmask, scan_id, mask = (gat(t_) for t_ in (vps, dirs, depth, inc, lmask, scan_id, mask))
            self.order, self.rank = order, rank
            mark('plan_permute')
        else:
            self.order = self.rank = None
            self.scan_seg, self.skipped_wavefronts, self.blk_skip = None, 0.0, None
        if not bool(vps.any()):
            vps = None                      # sensor-frame scans: viewpoints are the origin, nothing to stream
        self.ps = ops.PointSet(vps, dirs, depth, inc, lmask, scan_id)
        self.nbr_full, self.mask_full = nbr, mask
        self.centre_idx = None
        if active_only and mask is not None:
            self.centre_idx = torch.nonzero(mask).reshape(-1).to(torch.int32).contiguous()
            nbr = nbr[self.centre_idx.long()].contiguous()
            mask = None
        self.nbr, self.mask = nbr, mask
        # the transposed neighbour lists and their block table serve the backward kernels (pose / exponent gradients, models
        # with more than three weights, the two-kernel ablations); the one-pass evaluation never reads them, so they are
        # built on first need (ensure_backward_tables: 0.7 + 1.9 ms at C2)
        self._csr = self._bwd_table = None
        self._bwd_layout, self._block_tables = bwd_layout, bool(block_tables)
        if not lazy_backward:
            self.ensure_backward_tables()
        mark('plan_transpose')
        self.lane_perm = None          # (dc_consistency_bwd can also take a per-block lane map; the layout does it here)
        # block tables: distinct rows per 256-point block + 16-bit block-local positions (gathers served from LDS)
        self.fwd_table = None
        if block_tables:
            # a ragged table (radius neighbourhoods: rows padded with -1 to the longest one) goes in as CSR lists: every block then
            # has as many slots as its longest row, not as the longest row of the whole cloud, and the rows are packed (no holes)
            ragged = nbr.shape[1] > 16 and nbr.shape[0] > 0 and bool((nbr[:, -1] < 0).any())
            if ragged:
                self.fwd_table = ops.block_table(csr=ops.table_to_csr(nbr), layout='slots', own_rows='csr' if self.centre_idx is None else False)
            else:
                self.fwd_table = ops.block_table(nbr=nbr, own_rows=self.centre_idx is None)
        self.fwd_rows_active = self.fwd_rows_active_loss = 0
        self.fwd_table_loss = None
        if (block_tables and self.fwd_table is not None and mask is not None and self.centre_idx is None and nbr.shape[1] <= 16
                and nbr.shape[0] > 0 and getattr(self, '_wave_packed', False)):
            # what the one-pass loss + dL/dw evaluation stages: only the rows the centres INSIDE the mask gather -- the points
            # outside keep the reference to themselves (a block's own rows stay in its list) and nothing else
            rows_ = torch.arange(self.n, device=dev, dtype=torch.int32)[:, None]
            nbr_loss = torch.where(mask[:, None] | (nbr == rows_), nbr, torch.full_like(nbr, -1)).contiguous()
            self.fwd_table_loss = ops.block_table(nbr=nbr_loss, own_rows=True)
            del nbr_loss
        if self.fwd_table is not None and getattr(self, 'blk_skip', None) is not None and self.centre_idx is None:
            # (on the host: two copies of a few kilobytes -- as tensor expressions, six more torch kernels to load in a fresh process)
            keep = self.blk_skip.cpu().numpy() == 0
            rows_per_block = np.diff(self.fwd_table.blk_ptr.cpu().numpy())
            self.fwd_rows_active = int(rows_per_block[keep].max()) if keep.any() else 0
            if self.fwd_table_loss is not None:
                rows_per_block = np.diff(self.fwd_table_loss.blk_ptr.cpu().numpy())
                self.fwd_rows_active_loss = int(rows_per_block[keep].max()) if keep.any() else 0
        mark('plan_block_tables')
        self.count = float(nbr.shape[0] if mask is None else int(mask.sum().item()))

        # ---- internal point format -------------------------------------------------------------------
        # 'q32': 32-bit fixed point sized for `margin` x the extent of the initial map; 'float': the clouds' own dtype;
        # 'f64': float32 clouds promoted to float64 (twice the traffic, no resolution limit); 'auto': q32 for float32
        # clouds while its resolution keeps the 1e-5 parity bar (ops.QFormat.MAX_AUTO_SCALE), otherwise f64
        self.qfmt = None
        self.status = torch.zeros((1,), dtype=torch.int32, device=dev)
        if point_format in ('auto', 'q32') and self.dtype == torch.float32:
            lo, hi = ops.points_extent(x0)                                   # one small reduction, one synchronisation
            qfmt = ops.QFormat.for_extent(lo, hi)
            if point_format == 'q32' or qfmt.scale <= ops.QFormat.MAX_AUTO_SCALE:
                self.qfmt = qfmt
            else:
                point_format = 'f64'
        elif point_format == 'q32':
            raise TypeError('the q32 point format goes with float32 clouds')
        if point_format == 'f64' and self.dtype == torch.float32:
            ps = self.ps
            up = lambda t: None if t is None else t.double()
            self.ps = ops.PointSet(up(ps.vps), up(ps.dirs), up(ps.depth), up(ps.inc), ps.lmask, ps.scan_id)
            self.dtype = torch.float64
        self.point_format = 'q32' if self.qfmt is not None else ('f64' if self.dtype == torch.float64 else 'f32')
        pdt = torch.int32 if self.qfmt is not None else self.dtype
        self.x = torch.empty((self.n, 4), dtype=pdt, device=dev)
        self.rec = torch.empty((nbr.shape[0], 8), dtype=pdt, device=dev)
        # sized by the library for the largest term count (ordinary columns + the two row buffers of chained steps)
        self.partials = torch.empty((ops.lib().dc_sequence_partials_count(self.n, ops.nv.MAX_MODEL_TERMS, self.n_scans),),
                                    dtype=torch.float64, device=dev)
        self.version = 0
        self._desc = None
        self.use_basis = bool(basis)
        self.use_pose_kernel = bool(pose_kernel)
        self._pose_table = self._local_basis = None      # built on the first pose-gradient evaluation
        self._basis = None             # (key, rows, poses12, exponent): the last two pin the storage the key names
        self._poses_key = self._poses12 = self._poses_ref = None

    # ------------------------------------------------------------------------------------------------
    @on_device
    def ensure_backward_tables(self):
        """Transposed neighbour lists (incoming edges per point) + their block table, built once."""
        if self._csr is None:
            self._csr = ops.knn_transpose(self.nbr, n_dst=self.n)
            self._bwd_table = ops.block_table(csr=self._csr, layout=self._bwd_layout) if self._block_tables else None
            self._desc = None                  # the descriptor carries their addresses
        return self._csr

    csr_ptr = property(lambda self: self.ensure_backward_tables()[0])
    csr_src = property(lambda self: self.ensure_backward_tables()[1])

    @property
    def bwd_table(self):
        self.ensure_backward_tables()
        return self._bwd_table

    def desc(self, n_terms):
        """dcSequenceDesc for the one-call native evaluation (dc_sequence_eval)."""
        if self._desc is None or self._desc.n_terms != n_terms:
            d = nv.SequenceDesc()
            d.n, d.k, d.n_scans = self.n, self.k, self.n_scans
            d.dtype = nv.DC_F32 if self.dtype == torch.float32 else nv.DC_F64
            d.point_fmt = nv.DC_Q32 if self.qfmt is not None else d.dtype
            if self.qfmt is not None:
                for i in range(4):
                    d.qparams[i] = self.qfmt._c[i]
            p = lambda t: None if t is None else t.data_ptr()
            ps = self.ps
            d.vps, d.dirs, d.depth, d.inc, d.lmask, d.scan_id = p(ps.vps), p(ps.dirs), p(ps.depth), p(ps.inc), p(ps.lmask), p(ps.scan_id)
            csr = self._csr or (None, None)           # absent until an evaluation needs them (DC_ERR_BACKWARD_TABLES)
            d.nbr, d.csr_ptr, d.csr_src, d.mask = p(self.nbr), p(csr[0]), p(csr[1]), p(self.mask)
            d.lane_perm = p(self.lane_perm)
            d.fwd_table = None if self.fwd_table is None else self.fwd_table.ref()
            ftl = getattr(self, 'fwd_table_loss', None)
            d.fwd_table_loss = None if ftl is None else ftl.ref()
            d.fwd_rows_active_loss = int(getattr(self, 'fwd_rows_active_loss', 0)) if ftl is not None else 0
            d.bwd_table = None if self._bwd_table is None else self._bwd_table.ref()
            d.centre_idx, d.n_centres = p(self.centre_idx), (0 if self.centre_idx is None else self.centre_idx.shape[0])
            d.x, d.rec, d.partials = p(self.x), p(self.rec), p(self.partials)
            d.partials_count = self.partials.numel()
            d.status = p(self.status)
            d.scan_seg = p(getattr(self, 'scan_seg', None))
            skip = getattr(self, 'blk_skip', None) if (self.mask is not None and self.centre_idx is None) else None
            d.blk_skip, d.fwd_rows_active = p(skip), int(getattr(self, 'fwd_rows_active', 0)) if skip is not None else 0
            d.model_kind = nv.MODEL_KINDS[self.model_kind] if n_terms > 0 else 0
            d.n_terms = n_terms
            d.loss_kind, d.normalization, d.sqrt_ = nv.LOSS_KINDS[self.loss], int(self.normalization), int(self.sqrt)
            d.loss_kind |= {None: 0, 'skip_nans': nv.DC_LOSS_SKIP_NANS, 'only_finite': nv.DC_LOSS_ONLY_FINITE}[self.nan_policy]
            if d.model_kind != 0 and ps.inc is None:
                raise ValueError('the model needs incidence angles')
            self._desc = d
        return self._desc

    def _set_pose_tables(self, d, w, exponent, want_grad, want_exponent, want_pose):
        """Point the descriptor at the pose tables (built once, on the first evaluation that asks for pose gradients) and at the
        local basis rows valid for ``exponent`` (rebuilt when the exponent tensor changes identity or version), or clear the
        fields: pose-gradient evaluations then run as one launch (consistency_step_pose_kernel)."""
        ok = (self.use_pose_kernel and want_grad and want_pose and not want_exponent and w is not None and w.numel() in (1, 2)
              and self.qfmt is not None and self.ps.vps is None and self.centre_idx is None and self.fwd_table is not None
              and self.fwd_table.own_base is not None and self.k in (4, 8, 10, 16) and self.n_scans <= 32
              and d.model_kind != 0 and self._block_tables and self.nbr.shape[1] == self.k)
        if ok and self._pose_table is None:
            self._pose_table = self._build_pose_table()
        if not ok or self._pose_table is False:
            d.pose_table = d.local_basis = None
            return
        nt = w.numel()
        key = (exponent.data_ptr(), exponent._version, nt, d.model_kind)
        lb = self._local_basis
        if lb is None or lb[0] != key:
            ps = self.ps
            # per block of the pose table, in the order of its list: 24 B per listed row
            rows = torch.empty((max(self._pose_table[1]['ids'].numel(), 1), 6), dtype=torch.int32, device=self.device)
            check(lib().dc_points_local_basis(ptr(ps.dirs), ptr(ps.depth), ptr(ps.inc), ptr(ps.lmask), ptr(ps.scan_id), d.model_kind, nt,
                                              ptr(exponent), self.n, nv.DC_F32, ctypes.byref(self._pose_table[0]), ptr(rows),
                                              stream_ptr()), 'dc_points_local_basis')
            self._local_basis = lb = (key, rows, exponent)
        d.pose_table = ctypes.cast(ctypes.pointer(self._pose_table[0]), ctypes.c_void_p)
        d.local_basis = lb[1].data_ptr()

    def _build_pose_table(self):
        """dcPoseTable of this plan's forward table (dc_pose_table_build), or False when some block cannot take the pose kernel."""
        # (from the loss-only table when the plan has one: a centre outside the mask adds no edge gradient either, and its lists are
        # the ones that fit the kernel's 512-row tile under the wave-packed layout)
        ft, dev, nb = (getattr(self, 'fwd_table_loss', None) or self.fwd_table), self.device, (self.n + 255) // 256
        total = int(ft.blk_ptr[-1].item())
        t = dict(ids=torch.empty((max(total, 1),), dtype=torch.int32, device=dev),
                 loc=torch.empty((nb * self.k * 256,), dtype=torch.uint16, device=dev),
                 own_pos=torch.empty((self.n,), dtype=torch.uint16, device=dev),
                 row_seg=torch.empty((nb * (self.n_scans + 1),), dtype=torch.uint16, device=dev),
                 row_scan=torch.empty((max(total, 1),), dtype=torch.uint8, device=dev))
        info = torch.zeros((1,), dtype=torch.int32, device=dev)
        check(lib().dc_pose_table_build(ft.ref(), ptr(self.ps.scan_id), self.n, self.n_scans, self.k, ptr(t['ids']), ptr(t['loc']),
                                        ptr(t['own_pos']), ptr(t['row_seg']), ptr(t['row_scan']), ptr(info), stream_ptr()),
              'dc_pose_table_build')
        if int(info.item()) != 0:
            return False
        desc = nv.PoseTableDesc(ptr(ft.blk_ptr), ptr(t['ids']), ptr(t['loc']), ptr(t['own_pos']), ptr(t['row_seg']),
                                ptr(t['row_scan']))
        return desc, t

    def _set_basis(self, d, w, exponent, poses12, want_exponent, want_pose):
        """Point the descriptor at the basis rows valid for (poses12, exponent), building them when either changed; or clear the
        fields when this evaluation cannot use the basis form (pose / exponent gradients, other formats, no tables)."""
        f64 = self.qfmt is None and self.x.dtype == torch.float64 and self.ps.dirs.dtype == torch.float64
        ok = (self.use_basis and (self.qfmt is not None or f64) and w is not None
              and not want_exponent and not want_pose
              and self.fwd_table is not None and self._block_tables and self._bwd_layout == 'runs'
              and d.model_kind != 0)
        if not ok:
            d.basis = None
            return
        nt = w.numel()
        b = self._basis
        key = (poses12.data_ptr(), poses12._version, exponent.data_ptr(), exponent._version, nt)
        if b is None or b[0] != key:
            ps = self.ps
            # float32 clouds: int32 / float32 words on the q32 grid; float64 clouds: fp64 words
            rows = torch.empty((self.n, 6 + nt), dtype=torch.float64 if f64 else torch.int32, device=self.device)
            check(lib().dc_points_basis(ptr(ps.vps), ptr(ps.dirs), ptr(ps.depth), ptr(ps.inc), ptr(ps.lmask), ptr(ps.scan_id),
                                        ptr(poses12), self.n_scans, d.model_kind, nt, ptr(exponent), self.n,
                                        nv.DC_F64 if f64 else nv.DC_F32, None if f64 else self.qfmt._c, ptr(rows),
                                        ptr(self.status), stream_ptr()), 'dc_points_basis')
            # the tensors are kept alive with the entry, so their addresses cannot be recycled for other poses / exponents
            self._basis = b = (key, rows, poses12, exponent)
        d.basis = b[1].data_ptr()

    @on_device
    def eval_native(self, w, exponent, poses12, out, want_grad=True, want_exponent=False, want_pose=False):
        """One host call per evaluation.  w, exponent: fp64 device vectors [P]; poses12: fp64 device [S,12];
        out: fp64 device [2 + 2P + 12S] <- {sum loss, count, d/dw, d/dexponent, d/d[R|t]}."""
        nt = 0 if w is None else w.numel()
        d = self.desc(nt)
        need(poses12, (self.n_scans, 12), dtype=torch.float64, name='poses12', device=self.device)
        need(out, (2 + 2 * nt + 12 * self.n_scans,), dtype=torch.float64, name='out', device=self.device)
        if nt:
            need(w, (nt,), dtype=torch.float64, name='w', device=self.device)
            need(exponent, (nt,), dtype=torch.float64, name='exponent', device=self.device)
        for attempt in (0, 1):
            self._set_basis(d, w, exponent, poses12, want_exponent, want_pose)
            self._set_pose_tables(d, w, exponent, want_grad, want_exponent, want_pose)
            rc = lib().dc_sequence_eval(ctypes.byref(d), ptr(w), ptr(exponent), ptr(poses12), int(want_grad),
                                        int(want_exponent), int(want_pose), ptr(out), stream_ptr())
            if rc != nv.DC_ERR_BACKWARD_TABLES or attempt:
                break
            self.ensure_backward_tables()          # this evaluation walks the transposed lists: build them now, once
            d = self.desc(nt)
        check(rc, 'dc_sequence_eval')
        self.version += 1
        return out

    @on_device
    def step_native(self, w, exponent, poses12, out, exp_avg, exp_avg_sq, t, grad_scale, lr, betas, eps, weight_decay):
        """Evaluation + Adam step of ``w`` (in place) in one host call (dc_sequence_step); out as in eval_native."""
        nt = w.numel()
        d = self.desc(nt)
        need(poses12, (self.n_scans, 12), dtype=torch.float64, name='poses12', device=self.device)
        need(out, (2 + 2 * nt + 12 * self.n_scans,), dtype=torch.float64, name='out', device=self.device)
        for name, v in (('w', w), ('exponent', exponent), ('exp_avg', exp_avg), ('exp_avg_sq', exp_avg_sq)):
            need(v, (nt,), dtype=torch.float64, name=name, device=self.device)
        for attempt in (0, 1):
            self._set_basis(d, w, exponent, poses12, False, False)
            rc = lib().dc_sequence_step(ctypes.byref(d), ptr(w), ptr(exponent), ptr(poses12), ptr(exp_avg), ptr(exp_avg_sq),
                                        int(t), float(grad_scale), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                        float(weight_decay), ptr(out), stream_ptr())
            if rc != nv.DC_ERR_BACKWARD_TABLES or attempt:
                break
            self.ensure_backward_tables()
            d = self.desc(nt)
        check(rc, 'dc_sequence_step')
        self.version += 1
        return out

    @on_device
    def step_chained(self, w, exponent, poses12, out_prev, exp_avg, exp_avg_sq, t, has_prev, ready, grad_scale, lr, betas, eps,
                     weight_decay, w_used_prev=None):
        """One step of a CHAIN (dc_sequence_step_chained): evaluation number ``t`` is launched, and the same launch first
        finishes evaluation ``t - 1`` when ``has_prev`` (its sums -> out_prev, its Adam update on ``w``; the weights it had
        used -> ``w_used_prev`` when given).  Returns False when this plan / model cannot chain (the caller steps with
        step_native then)."""
        nt = w.numel()
        d = self.desc(nt)
        need(out_prev, (2 + 2 * nt + 12 * self.n_scans,), dtype=torch.float64, name='out_prev', device=self.device)
        if w_used_prev is not None:
            need(w_used_prev, (nt,), dtype=torch.float64, name='w_used_prev', device=self.device)
            for name, v in (('w', w), ('exponent', exponent), ('exp_avg', exp_avg), ('exp_avg_sq', exp_avg_sq)):
                need(v, (nt,), dtype=torch.float64, name=name, device=self.device)
            need(ready, (16,), dtype=torch.int32, name='ready', device=self.device)
            self._set_basis(d, w, exponent, poses12, False, False)
            rc = lib().dc_sequence_step_chained_rec(ctypes.byref(d), ptr(w), ptr(exponent), ptr(poses12), ptr(exp_avg), ptr(exp_avg_sq),
                                                    int(t), int(bool(has_prev)), float(grad_scale), float(lr), float(betas[0]),
                                                    float(betas[1]), float(eps), float(weight_decay), ptr(ready), ptr(out_prev),
                                                    ptr(w_used_prev), stream_ptr())
            if rc in (-4, nv.DC_ERR_BACKWARD_TABLES):
                return False
            check(rc, 'dc_sequence_step_chained_rec')
            self.version += 1
            return True
        need(ready, (16,), dtype=torch.int32, name='ready', device=self.device)
        for name, v in (('w', w), ('exponent', exponent), ('exp_avg', exp_avg), ('exp_avg_sq', exp_avg_sq)):
            need(v, (nt,), dtype=torch.float64, name=name, device=self.device)
        self._set_basis(d, w, exponent, poses12, False, False)
        rc = lib().dc_sequence_step_chained(ctypes.byref(d), ptr(w), ptr(exponent), ptr(poses12), ptr(exp_avg), ptr(exp_avg_sq),
                                            int(t), int(bool(has_prev)), float(grad_scale), float(lr), float(betas[0]),
                                            float(betas[1]), float(eps), float(weight_decay), ptr(ready), ptr(out_prev), stream_ptr())
        if rc in (-4, nv.DC_ERR_BACKWARD_TABLES):
            return False
        check(rc, 'dc_sequence_step_chained')
        self.version += 1
        return True

    @on_device
    def step_linked(self, prev_plan, prev_parity, finish, acc_in, w, exponent, poses12, exp_avg, exp_avg_sq, step, stamp, ready, out_prev,
                    grad_scale, lr, betas, eps, weight_decay, w_used_prev=None):
        """One launch of a chain over the SEVERAL sequences of one loss (dc_sequence_step_linked): this sequence is evaluated and the
        launch before it -- ``prev_plan``'s, buffer ``prev_parity`` -- is finished first: its rows summed onto ``acc_in`` into
        ``out_prev`` (finish = 1), with the Adam update when they complete a step (finish = 2); finish = 0: nothing pending.
        Returns False when this plan / model cannot chain."""
        nt = w.numel()
        d = self.desc(nt)
        need(poses12, (self.n_scans, 12), dtype=torch.float64, name='poses12', device=self.device)
        need(out_prev, (None,), dtype=torch.float64, name='out_prev', device=self.device)
        assert out_prev.numel() >= 2 + 2 * nt + 12 * self.n_scans
        need(ready, (16,), dtype=torch.int32, name='ready', device=self.device)
        self._set_basis(d, w, exponent, poses12, False, False)
        dp = None if (prev_plan is None or not finish) else prev_plan.desc(nt)
        rc = lib().dc_sequence_step_linked(ctypes.byref(d), None if dp is None else ctypes.byref(dp), int(prev_parity), int(finish), ptr(acc_in),
                                           ptr(w), ptr(exponent), ptr(poses12), ptr(exp_avg), ptr(exp_avg_sq), int(step), int(stamp),
                                           float(grad_scale), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                           ptr(ready), ptr(out_prev), ptr(w_used_prev), stream_ptr())
        if rc in (-4, nv.DC_ERR_BACKWARD_TABLES):
            return False
        check(rc, 'dc_sequence_step_linked')
        self.version += 1
        return True

    @on_device
    def flush_linked(self, parity, acc_in, w, exp_avg, exp_avg_sq, step, stamp, ready, out, grad_scale, lr, betas, eps, weight_decay):
        """Finish the last launch of a linked chain (this plan's rows, buffer ``parity``, + ``acc_in`` -> out; Adam update ``step``)."""
        d = self.desc(w.numel())
        need(out, (2 + 2 * w.numel() + 12 * self.n_scans,), dtype=torch.float64, name='out', device=self.device)
        check(lib().dc_sequence_chain_flush_linked(ctypes.byref(d), int(parity), ptr(acc_in), ptr(w), ptr(exp_avg), ptr(exp_avg_sq), int(step),
                                                   int(stamp), float(grad_scale), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                   float(weight_decay), ptr(ready), ptr(out), stream_ptr()), 'dc_sequence_chain_flush_linked')
        return out

    @on_device
    def eval_after_update(self, w, exponent, poses12, out, exp_avg, exp_avg_sq, t, grad_sum, ready, grad_scale, lr, betas, eps,
                          weight_decay, w_used=None):
        """Evaluation number ``t`` whose launch first takes Adam update ``t - 1`` from ``grad_sum`` (the previous evaluation's
        dL/dw, already summed over sequences / ranks; None for the first call), then the ordinary reduction of this
        evaluation into ``out`` (dc_sequence_eval_after_update); ``w_used`` (fp64 [P], optional) <- the weights this evaluation
        uses.  Returns False when this plan / model cannot do that."""
        nt = w.numel()
        d = self.desc(nt)
        need(out, (2 + 2 * nt + 12 * self.n_scans,), dtype=torch.float64, name='out', device=self.device)
        need(ready, (16,), dtype=torch.int32, name='ready', device=self.device)
        if grad_sum is not None:
            need(grad_sum, (nt,), dtype=torch.float64, name='grad_sum', device=self.device)
        self._set_basis(d, w, exponent, poses12, False, False)
        rc = lib().dc_sequence_eval_after_update(ctypes.byref(d), ptr(w), ptr(exponent), ptr(poses12), ptr(exp_avg), ptr(exp_avg_sq),
                                                 int(t), ptr(grad_sum), float(grad_scale), float(lr), float(betas[0]),
                                                 float(betas[1]), float(eps), float(weight_decay), ptr(ready), ptr(out), ptr(w_used),
                                                 stream_ptr())
        if rc in (-4, nv.DC_ERR_BACKWARD_TABLES):
            return False
        check(rc, 'dc_sequence_eval_after_update')
        self.version += 1
        return True

    @on_device
    def chain_flush(self, w, out, exp_avg, exp_avg_sq, t, grad_scale, lr, betas, eps, weight_decay):
        """Finish evaluation ``t`` of a chain: its sums -> out, its Adam update on ``w`` (dc_sequence_chain_flush)."""
        nt = w.numel()
        d = self.desc(nt)
        need(out, (2 + 2 * nt + 12 * self.n_scans,), dtype=torch.float64, name='out', device=self.device)
        check(lib().dc_sequence_chain_flush(ctypes.byref(d), ptr(w), ptr(exp_avg), ptr(exp_avg_sq), int(t), float(grad_scale),
                                            float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), ptr(out),
                                            stream_ptr()), 'dc_sequence_chain_flush')
        return out

    @on_device
    def eval_gated(self, w, exponent, poses12, out, inlier_ratio=1.0, inlier_max_loss=None, inlier_loss_mult=1.0,
                   want_grad=True, want_exponent=False, want_pose=False):
        """Evaluation with the quantile-inlier gating of loss.py:256-277 (per sequence, as batch_loss applies it): of the
        masked points only those whose raw loss is <= inlier_loss_mult * quantile(raw loss, inlier_ratio) (and <=
        inlier_max_loss) count.  out as in eval_native, out[1] = number of inliers.  Five launches + the quantile (a
        sort of the masked losses) instead of one call: forward with raw pointwise losses -> bound -> dc_consistency_gate
        (drops the outliers' records, re-sums) -> backward."""
        nt = 0 if w is None else w.numel()
        need(out, (2 + 2 * nt + 12 * self.n_scans,), dtype=torch.float64, name='out', device=self.device)
        self.P, self.w, self.e = poses12, w, exponent
        kind = self.model_kind if w is not None else None
        ops.points_fwd(self.ps, poses12, kind, w, exponent, stride=4, qfmt=self.qfmt, out=self.x, status=self.status)
        fw = ops.consistency_fwd(self.x, self.nbr, mask=self.mask, loss=self.loss, normalization=self.normalization,
                                 sqrt=self.sqrt, rec=self.rec, want_pointwise=True, raw_pointwise=True,
                                 partials=self.partials, qfmt=self.qfmt, centre_idx=self.centre_idx, table=self.fwd_table)
        raw = fw['pointwise']
        bound = None if inlier_max_loss is None else torch.as_tensor(inlier_max_loss, dtype=raw.dtype, device=raw.device)
        if inlier_ratio < 1.0:
            q = _quantile(raw if self.mask is None else raw[self.mask], float(inlier_ratio))
            if inlier_loss_mult != 1.0:
                q = inlier_loss_mult * q
            bound = q if bound is None else torch.min(bound, q)
        if bound is None:
            raise ValueError('eval_gated needs inlier_ratio < 1 or inlier_max_loss')
        inl = raw <= bound                                   # the rows dc_consistency_gate keeps (same comparison, same dtype)
        self.inlier_bound, self.inlier_rows = bound, (inl if self.mask is None else inl & self.mask)
        ops.consistency_gate(raw, self.rec, bound.to(torch.float64).reshape(()), mask=self.mask, sqrt=self.sqrt,
                             partials=self.partials, sums=out[:2], qfmt=self.qfmt)
        if want_grad:
            gw, ge, gp = self.backward(want_exponent=want_exponent, want_pose=want_pose)
            out[2:2 + nt], out[2 + nt:2 + 2 * nt], out[2 + 2 * nt:] = gw, ge, gp.reshape(-1)
        else:
            out[2:].zero_()
        if self.status is not None:
            out[0] = torch.where(self.status[0] != 0, torch.full_like(out[0], float('nan')), out[0])
        self.version += 1
        return out

    # ------------------------------------------------------------------------------------------------
    def poses12(self, poses):
        """[S,4,4] poses -> contiguous fp64 [S,12] on the plan's device.  The last conversion is kept: an optimisation
        without pose corrections passes the same (unmodified) tensor every iteration (train.py:220-312)."""
        if isinstance(poses, torch.Tensor):
            key = (poses.data_ptr(), poses._version, poses.dtype, poses.device)
            if self._poses_key == key and not poses.requires_grad:
                return self._poses12
        else:
            poses, key = torch.as_tensor(poses, device=self.device), None
        assert poses.shape == (self.n_scans, 4, 4), poses.shape
        out = poses.detach().to(device=self.device, dtype=torch.float64)[:, :3, :].reshape(self.n_scans, 12).contiguous()
        if key is not None and not poses.requires_grad:
            self._poses_key, self._poses12, self._poses_ref = key, out, poses
        return out

    def forward(self, w, exponent, poses, want_pointwise=False, want_eigvals=False):
        """One forward evaluation; returns dict(sums=[sum of pointwise loss over the mask, mask count], ...)."""
        self.P = self.poses12(poses)
        self.w = None if w is None else w.detach().reshape(-1).to(torch.float64).contiguous()
        self.e = None if exponent is None else exponent.detach().reshape(-1).to(device=self.device, dtype=torch.float64).contiguous()
        kind = self.model_kind if self.w is not None else None
        ops.points_fwd(self.ps, self.P, kind, self.w, self.e, stride=4, qfmt=self.qfmt, out=self.x, status=self.status)
        if self.centre_idx is not None and (want_pointwise or want_eigvals):
            # per-point outputs for ALL points: the complete neighbour table, records into a scratch buffer
            out = ops.consistency_fwd(self.x, self.nbr_full, mask=self.mask_full, loss=self.loss,
                                      normalization=self.normalization, sqrt=self.sqrt, want_pointwise=want_pointwise,
                                      want_eigvals=want_eigvals, qfmt=self.qfmt)
            out['rec'] = None
            return out
        out = ops.consistency_fwd(self.x, self.nbr, mask=self.mask, loss=self.loss, normalization=self.normalization,
                                  sqrt=self.sqrt, rec=self.rec, want_pointwise=want_pointwise,
                                  want_eigvals=want_eigvals, partials=self.partials, qfmt=self.qfmt,
                                  centre_idx=self.centre_idx, table=self.fwd_table)
        self.version += 1
        return out

    def backward(self, want_exponent=False, want_pose=False):
        """Gradients of the *sum* of the pointwise loss over the mask w.r.t. (w [P], exponent [P], [R|t] [S,3,4])."""
        kind = self.model_kind if self.w is not None else None
        _, grads = ops.consistency_bwd(self.x, self.rec, self.csr_ptr, self.csr_src, self.ps, self.P, kind, self.w, self.e,
                                       want_exponent=want_exponent, want_pose=want_pose, partials=self.partials,
                                       qfmt=self.qfmt, lane_perm=self.lane_perm, table=self.bwd_table)
        return grads

    STATUS_OVERFLOW, STATUS_CHAIN_TIMEOUT = 1, 2

    def status_bits(self):
        """The sequence's status word (synchronises): bit 0 = an evaluation produced points outside the q32 format's extent
        or NaN points, bit 1 = a chained launch's wait for its weights ran out.  Either makes the losses NaN until
        ``clear_status``."""
        return int(self.status.item())

    def overflowed(self):
        """True when an evaluation produced points outside the q32 format's extent (poses moved far from the initial
        map) or NaN points; the losses of such evaluations are NaN.  Synchronises."""
        return bool(self.status_bits() & self.STATUS_OVERFLOW)

    def chain_timed_out(self):
        """True when a chained step gave up waiting for the weights of its launch (its sums are NaN).  Synchronises."""
        return bool(self.status_bits() & self.STATUS_CHAIN_TIMEOUT)

    def clear_status(self):
        self.status.zero_()

    def unpermute(self, t):
        """Per-point tensor in plan order -> the caller's scan-major order."""
        return t if self.rank is None else t[self.rank]

    def points(self):
        """Current global points [N,3] in the caller's order (float tensor of the cloud dtype)."""
        if self.qfmt is not None:
            o = torch.as_tensor(self.qfmt.origin, dtype=torch.float64, device=self.device)
            x = (o + self.x[:, :3].double() * self.qfmt.scale).to(self.dtype)
        else:
            x = self.x[:, :3]
        return self.unpermute(x)


class PlanRegistry(object):
    """Plans (SequencePlan, ops.IcpSequence) of the clouds / neighbourhoods / correspondences an optimisation loop passes
    every iteration (train.py:212-215).  An entry is keyed by the identity AND the version counter of every tensor it was built from and
    holds references to them, so an id can neither be recycled after garbage collection nor can an in-place edit go
    unnoticed; the registry keeps the few most recent entries (one per sequence of a training / validation set)."""

    def __init__(self, capacity=64):
        self.capacity, self.entries = capacity, []

    @staticmethod
    def key_of(tensors, extra):
        return tuple((id(t), t._version) if isinstance(t, torch.Tensor) else t for t in tensors) + tuple(extra)

    def get(self, tensors, extra, build):
        key = self.key_of(tensors, extra)
        for i, (k, refs, plan) in enumerate(self.entries):
            if k == key and all(a is b for a, b in zip(refs, tensors)):
                if i:
                    self.entries.insert(0, self.entries.pop(i))
                return plan
        plan = build()
        self.entries.insert(0, (key, list(tensors), plan))
        del self.entries[self.capacity:]
        return plan

    def clear(self):
        self.entries = []

    def keys(self):
        return [k for k, _, _ in self.entries]

    def discard_except(self, keep):
        """Drop every entry whose key is not in ``keep`` (train() releases what it added and nothing else)."""
        self.entries = [e for e in self.entries if e[0] in keep]


class _ConsistencyLoss(torch.autograd.Function):
    """sum over the mask of the pointwise loss of one sequence, differentiable w.r.t. w, exponent, poses.

    Forward and backward kernels are issued together by ONE dc_sequence_eval call whenever a gradient will be needed
    (the gradients of the sum are a handful of numbers); ``backward`` only scales them by the upstream gradient."""

    @staticmethod
    def forward(ctx, plan, w, exponent, poses, gating=None):
        need_w = isinstance(w, torch.Tensor) and w.requires_grad
        need_e = isinstance(exponent, torch.Tensor) and exponent.requires_grad
        need_p = isinstance(poses, torch.Tensor) and poses.requires_grad
        dev = plan.device
        wv = None if w is None else _as_f64_vector(w, dev)
        ev = None if w is None else _as_f64_vector(exponent, dev)
        nt = 0 if wv is None else wv.numel()
        P = plan.poses12(poses)
        out = torch.empty((2 + 2 * nt + 12 * plan.n_scans,), dtype=torch.float64, device=dev)
        if gating:                                       # quantile inliers (loss.py:256-277): out[1] = their number
            plan.eval_gated(wv, ev, P, out, want_grad=need_w or need_e or need_p, want_exponent=need_e, want_pose=need_p,
                            **gating)
        else:
            plan.eval_native(wv, ev, P, out, want_grad=need_w or need_e or need_p, want_exponent=need_e, want_pose=need_p)
        plan.w, plan.e, plan.P = wv, ev, P              # for PlanCloud / plan.backward() users
        ctx.save_for_backward(out)
        ctx.meta = (nt, plan.n_scans, None if w is None else (w.shape, w.dtype, w.device),
                    (exponent.shape, exponent.dtype, exponent.device) if isinstance(exponent, torch.Tensor) else None,
                    (poses.dtype, poses.device) if isinstance(poses, torch.Tensor) else None)
        if gating or plan.nan_policy:                    # the evaluation's own count: inliers / points whose loss was kept
            n_kept = out[1]
            ctx.mark_non_differentiable(n_kept)
            return out[0], n_kept
        return out[0]                                    # a view: no copy kernel; `out` itself is what backward reads

    @staticmethod
    def backward(ctx, grad_out, *unused):
        (out,) = ctx.saved_tensors
        nt, ns, wmeta, emeta, pmeta = ctx.meta
        g = grad_out * out[2:]                           # fp64 on the plan's device, like the upstream gradient of out[0]
        gw = ge = gp = None
        if ctx.needs_input_grad[1] and wmeta is not None:
            gw = g[:nt].reshape(wmeta[0])
            if gw.dtype != wmeta[1] or gw.device != wmeta[2]:
                gw = gw.to(device=wmeta[2], dtype=wmeta[1])
        if ctx.needs_input_grad[2] and emeta is not None:
            ge = g[nt:2 * nt].reshape(emeta[0]).to(device=emeta[2], dtype=emeta[1])
        if ctx.needs_input_grad[3] and pmeta is not None:
            gp = torch.zeros((ns, 4, 4), dtype=torch.float64, device=out.device)
            gp[:, :3, :] = g[2 * nt:].reshape(ns, 3, 4)
            gp = gp.to(device=pmeta[1], dtype=pmeta[0])
        return None, gw, ge, gp, None


def _quantile(v, q):
    """torch.quantile(v, q) (linear interpolation between the two order statistics, in v's dtype -- what loss.py:259 calls);
    beyond torch.quantile's 16 M-element limit the same arithmetic on an explicit sort."""
    if v.numel() <= 16_000_000:
        return torch.quantile(v, q, dim=0)
    s, _ = torch.sort(v)
    rank = torch.tensor(q, dtype=v.dtype, device=v.device) * (v.numel() - 1)
    lo = rank.floor().long()
    hi = torch.clamp(lo + 1, max=v.numel() - 1)
    return torch.lerp(s[lo], s[hi], rank - lo.to(v.dtype))


def _as_f64_vector(t, dev):
    """Detached contiguous fp64 vector on ``dev``; no copies and no extra dispatches when ``t`` already is one."""
    v = t.detach().reshape(-1)
    if v.dtype != torch.float64 or v.device != dev:
        v = v.to(device=dev, dtype=torch.float64)
    return v if v.is_contiguous() else v.contiguous()


def consistency_loss(plan, w, exponent, poses, inlier_ratio=1.0, inlier_max_loss=None, inlier_loss_mult=1.0):
    """(sum of pointwise loss over the mask, mask count) of one sequence; the sum carries the autograd graph.  With
    quantile-inlier gating (loss.py:256-277) the count is the number of inliers, a device scalar."""
    if inlier_ratio < 1.0 or inlier_max_loss is not None:
        gating = dict(inlier_ratio=float(inlier_ratio), inlier_max_loss=inlier_max_loss, inlier_loss_mult=float(inlier_loss_mult))
        return _ConsistencyLoss.apply(plan, w, exponent, poses, gating)
    if plan.nan_policy:                                  # NaN-dropping reduction: the count is this evaluation's (a device scalar)
        return _ConsistencyLoss.apply(plan, w, exponent, poses)
    return _ConsistencyLoss.apply(plan, w, exponent, poses), plan.count


class KernelTimer:
    """HIP-event timing of the hot kernels (the three of the fused step and dc_features_fwd's), recorded inside the library on
    the launch stream."""
    KINDS = ('points_fwd', 'consistency_fwd', 'consistency_bwd', 'features_fwd')

    def __init__(self, every=1):
        """every: time every N-th launch of each kernel (an event pair idles the GPU for a few microseconds)."""
        self.every = int(every)

    def __enter__(self):
        check(lib().dc_profiler_reset(), 'dc_profiler_reset')
        check(lib().dc_profiler_enable(self.every), 'dc_profiler_enable')
        return self

    def __exit__(self, *exc):
        check(lib().dc_profiler_enable(0), 'dc_profiler_enable')

    def read(self):
        """{kernel: (mean ms per launch, launches)}; waits for the recorded launches."""
        out = {}
        for i, name in enumerate(self.KINDS):
            tot, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
            check(lib().dc_profiler_read(i, ctypes.byref(tot), ctypes.byref(cnt)), 'dc_profiler_read')
            if cnt.value:
                out[name] = (tot.value / cnt.value, cnt.value)
        return out

    def kernels(self):
        """{kind: source-level name of the kernel instantiation its last launch used}."""
        out = {}
        for i, name in enumerate(self.KINDS):
            buf = ctypes.create_string_buffer(256)
            check(lib().dc_profiler_kernel(i, buf, 256), 'dc_profiler_kernel')
            out[name] = buf.value.decode().strip('()')
        return out


class PoseSequenceTrainer:
    """One sequence of train()'s loop WITH pose corrections (train.py:300-322; scripts/model_poses_learning:71) on the library's own
    launches: per iteration ``evaluate`` = dc_sequence_eval with the corrected poses (one launch of consistency_step_pose_kernel +
    the reduction where the plan qualifies) and ``finish`` = dc_pose_train_finish -- the adjoint of the pose chain, the first pose
    kept fixed, torch.optim.Adam's update on the corrections (and on the model weights when given), the corrected poses of the next
    iteration and the iteration's record, in ONE launch.  The corrections are kept in fp64 (``delta``; the caller's tensor is
    written back by ``store``): Adam on float32 corrections differs from this in the rounding of every update."""

    @on_device
    def __init__(self, plan, poses0, deltas, zero_first, lr, betas=(0.9, 0.999), eps=1e-8, n_terms=2, icp_model_kind=None):
        """``plan``: a SequencePlan (map-consistency loss) or, with ``icp_model_kind`` (the model's kernel kind, '' for none), an
        ops.IcpSequence (point-to-plane / point-to-point ICP loss of the sequence's scan pairs)."""
        dev = plan.device
        S = plan.n_scans
        self.plan, self.S, self.nt, self.device = plan, S, int(n_terms), dev
        self.icp = icp_model_kind is not None
        self.icp_kind = icp_model_kind or None
        self.head = 1 if self.icp else 2
        self.T0 = poses0.detach().to(device=dev, dtype=torch.float64).reshape(S, 16).contiguous()
        self.delta = deltas.detach().to(device=dev, dtype=torch.float64).contiguous().clone()
        self.nd = self.delta.shape[0]
        assert self.delta.shape == (self.nd, 6) and self.nd in (1, S)
        self.d_m, self.d_v = torch.zeros_like(self.delta), torch.zeros_like(self.delta)
        self.step = torch.zeros((), dtype=torch.int64, device=dev)
        self.zero_first, self.lr, self.betas, self.eps = int(bool(zero_first)), float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.T = torch.empty((S, 16), dtype=torch.float64, device=dev)         # the corrected poses of the current iteration
        self.P12 = torch.empty((S, 12), dtype=torch.float64, device=dev)
        self.out = torch.zeros((self.head + 2 * self.nt + 12 * S,), dtype=torch.float64, device=dev)
        check(lib().dc_pose_correct_fwd(ptr(self.T0), ptr(self.delta), S, self.nd, ptr(self.T), stream_ptr()), 'dc_pose_correct_fwd')
        self.P12.copy_(self.T[:, :12])

    @property
    def record_len(self):
        return self.head + 2 * self.nt + 12 * self.S + self.nt + 6 * self.nd + 12 * self.S


    def evaluate(self, w, exponent):
        """Loss (sums) and gradients (dL/dw, dL/d[R|t]) for the current corrected poses -> self.out."""
        if self.icp:
            self.plan.eval(self.P12, self.icp_kind, w if self.icp_kind else None, exponent if self.icp_kind else None, out=self.out)
        else:
            self.plan.eval_native(w, exponent, self.P12, self.out, want_grad=True, want_pose=True)
        return self.out

    def loss_of(self, row):
        """The mean loss train() reports, from a record row."""
        if self.icp:
            return float(row[0])
        return float(row[0] / row[1]) if row[1] > 0 else float('nan')

    @on_device
    def finish(self, w=None, w_m=None, w_v=None, lr_w=0.0, ring=None, totals=None, rec_extra=None):
        """Backward through the pose chain, the optimiser steps, the next iteration's poses (in place); row (step mod rows) of
        ``ring`` [rows, record_len (+ len(rec_extra))] <- {sums, weights, corrections, corrected poses} of THIS iteration (then the
        ``rec_extra`` doubles: the joint sums of the iteration's losses).  Every pointer is the same from call to call (the ring
        slot follows the device step counter): a captured iteration replays correctly."""
        check(lib().dc_pose_train_finish(ptr(self.out), 1 if self.icp else 0, self.nt, self.S, ptr(w), ptr(w_m), ptr(w_v), ptr(self.T0), ptr(self.delta),
                                         ptr(self.d_m), ptr(self.d_v), self.nd, self.zero_first, ptr(self.step), float(lr_w), self.lr,
                                         self.betas[0], self.betas[1], self.eps, ptr(self.T), ptr(ring), 0 if ring is None else ring.shape[0],
                                         ptr(self.T), ptr(self.P12), ptr(totals), ptr(rec_extra), 0 if rec_extra is None else rec_extra.numel(),
                                         stream_ptr()), 'dc_pose_train_finish')

    @staticmethod
    @on_device
    def combine(trainers, totals):
        """totals [2 + P] <- {loss, divisor, dL/dw} over the sequences of one loss (dc_pose_train_combine); the trainers have
        evaluated.  Several sequences: every finish() of the group then takes ``totals``."""
        t0 = trainers[0]
        arr = (ctypes.c_void_p * len(trainers))(*[t.out.data_ptr() for t in trainers])
        check(lib().dc_pose_train_combine(ctypes.cast(arr, ctypes.c_void_p), len(trainers), 1 if t0.icp else 0, t0.nt, ptr(totals),
                                          stream_ptr()), 'dc_pose_train_combine')
        return totals

    def evaluate_finish(self, w, exponent, w_step=None, w_m=None, w_v=None, lr_w=0.0, ring=None, rec_extra=None):
        """``evaluate`` + ``finish`` of an ICP sequence whose loss waits for nobody (one sequence, one rank) in ONE launch
        (dc_icp_sequence_step).  Returns False -- nothing launched -- when this sequence cannot take it."""
        if not self.icp or not self.icp_kind or getattr(self, '_no_fused', False):
            return False
        f = getattr(self, '_fin', None)
        if f is None:
            f = self._fin = nv.PoseTrainStepDesc()
            f.poses0, f.deltas, f.d_m, f.d_v = self.T0.data_ptr(), self.delta.data_ptr(), self.d_m.data_ptr(), self.d_v.data_ptr()
            f.n_deltas, f.zero_first, f.step = self.nd, self.zero_first, self.step.data_ptr()
            f.lr_d, f.beta1, f.beta2, f.eps = self.lr, self.betas[0], self.betas[1], self.eps
            f.poses_used = f.poses_next = self.T.data_ptr()
            f.poses12_next = self.P12.data_ptr()
        p_ = lambda t: None if t is None else t.data_ptr()
        f.w, f.w_m, f.w_v, f.lr_w = p_(w_step), p_(w_m), p_(w_v), float(lr_w)
        f.record, f.ring_rows = p_(ring), (0 if ring is None else ring.shape[0])
        f.record_extra, f.n_record_extra = p_(rec_extra), (0 if rec_extra is None else rec_extra.numel())
        if not self.plan.step(self.P12, self.icp_kind, w, exponent, self.out, f):
            self._no_fused = True
            return False
        return True

    @staticmethod
    def combine2(trainers, val_trainers, totals2):
        """totals2 [2, 2 + P] <- {loss, divisor, dL/dw} of the training sequences' loss and of the validation sequences' loss in
        ONE launch (dc_pose_train_combine2); either list may be empty (zeros).  With sharded sequences ``totals2`` is what the
        iteration's one all-reduce sums over the ranks."""
        return combine_sums([t.out for t in trainers], [v.out for v in val_trainers], totals2,
                            layout=1 if (list(trainers) + list(val_trainers))[0].icp else 0)

    def split_record(self, row):
        """(sums, weights, corrections [nd,6], corrected poses [S,4,4]) of a record row (a CPU tensor)."""
        a = self.head + 2 * self.nt + 12 * self.S
        sums, w = row[:a], row[a:a + self.nt]
        d = row[a + self.nt:a + self.nt + 6 * self.nd].reshape(self.nd, 6)
        P = row[a + self.nt + 6 * self.nd:a + self.nt + 6 * self.nd + 12 * self.S].reshape(self.S, 3, 4)
        T = torch.zeros((self.S, 4, 4), dtype=row.dtype)
        T[:, :3, :] = P
        T[:, 3, 3] = 1.0
        return sums, w, d, T


@on_device
def combine_sums(outs_a, outs_b, totals2, layout=0):
    """totals2 [2, 2 + P] <- the sums {loss, divisor, dL/dw} of two groups of evaluation outputs (dc_pose_train_combine2)."""
    nt = totals2.shape[1] - 2
    need(totals2, (2, 2 + nt), dtype=torch.float64, name='totals2')
    arr_a = (ctypes.c_void_p * max(len(outs_a), 1))(*[o.data_ptr() for o in outs_a])
    arr_b = (ctypes.c_void_p * max(len(outs_b), 1))(*[o.data_ptr() for o in outs_b])
    check(lib().dc_pose_train_combine2(ctypes.cast(arr_a, ctypes.c_void_p), len(outs_a), ctypes.cast(arr_b, ctypes.c_void_p), len(outs_b),
                                       int(layout), nt, ptr(totals2), stream_ptr()), 'dc_pose_train_combine2')
    return totals2


class SequenceTrainer:
    """The per-iteration body of train.py:220-312 for ball neighbourhoods and the min-eigenvalue / trace loss,
    without Python in the loop: dc_sequence_eval (fwd + bwd) and dc_adam_step (torch.optim.Adam semantics) on
    device-resident fp64 parameters.  Sequences are independent: every rank evaluates the sequences it
    owns and one all-reduce of [sum loss, count, dL/dw] per step joins the ranks; all ranks then take the same
    Adam step.  ``evaluate`` / ``adam`` are injectable so the sharding logic can be exercised without a GPU."""

    def __init__(self, plans, w, exponent, poses, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 process_group=None, distributed=False, evaluate=None, adam=None, device=None, chained=False):
        self.plans = list(plans)
        dev = device if device is not None else self.plans[0].device
        self.w = torch.as_tensor(w, dtype=torch.float64).reshape(-1).to(dev).contiguous()
        self.exponent = torch.as_tensor(exponent, dtype=torch.float64).reshape(-1).to(dev).contiguous()
        self.nt = self.w.numel()
        # a single local sequence with the native evaluation and optimiser: one host call, Adam inside the reduction kernel
        self.fused_step = evaluate is None and adam is None and len(self.plans) == 1 and not distributed and self.nt > 0
        self.evaluate = evaluate or (lambda plan, w_, e_, P, out: plan.eval_native(w_, e_, P, out))
        self.adam = adam or self._adam_native
        self.poses12 = [p.poses12(T) if hasattr(p, 'poses12') else T for p, T in zip(self.plans, poses)]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.w)
        self.exp_avg_sq = torch.zeros_like(self.w)
        self.t = 0
        # chained steps (single local sequence): every launch also finishes the previous step, so step() returns the sums
        # of the PREVIOUS evaluation and flush() those of the last one; see dc_sequence_step_chained
        self.chained = bool(chained) and self.fused_step
        # several ranks (or an injected reduction) and one local sequence: the update of step t rides in the launch of step
        # t + 1 (dc_sequence_eval_after_update), three launches per step instead of four; the returned sums are current, only
        # the weights lag by the one update flush() applies
        self.update_in_next = (bool(chained) and not self.fused_step and evaluate is None and adam is None
                               and len(self.plans) >= 1 and self.nt > 0 and (distributed or len(self.plans) == 1))
        # several local sequences in one loss, one rank: a chain over the sequences -- ONE launch per sequence and step, each launch
        # finishing the one before it (dc_sequence_step_linked); step() returns the PREVIOUS step's sums like ``chained``
        self.linked = (bool(chained) and evaluate is None and adam is None and len(self.plans) > 1 and self.nt > 0 and not distributed)
        self._pending = False
        self._grad_prev = None         # update_in_next: the summed gradient of the last step when the caller reduced it (use_sums)
        self.ready = torch.zeros((16,), dtype=torch.int32, device=dev)      # 64 bytes: the chain's published weights
        self.outs = [torch.zeros((2 + 2 * self.nt + 12 * p.n_scans,), dtype=torch.float64, device=dev) for p in self.plans]
        self.link_acc = torch.zeros((max(o.numel() for o in self.outs),), dtype=torch.float64, device=dev)      # linked: the step's running sums
        self.acc = torch.zeros((2 + self.nt,), dtype=torch.float64, device=dev)       # [sum loss, count, dL/dw]
        self.distributed, self.group = distributed, process_group
        self.count = float(sum(p.count for p in self.plans))
        if distributed:
            from .distributed import all_reduce_sum
            self.count = float(all_reduce_sum(torch.tensor([self.count], dtype=torch.float64, device=dev), process_group).item())

    @on_device
    def _adam_native(self, grad_sum):
        check(lib().dc_adam_step(ptr(self.w), ctypes.c_void_p(grad_sum.data_ptr()), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                 self.nt, self.t, self._grad_scale(), self.lr, self.betas[0], self.betas[1], self.eps,
                                 self.weight_decay, stream_ptr()), 'dc_adam_step')

    def flush(self, out=None):
        """Chained mode: finish the last launched step (its sums -> ``out`` or the trainer's own buffer, and its Adam
        update); returns the sums.  No-op otherwise."""
        buf = self.outs[0] if out is None else out
        if self._pending and self.linked:
            S = len(self.plans)
            self.plans[-1].flush_linked(self.t & 1, self.link_acc, self.w, self.exp_avg, self.exp_avg_sq, self.t, self.t * S + S + 1,
                                        self.ready, buf if buf.numel() == self.outs[-1].numel() else self.outs[-1], self._grad_scale(), self.lr,
                                        self.betas, self.eps, self.weight_decay)
            if buf.numel() != self.outs[-1].numel():
                buf[:2 + self.nt].copy_(self.outs[-1][:2 + self.nt])
            self._pending = False
            return buf[:2 + self.nt]
        if self._pending and self.update_in_next:
            g = self._grad_prev if self._grad_prev is not None else (self.acc if len(self.plans) > 1 else self.outs[0])[2:2 + self.nt]
            self.adam(g)                                          # the last evaluation's (all-reduced) gradient, step self.t
            self._pending = False
        elif self._pending:
            self.plans[0].chain_flush(self.w, buf, self.exp_avg, self.exp_avg_sq, self.t, self._grad_scale(), self.lr,
                                      self.betas, self.eps, self.weight_decay)
            self._pending = False
        return buf[:2 + self.nt] if (len(self.outs) == 1 or self.linked) else self.acc

    def use_sums(self, acc):
        """``acc`` [2 + P] = {sum loss, count, dL/dw} of the last step over all sequences and ranks (step(defer_reduce=True)): the
        gradient the next launch's Adam update takes."""
        need(acc, (2 + self.nt,), dtype=torch.float64, name='acc', device=self.w.device)
        self._grad_prev = acc[2:]

    def _grad_scale(self):
        """1 / number of masked points of all sequences (the mean reduction, loss.py:205-213); no masked point at all:
        NaN, the mean of an empty tensor, as in the reference."""
        return 1.0 / self.count if self.count > 0 else float('nan')

    def step(self, out_prev=None, w_used_prev=None, require_chain=False, defer_reduce=False):
        """One optimisation step; returns the device tensor [sum loss, count, dL/dw...] summed over sequences / ranks
        (mean loss = acc[0] / acc[1]).  No host synchronisation.  Chained mode only: ``out_prev`` (fp64 [2 + 2P + 12S]) takes
        the PREVIOUS evaluation's sums instead of the trainer's own buffer and ``w_used_prev`` (fp64 [P]) the weights that
        evaluation used -- a training log's record of an iteration, written by the launch itself.  ``update_in_next`` mode: the
        sums are current, ``w_used_prev`` takes the weights THIS evaluation uses; ``defer_reduce``: the sequences' sums stay in
        ``self.outs`` and the caller hands the joint (all-reduced) sums back with use_sums() before the next step."""
        if self.linked:
            T, S = self.t + 1, len(self.plans)
            buf = self.outs[0] if out_prev is None else out_prev
            for i, (plan, P) in enumerate(zip(self.plans, self.poses12)):
                if i == 0:                       # finishes the last sequence of step T - 1: its totals, the Adam update, the record
                    ok = plan.step_linked(self.plans[-1], (T - 1) & 1, 2 if self._pending else 0, self.link_acc if self._pending else None,
                                          self.w, self.exponent, P, self.exp_avg, self.exp_avg_sq, T, T * S + 1, self.ready, buf,
                                          self._grad_scale(), self.lr, self.betas, self.eps, self.weight_decay, w_used_prev=w_used_prev)
                else:                            # finishes sequence i - 1 of this step: its sums onto the step's running sums
                    ok = plan.step_linked(self.plans[i - 1], T & 1, 1, None if i == 1 else self.link_acc, self.w, self.exponent, P,
                                          self.exp_avg, self.exp_avg_sq, T, T * S + 1 + i, self.ready, self.link_acc,
                                          self._grad_scale(), self.lr, self.betas, self.eps, self.weight_decay)
                if not ok:
                    if i or self._pending:
                        raise RuntimeError('a sequence plan refused a linked step in the middle of a chain')
                    self.linked = False
                    break
            if self.linked:
                self.t += 1
                self._pending = True
                return buf[:2 + self.nt]
            if require_chain:
                return None                      # (nothing was launched)
        if self.chained:
            buf = self.outs[0] if out_prev is None else out_prev
            ok = self.plans[0].step_chained(self.w, self.exponent, self.poses12[0], buf, self.exp_avg, self.exp_avg_sq,
                                            self.t + 1, self._pending, self.ready, self._grad_scale(), self.lr, self.betas, self.eps,
                                            self.weight_decay, w_used_prev=w_used_prev)
            if ok:
                self.t += 1
                self._pending = True
                return buf[:2 + self.nt]
            self.flush()
            self.chained = False                 # this plan cannot chain: ordinary steps from here on
            if require_chain:
                return None                      # (nothing was launched)
        if self.fused_step:
            self.t += 1
            out = self.plans[0].step_native(self.w, self.exponent, self.poses12[0], self.outs[0], self.exp_avg,
                                            self.exp_avg_sq, self.t, self._grad_scale(), self.lr, self.betas, self.eps,
                                            self.weight_decay)
            return out[:2 + self.nt]
        if self.update_in_next:
            several = len(self.plans) > 1
            acc = self.acc if several else self.outs[0][:2 + self.nt]
            # (the launch reads the gradient before the reduction that follows it on the stream overwrites `out`)
            grad = (self._grad_prev if self._grad_prev is not None else acc[2:]) if self._pending else None
            ok = self.plans[0].eval_after_update(self.w, self.exponent, self.poses12[0], self.outs[0], self.exp_avg, self.exp_avg_sq,
                                                 self.t + 1, grad, self.ready, self._grad_scale(),
                                                 self.lr, self.betas, self.eps, self.weight_decay, w_used=w_used_prev)
            if ok:
                # the other local sequences follow on the stream: their launches start after the first one has published
                # the updated weights (kernels of one stream run in order)
                for plan, P, out in zip(self.plans[1:], self.poses12[1:], self.outs[1:]):
                    self.evaluate(plan, self.w, self.exponent, P, out)
                self.t += 1
                self._pending = True
                if defer_reduce:
                    # the caller joins self.outs (with whatever else the iteration's ONE all-reduce carries) and hands the summed
                    # gradient back through use_sums()
                    return self.outs
                self._grad_prev = None
                if several:
                    acc.zero_()
                    for o in self.outs:
                        acc += o[:2 + self.nt]
                if self.distributed:
                    from .distributed import all_reduce_sum
                    all_reduce_sum(acc, self.group)
                return acc
            self.flush()
            self.update_in_next = False
            if require_chain:
                return None                      # (nothing was launched)
        for plan, P, out in zip(self.plans, self.poses12, self.outs):
            self.evaluate(plan, self.w, self.exponent, P, out)
        if len(self.outs) == 1:
            acc = self.outs[0][:2 + self.nt]          # reduced in place across ranks below
        else:
            acc = self.acc
            acc.zero_()
            for o in self.outs:
                acc += o[:2 + self.nt]
        if self.distributed:
            from .distributed import all_reduce_sum
            all_reduce_sum(acc, self.group)
        self.t += 1
        self.adam(acc[2:2 + self.nt])
        return acc
