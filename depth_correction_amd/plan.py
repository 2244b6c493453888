"""Fused per-sequence map-consistency evaluation (the MI355X hot path).

One ``SequencePlan`` holds everything that is constant over the optimisation of one sequence
(train.py:94-215 builds exactly these once: local feature clouds, global neighbourhoods, masks) laid out
for the GPU, and evaluates one training iteration (eval.py:85-112: model -> pose transform -> concatenate
-> neighbourhood features -> loss, plus its backward) in three kernels:

    dc_points_fwd       corrected, posed points of all scans            (K1-K3)
    dc_consistency_fwd  covariance, smallest eigenpair, loss, bwd record (K5-K15)
    dc_consistency_bwd  dL/dx gather + dL/dw, dL/dexponent, dL/dpose     (K18)

Layout decisions:
  * points are permuted once into Morton order of the initial global cloud, so the K neighbours of
    consecutive lanes share cache lines (the reference's scan-major order has no locality at all);
  * float32 clouds keep their intermediate points / means as 32-bit fixed point (DC_Q32), float64 clouds
    as float64 -- on-chip arithmetic is fp64 either way;
  * the backward runs over the transposed neighbour list (no atomics, bitwise reproducible).
Results are reported in the caller's point order; per-point outputs are un-permuted lazily.
"""
from __future__ import annotations

import torch

from . import ops
from ._native import need

__all__ = ['SequencePlan', 'consistency_loss']


class SequencePlan:
    def __init__(self, clouds, poses, neighbors, mask=None, model_kind='ScaledPolynomial', loss='min_eigval_loss',
                 normalization=True, sqrt=False, spatial_sort=True, point_format='auto'):
        """
        :param clouds: per-scan dicts / objects with vps [n,3], dirs [n,3], depth [n,1], inc_angles [n,1], mask [n]
                       (local feature clouds, sensor frame), GPU tensors of one dtype.
        :param poses: [S,4,4] initial scan poses (used for the layout only; every evaluation takes its own).
        :param neighbors: [N,K] int neighbour indices of the global cloud (scan-major order, -1 = missing).
        :param mask: [N] bool global mask (None = all points).
        """
        get = (lambda c, f: c[f]) if isinstance(clouds[0], dict) else getattr
        vps = torch.cat([get(c, 'vps') for c in clouds]).contiguous()
        dirs = torch.cat([get(c, 'dirs') for c in clouds]).contiguous()
        depth = torch.cat([get(c, 'depth').reshape(-1) for c in clouds]).contiguous()
        incs = [get(c, 'inc_angles') for c in clouds]
        inc = None if incs[0] is None else torch.cat([i.reshape(-1) for i in incs]).contiguous()
        lms = [get(c, 'mask') for c in clouds]
        dev = dirs.device
        sizes = [len(get(c, 'dirs')) for c in clouds]
        lmask = None if lms[0] is None else torch.cat(lms).contiguous()
        scan_id = torch.repeat_interleave(torch.arange(len(clouds), dtype=torch.int32, device=dev),
                                          torch.as_tensor(sizes, device=dev))
        self.n, self.n_scans, self.device, self.dtype = dirs.shape[0], len(clouds), dev, dirs.dtype
        self.sizes = sizes
        self.model_kind, self.loss, self.normalization, self.sqrt = model_kind, loss, bool(normalization), bool(sqrt)
        nbr = ops.as_index32(neighbors)
        need(nbr, (self.n, None), dtype=torch.int32, name='neighbors', device=dev)
        self.k = nbr.shape[1]
        if mask is not None:
            need(mask, (self.n,), dtype=torch.bool, name='mask', device=dev)

        # ---- layout: Morton order of the initial global cloud ----------------------------------
        ps0 = ops.PointSet(vps, dirs, depth, inc, lmask, scan_id)
        P0 = self.poses12(poses)
        x0 = ops.points_fwd(ps0, P0)
        if spatial_sort and self.n > 1:
            order = ops.spatial_order(x0).long()
            rank = torch.empty_like(order)
            rank[order] = torch.arange(self.n, device=dev)
            nbr_l = nbr.long()[order]
            nbr = torch.where(nbr_l >= 0, rank[nbr_l.clamp(min=0)], nbr_l).to(torch.int32).contiguous()
            vps, dirs, depth = vps[order].contiguous(), dirs[order].contiguous(), depth[order].contiguous()
            inc = None if inc is None else inc[order].contiguous()
            lmask = None if lmask is None else lmask[order].contiguous()
            scan_id = scan_id[order].contiguous()
            mask = None if mask is None else mask[order].contiguous()
            self.order, self.rank = order, rank
        else:
            self.order = self.rank = None
        self.ps = ops.PointSet(vps, dirs, depth, inc, lmask, scan_id)
        self.nbr, self.mask = nbr, mask
        self.csr_ptr, self.csr_src = ops.knn_transpose(nbr)
        self.count = float(self.n if mask is None else int(mask.sum().item()))

        # ---- internal point format -------------------------------------------------------------------
        if point_format == 'auto':
            point_format = 'q32' if self.dtype == torch.float32 else 'float'
        self.qfmt = None
        if point_format == 'q32':
            lo, hi = x0.min(0).values.tolist(), x0.max(0).values.tolist()
            self.qfmt = ops.QFormat.for_extent(lo, hi)
        pdt = torch.int32 if self.qfmt is not None else self.dtype
        self.x = torch.empty((self.n, 4), dtype=pdt, device=dev)
        self.rec = torch.empty((self.n, 8), dtype=pdt, device=dev)
        rows = ops.lib().dc_partial_rows(self.n)
        nacc = 2 * ops.nv.MAX_MODEL_TERMS + 12 * self.n_scans
        self.partials = torch.empty((rows * max(nacc, 2),), dtype=torch.float64, device=dev)
        self.version = 0

    # ------------------------------------------------------------------------------------------------
    def poses12(self, poses):
        poses = torch.as_tensor(poses, device=self.device) if not isinstance(poses, torch.Tensor) else poses
        assert poses.shape == (self.n_scans, 4, 4), poses.shape
        return poses.detach().to(device=self.device, dtype=torch.float64)[:, :3, :].reshape(self.n_scans, 12).contiguous()

    def forward(self, w, exponent, poses, want_pointwise=False, want_eigvals=False):
        """One forward evaluation; returns dict(sums=[sum of pointwise loss over the mask, mask count], ...)."""
        self.P = self.poses12(poses)
        self.w = None if w is None else w.detach().reshape(-1).to(torch.float64).contiguous()
        self.e = None if exponent is None else exponent.detach().reshape(-1).to(device=self.device, dtype=torch.float64).contiguous()
        kind = self.model_kind if self.w is not None else None
        ops.points_fwd(self.ps, self.P, kind, self.w, self.e, stride=4, qfmt=self.qfmt, out=self.x)
        out = ops.consistency_fwd(self.x, self.nbr, mask=self.mask, loss=self.loss, normalization=self.normalization,
                                  sqrt=self.sqrt, rec=self.rec, want_pointwise=want_pointwise,
                                  want_eigvals=want_eigvals, partials=self.partials, qfmt=self.qfmt)
        self.version += 1
        return out

    def backward(self, want_exponent=False, want_pose=False):
        """Gradients of the *sum* of the pointwise loss over the mask w.r.t. (w [P], exponent [P], [R|t] [S,3,4])."""
        kind = self.model_kind if self.w is not None else None
        _, grads = ops.consistency_bwd(self.x, self.rec, self.csr_ptr, self.csr_src, self.ps, self.P, kind, self.w, self.e,
                                       want_exponent=want_exponent, want_pose=want_pose, partials=self.partials,
                                       qfmt=self.qfmt)
        return grads

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


class _ConsistencyLoss(torch.autograd.Function):
    """sum over the mask of the pointwise loss of one sequence, differentiable w.r.t. w, exponent, poses."""

    @staticmethod
    def forward(ctx, plan, w, exponent, poses):
        out = plan.forward(w, exponent, poses)
        ctx.plan, ctx.version = plan, plan.version
        ctx.w_shape = None if w is None else w.shape
        ctx.w_dtype = None if w is None else w.dtype
        ctx.e_info = (exponent.shape, exponent.dtype, exponent.device) if isinstance(exponent, torch.Tensor) else None
        ctx.p_info = (poses.dtype, poses.device)
        return out['sums'][0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        plan = ctx.plan
        if plan.version != ctx.version:
            raise RuntimeError('SequencePlan was evaluated again before backward(); its buffers were overwritten')
        need_w, need_e, need_p = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        gw, ge, gT = plan.backward(want_exponent=need_e, want_pose=need_p)
        out_w = (grad_out * gw).reshape(ctx.w_shape).to(ctx.w_dtype) if need_w else None
        out_e = None
        if need_e:
            shp, dt, dv = ctx.e_info
            out_e = (grad_out * ge).reshape(shp).to(device=dv, dtype=dt)
        out_p = None
        if need_p:
            g = torch.zeros((plan.n_scans, 4, 4), dtype=torch.float64, device=plan.device)
            g[:, :3, :] = grad_out * gT
            out_p = g.to(device=ctx.p_info[1], dtype=ctx.p_info[0])
        return None, out_w, out_e, out_p


def consistency_loss(plan, w, exponent, poses):
    """(sum of pointwise loss over the mask, mask count) of one sequence; the sum carries the autograd graph."""
    return _ConsistencyLoss.apply(plan, w, exponent, poses), plan.count
