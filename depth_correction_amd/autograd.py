"""torch.autograd bridges of the HIP operators used by the reference-API layer (un-fused path).

``neighborhood_features`` is DepthCloud.update_features (depth_cloud.py:426-433) as ONE op: forward =
dc_features_fwd (gather + mean + covariance + eigh + normals + incidence angles, nothing materialised),
backward = dc_features_bwd (hand-derived, over the transposed neighbour list).  Gradients flow from
``mean``, ``cov`` and ``eigvals`` to ``points`` -- the only paths the reference's losses use;
eigenvectors / normals / incidence angles are returned detached, as constants of the graph.
"""
from __future__ import annotations

import torch

from . import ops

__all__ = ['neighborhood_features', 'NeighborhoodGraph']


class NeighborhoodGraph:
    """int32 neighbour table + its transpose, built once per neighbourhood set and cached on the cloud."""

    def __init__(self, neighbors, nbr=None):
        self.version = neighbors._version
        self.nbr = ops.as_index32(neighbors) if nbr is None else nbr        # nbr: the int32 table `neighbors` was made from
        self._csr = None

    @staticmethod
    def of(neighbors):
        """Graph of a neighbour tensor, cached ON the tensor object (the same ``neighbors`` tensor is re-attached to a
        fresh global cloud every iteration, preproc.py:214); in-place edits (concatenate's index shift) invalidate."""
        g = getattr(neighbors, '_dc_graph', None)
        if g is None or g.version != neighbors._version:
            g = NeighborhoodGraph(neighbors)
            neighbors._dc_graph = g
        return g

    @property
    def csr(self):
        if self._csr is None:
            self._csr = ops.knn_transpose(self.nbr)
        return self._csr


class _Features(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, graph, dirs, mean_weights, scale):
        x = points.detach().contiguous()
        f = ops.features_fwd(x, graph.nbr, dirs=None if dirs is None else dirs.detach().contiguous(),
                             mean_weights=None if mean_weights is None else mean_weights.detach().contiguous(),
                             scale=scale, want_saved=True, want_weights=True,
                             want=('mean', 'cov', 'eigvals', 'eigvecs') + (('normals', 'inc_angles') if dirs is not None else ()))
        ctx.graph, ctx.scale = graph, scale
        ctx.save_for_backward(x, f['cmean'], f['invd'], f['nvalid'], f['eigvecs'])
        k = graph.nbr.shape[1]
        outs = (f['mean'], f['cov'], f['eigvals'], f['eigvecs'], f['normals'], f['inc_angles'],
                f['weights'].reshape(-1, k, 1))
        ctx.mark_non_differentiable(*[o for o in outs[3:] if o is not None])
        return outs

    @staticmethod
    def backward(ctx, g_mean, g_cov, g_eig, *unused):
        if ctx.scale:
            raise NotImplementedError('backward through distance-scaled neighbour weights (nn_scale) is not implemented')
        x, cmean, invd, nvalid, eigvecs = ctx.saved_tensors
        cont = lambda t: None if t is None else t.contiguous()
        csr_ptr, csr_src = ctx.graph.csr
        gp = ops.features_bwd(x, csr_ptr, csr_src, cmean, invd, nvalid, eigvecs=eigvecs, grad_mean=cont(g_mean),
                              grad_cov=cont(g_cov), grad_eigvals=cont(g_eig))
        return gp, None, None, None, None


def neighborhood_features(points, graph, dirs=None, mean_weights=None, scale=None):
    """(mean, cov, eigvals, eigvecs, normals, inc_angles, weights [N,K,1]) of the neighbourhoods in ``graph``."""
    return _Features.apply(points, graph, dirs, mean_weights, scale)
