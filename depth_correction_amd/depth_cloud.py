"""DepthCloud with the reference's API (depth_cloud.py:18-740), backed by the HIP operators.

A cloud is (viewpoints, unit directions, depth); everything else is derived.  The container semantics the
reference's callers rely on are kept (SURVEY 7 "API fidelity quirks"): methods mutate ``self`` and return
None, tensors are never modified in place (a shallow ``copy()`` is a snapshot), ``transform`` keeps only
``mask`` and ``normals``, ``concatenate(dependent=True)`` shifts neighbour indices, ``cloud[mask]`` slices
the per-point fields only.

What differs is where the arithmetic runs:
  * ``update_neighbors`` / ``update_dir_neighbors``  -> GPU grid search (dc_knn_build / dc_radius_*), bit-exact
    with the cKDTree ordering;
  * ``update_mean`` / ``update_weights`` / ``update_cov`` / ``update_eig`` / ``update_normals`` /
    ``update_incidence_angles`` (= ``update_features``)  -> ONE fused kernel (dc_features_fwd) whose results
    are cached per (points, neighbours, weights, scale) and handed out field by field, with a hand-derived
    backward (dc_features_bwd) instead of autograd over ``[N,K,3,3]`` tensors;
  * ``vp_dispersion`` / ``dir_dispersion``  -> dc_dispersion.
``update_points`` / ``transform`` are the reference's one-line tensor expressions (plain torch, any device).
The neighbourhood operators need GPU tensors and raise otherwise: there is no CPU implementation.
ROS / open3d conveniences (to_msg, visualize, estimate_normals, to_mesh) are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch
from numpy.lib.recfunctions import merge_arrays, structured_to_unstructured, unstructured_to_structured

from . import ops
from .autograd import NeighborhoodGraph, neighborhood_features
from .nearest_neighbors import ball_angle_to_distance, nearest_neighbors

__all__ = ['DepthCloud']

_STALE = object()        # marker: distances requested (keep_neighbors) but not materialised yet


class DepthCloud(object):
    source_fields = ['vps', 'dirs', 'depth']
    sliced_fields = source_fields + ['points', 'mean', 'cov', 'eigvals', 'eigvecs', 'normals', 'inc_angles', 'trace',
                                     'loss', 'mask']
    not_sliced_fields = ['neighbors', 'weights', 'distances', 'neighbor_points',
                         'dir_neighbors', 'dir_neighbor_weights', 'dir_distances']
    all_fields = sliced_fields + not_sliced_fields

    def __init__(self, vps=None, dirs=None, depth=None, points=None, mean=None, cov=None, eigvals=None, eigvecs=None,
                 normals=None, inc_angles=None, trace=None, loss=None, mask=None, neighbors=None, distances=None,
                 neighbor_points=None, weights=None, dir_neighbors=None, dir_neighbor_weights=None, dir_distances=None):
        if vps is None:
            vps = torch.zeros((1, 3))
        assert isinstance(vps, torch.Tensor) and vps.shape[-1] == 3
        assert isinstance(dirs, torch.Tensor) and dirs.shape[-1] == 3
        assert dirs.shape == vps.shape or vps.shape == (1, 3)
        assert isinstance(depth, torch.Tensor) and depth.shape[-1] == 1 and depth.shape[:-1] == dirs.shape[:-1]
        self.vps, self.dirs, self.depth = vps, dirs, depth
        self.points, self.mean, self.cov, self.eigvals, self.eigvecs = points, mean, cov, eigvals, eigvecs
        self.normals, self.inc_angles, self.trace, self.loss, self.mask = normals, inc_angles, trace, loss, mask
        self.neighbors, self.weights, self._distances, self.neighbor_points = neighbors, weights, distances, neighbor_points
        self.dir_neighbors, self.dir_neighbor_weights, self.dir_distances = dir_neighbors, dir_neighbor_weights, dir_distances
        self._feat = None           # (key, outputs) of the last fused feature evaluation

    # ---- container ---------------------------------------------------------------------------------
    @property
    def distances(self):
        if self._distances is _STALE:
            x = self.get_points()
            self._distances = torch.linalg.norm(x.unsqueeze(dim=1) - x[self.neighbors], dim=-1)
        return self._distances

    @distances.setter
    def distances(self, value):
        self._distances = value

    def _raw(self, f):
        return self._distances if f == 'distances' else getattr(self, f)

    def _present(self):
        return {f: self._raw(f) for f in DepthCloud.all_fields if self._raw(f) is not None}

    def copy(self):
        return DepthCloud(**self._present())

    def clone(self):
        return DepthCloud(**{f: (x if x is _STALE else x.clone()) for f, x in self._present().items()})

    def size(self):
        return self.dirs.shape[0]

    __len__ = size

    def device(self):
        return self.depth.device

    def __getitem__(self, item):
        if isinstance(item, list) and len(item) > 0 and isinstance(item[0], str):
            return DepthCloud(**{f: getattr(self, f) for f in item})
        if isinstance(item, torch.Tensor) and item.dtype == torch.bool and item.dim() == 1 and item.is_cuda:
            # x[mask] runs a count + a partition + a gather per FIELD; here every field's kept rows are placed by one pair of launches
            # (dc_compact_rows) -- outside autograd; fields that carry a gradient take the index_select they can differentiate
            names = [f for f in DepthCloud.sliced_fields if getattr(self, f) is not None]
            fields = [getattr(self, f) for f in names]
            n = len(self)
            plain = [f.shape[0] == n and not (f.requires_grad and torch.is_grad_enabled()) for f in fields]
            if item.shape[0] == n and all(plain):
                kept = []
                for c0 in range(0, len(fields), 8):              # (eight arrays per call)
                    kept += ops.compact_rows(item, [f.detach() for f in fields[c0:c0 + 8]])
                return DepthCloud(**dict(zip(names, kept)))
            item = item.nonzero().squeeze(1)
            return DepthCloud(**{f: x.index_select(0, item) if x.shape[0] == n else x[item] for f, x in zip(names, fields)})
        return DepthCloud(**{f: getattr(self, f)[item] for f in DepthCloud.sliced_fields if getattr(self, f) is not None})

    def __add__(self, other):
        return DepthCloud.concatenate([self, other], dependent=True)

    def to(self, device=None, dtype=None, float_type=None, int_type=None):
        out = {}
        for f, x in self._present().items():
            if x is _STALE:
                out[f] = x
                continue
            is_f = x.dtype.is_floating_point
            t = None
            if (float_type and is_f) or (int_type and not is_f) or (dtype and dtype.is_floating_point == is_f):
                t = dtype or float_type or int_type
            out[f] = x.to(device=device, dtype=t)
        return DepthCloud(**out)

    def cpu(self):
        return self.to(torch.device('cpu'))

    def gpu(self):
        return self.to(torch.device('cuda:0'))

    def type(self, dtype=None):
        if dtype is None:
            assert self.vps.dtype == self.dirs.dtype == self.depth.dtype
            return self.vps.dtype
        for f, x in self._present().items():
            if x is not _STALE and dtype.is_floating_point == x.dtype.is_floating_point:
                setattr(self, f, x.type(dtype))
        self._feat = None
        return self

    def float(self):
        return self.type(torch.float32)

    def double(self):
        return self.type(torch.float64)

    def detach(self):
        for f, x in self._present().items():
            if x is not _STALE:
                setattr(self, f, x.detach())
        return self

    @staticmethod
    def concatenate(clouds, fields=None, dependent=False):
        if not fields:
            fields = DepthCloud.all_fields if dependent else DepthCloud.source_fields
        else:
            assert not dependent
        offsets = np.concatenate([[0], np.cumsum([len(c) for c in clouds])[:-1]])
        out = {}
        for f in fields:
            parts = [c._raw(f) for c in clouds]
            if any(p is _STALE for p in parts):
                parts = [getattr(c, f) for c in clouds]
            have = [p is not None for p in parts]
            if all(have):
                if f in ('dir_neighbors', 'neighbors'):
                    # as the reference: indices become global, shifted IN PLACE (depth_cloud.py:555-559)
                    for p, off in zip(parts, offsets):
                        p += int(off)
                out[f] = torch.cat(parts)
            elif any(have):
                print('Field %s not available for %i of %i clouds.' % (f, sum(have), len(clouds)))
        return DepthCloud(**out)

    @staticmethod
    def from_structured_array(arr, dtype=None, device=None):
        assert isinstance(arr, np.ndarray)
        grab = lambda names: structured_to_unstructured(arr[names], dtype=dtype)
        pts = grab(['x', 'y', 'z'])
        vps = grab(['vp_x', 'vp_y', 'vp_z']) if 'vp_x' in arr.dtype.names else None
        normals = grab(['normal_x', 'normal_y', 'normal_z']) if 'normal_x' in arr.dtype.names else None
        return DepthCloud.from_points(pts, vps=vps, normals=normals, device=device)

    @staticmethod
    def from_points(pts, vps=None, normals=None, dtype=None, device=None):
        """Cloud from end points (and viewpoints): dirs = (pts - vps) / |pts - vps|, zero-depth rays left as is."""
        if getattr(getattr(pts, 'dtype', None), 'names', None):
            return DepthCloud.from_structured_array(pts, device=device)
        if isinstance(dtype, type) or isinstance(dtype, np.dtype):       # numpy dtype (cfg.numpy_float_type())
            dtype = getattr(torch, np.dtype(dtype).name)
        pts = torch.as_tensor(pts, device=device)
        if pts.is_cuda and pts.dim() == 2 and pts.shape[1] == 3 and pts.dtype in (torch.float32, torch.float64) \
                and (dtype is None or dtype in (torch.float32, torch.float64)) and not pts.requires_grad:
            # device tensors: one kernel (dc_cloud_from_points) instead of five elementwise passes
            vps_t = None if vps is None else torch.as_tensor(vps, dtype=pts.dtype, device=pts.device).contiguous()
            assert vps_t is None or vps_t.shape == pts.shape
            v, dirs, depth, _ = ops.cloud_from_points(pts.contiguous(), vps_t, dtype=dtype or pts.dtype, want_zero_vps=True)
            kwargs = dict(vps=v, dirs=dirs, depth=depth)
        else:
            pts = pts.to(dtype) if dtype is not None else pts
            vps = torch.zeros_like(pts) if vps is None else torch.as_tensor(vps, dtype=dtype, device=device)
            assert vps.shape == pts.shape
            rays = pts - vps
            depth = rays.norm(dim=-1, keepdim=True)
            dirs = torch.where(depth > 0.0, rays / depth, rays)
            kwargs = dict(vps=vps, dirs=dirs, depth=depth)
        if normals is not None:
            kwargs['normals'] = torch.as_tensor(normals, dtype=kwargs['dirs'].dtype, device=kwargs['dirs'].device)
        dc = DepthCloud(**kwargs)
        return dc.to(device=device) if device else dc

    def to_structured_array(self, colors=None):
        f32 = lambda t: np.asarray(t.detach().cpu().numpy(), dtype=np.float32)
        parts = [unstructured_to_structured(f32(self.get_points()), names=['x', 'y', 'z']),
                 unstructured_to_structured(f32(self.vps.expand_as(self.dirs)), names=['vp_x', 'vp_y', 'vp_z'])]
        if self.normals is not None:
            parts.append(unstructured_to_structured(f32(self.normals), names=['normal_x', 'normal_y', 'normal_z']))
        if self.inc_angles is not None:
            parts.append(unstructured_to_structured(f32(self.inc_angles), names=['inc_angle']))
        if self.loss is not None:
            parts.append(unstructured_to_structured(f32(self.loss).reshape(-1, 1), names=['loss']))
        if self.mask is not None:
            parts.append(unstructured_to_structured(self.mask.detach().cpu().numpy().astype(np.uint8).reshape(-1, 1), names=['mask']))
        if colors is not None:
            parts.append(unstructured_to_structured(np.asarray(colors, dtype=np.float32), names=['r', 'g', 'b']))
        return merge_arrays(parts, flatten=True)

    # ---- K2 / K3 ------------------------------------------------------------------------------------
    def to_points(self):
        v, d, r = self.vps, self.dirs, self.depth
        if d.is_cuda and d.dtype in (torch.float32, torch.float64) and v.dtype == d.dtype == r.dtype and d.dim() == 2 and d.is_contiguous() \
                and v.is_contiguous() and r.is_contiguous() and v.device == d.device == r.device \
                and not (torch.is_grad_enabled() and (v.requires_grad or d.requires_grad or r.requires_grad)):
            return ops.to_points(v, d, r)          # the same two roundings in one kernel (dc_to_points)
        return self.vps + self.depth * self.dirs

    def update_points(self):
        self.points = self.to_points()
        self.neighbor_points = None
        self._feat = None

    def get_points(self):
        if self.points is None:
            self.update_points()
        return self.points

    def transform(self, T):
        assert isinstance(T, torch.Tensor) and T.shape == (4, 4)
        T = T.to(dtype=self.vps.dtype, device=self.vps.device)
        Rt, t = T[:3, :3].t(), T[:3, 3:].t()
        kwargs = {'mask': self.mask}
        if self.normals is not None:
            kwargs['normals'] = torch.matmul(self.normals, Rt)
        return DepthCloud(torch.matmul(self.vps, Rt) + t, torch.matmul(self.dirs, Rt), self.depth, **kwargs)

    # ---- K4 neighbourhoods ---------------------------------------------------------------------------
    def valid_neighbor_mask(self):
        assert self.neighbors is not None
        return self.neighbors >= 0

    def update_neighbors(self, k=None, r=None, _weights=True):
        assert self.points is not None
        self._distances, self.neighbors = nearest_neighbors(self.get_points(), self.get_points(), k=k, r=r)
        self.neighbor_points = None
        self._feat = None
        if not _weights:            # update_all: update_features writes the same validity weights a moment later
            self.weights = None
            return
        nbr = self.graph().nbr
        # valid_neighbor_mask() read off the int32 table (one kernel on the device)
        self.weights = ops.valid_weights(nbr) if nbr.is_cuda and nbr.is_contiguous() else (nbr >= 0).float()[..., None]
        self.weights._dc_validity = True
        self.neighbor_points = None
        self._feat = None

    def update_dir_neighbors(self, k=None, r=None, angle=None):
        assert self.dirs is not None
        if angle is not None:
            assert r is None
            r = float(ball_angle_to_distance(torch.as_tensor(angle)))
        self.dir_distances, self.dir_neighbors = nearest_neighbors(self.dirs, self.dirs, k=k, r=r)
        self.dir_neighbor_weights = (self.dir_neighbors >= 0).float()

    def update_distances(self):
        assert self.neighbors is not None
        self._distances = _STALE           # computed on first read: no loss consumes them

    def collect_neighbors(self, item):
        idx = self.neighbors[item].unique()
        return idx[idx >= 0]

    def filter_with_neighbors(self, item):
        return self[self.collect_neighbors(item)]

    def compute_neighbor_points(self):
        return self.get_points()[self.neighbors]

    def update_neighbor_points(self):
        self.neighbor_points = self.compute_neighbor_points()

    def get_neighbor_points(self):
        if self.neighbor_points is None:
            self.update_neighbor_points()
        return self.neighbor_points

    def graph(self):
        assert self.neighbors is not None
        return NeighborhoodGraph.of(self.neighbors)

    # ---- K5-K12 neighbourhood features (fused) --------------------------------------------------------------
    def _features(self, scale=None):
        x = self.get_points()
        w = self.weights
        key = (id(x), id(self.neighbors), id(w), scale, id(self.dirs))
        if self._feat is None or self._feat[0] != key:
            if not x.is_cuda:
                raise RuntimeError('DepthCloud neighbourhood features need GPU tensors (no CPU path): use cloud.gpu()')
            mw = None
            if w is not None and not getattr(w, '_dc_validity', False):
                mw = w.reshape(len(self), -1).to(x.dtype)        # explicit neighbour weights only enter the mean
            dirs = self.dirs if self.dirs.shape == x.shape else self.dirs.expand_as(x)
            names = ('mean', 'cov', 'eigvals', 'eigvecs', 'normals', 'inc_angles', 'weights')
            vals = neighborhood_features(x, self.graph(), dirs, mw, scale)
            self._feat = (key, dict(zip(names, vals)))
        return self._feat[1]

    def update_mean(self, invalid=0.0):
        self.mean = self._features(getattr(self, '_scale', None))['mean']

    def update_weights(self, scale=None):
        assert self.mean is not None
        self._scale = scale
        f = self._features(scale)
        self._feat = ((id(self.get_points()), id(self.neighbors), id(f['weights']), scale, id(self.dirs)), f)
        self.weights = f['weights']
        if not scale:
            self.weights._dc_validity = True

    def update_cov(self, correction=1, invalid=0.0):
        self.cov = self._features(getattr(self, '_scale', None))['cov']

    def compute_eig(self):
        f = self._features(getattr(self, '_scale', None))
        return f['eigvals'], f['eigvecs']

    def update_eig(self):
        self.eigvals, self.eigvecs = self.compute_eig()

    def orient_normals(self):
        cos = (self.dirs * self.normals).sum(dim=-1)
        self.normals = -torch.sign(cos)[..., None] * self.normals

    def update_normals(self):
        assert self.eigvecs is not None
        f = self._feat[1] if self._feat is not None and self._feat[1]['eigvecs'] is self.eigvecs else None
        if f is not None:
            self.normals = f['normals']
        else:
            self.normals = self.eigvecs[..., 0]
            self.orient_normals()

    def update_incidence_angles(self, use_normal_sign=False):
        assert self.dirs is not None and self.normals is not None
        f = self._feat[1] if self._feat is not None and self._feat[1]['normals'] is self.normals else None
        if f is not None and not use_normal_sign:
            self.inc_angles = f['inc_angles']
            return
        dot = (self.dirs * self.normals).sum(dim=-1)
        self.inc_angles = torch.arccos(-dot if use_normal_sign else dot.abs().clamp(max=1.0)).unsqueeze(-1)

    def update_features(self, scale=None):
        self._scale = scale
        self.update_mean()
        self.update_weights(scale=scale)
        self.update_cov()
        self.update_eig()
        self.update_normals()
        self.update_incidence_angles()

    def update_all(self, k=None, r=None, scale=None, keep_neighbors=False):
        self.update_points()
        if keep_neighbors:
            self.update_distances()
        else:
            self.update_neighbors(k=k, r=r, _weights=False)
        self.update_features(scale=scale)

    # ---- K13 helpers ------------------------------------------------------------------------------------------
    def _dispersion(self, vec):
        assert self.neighbors is not None
        vec = vec.expand_as(self.dirs).detach().contiguous()
        w = None if self.weights is None else self.weights.detach().reshape(len(self), -1).to(vec.dtype).contiguous()
        return ops.dispersion(vec, self.graph().nbr, w)

    def vp_dispersion(self):
        return self._dispersion(self.vps)

    def dir_dispersion(self):
        return self._dispersion(self.dirs)

    def mean_depth(self):
        assert self.neighbors is not None
        d, w = self.depth.squeeze(dim=1), self.weights.squeeze(dim=2)
        return (w * d[self.neighbors]).sum(dim=-1) / w.sum(dim=-1)

    def mean_vp_dist(self):
        w = self.weights.squeeze(dim=2)
        wsum = w.sum(dim=-1)
        vps = self.vps[self.neighbors]
        centre = (w[..., None] * vps).sum(dim=-2) / wsum[..., None]
        return (w * torch.linalg.norm(vps - centre[:, None], dim=-1)).sum(dim=-1) / wsum

    def vp_dispersion_to_depth2(self):
        return self.vp_dispersion() / self.mean_depth() ** 2

    def vp_dist_to_depth(self, mode='mean'):
        return self.mean_vp_dist() / self.mean_depth()
