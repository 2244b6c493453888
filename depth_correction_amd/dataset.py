"""Synthetic scan generators (host side, numpy only).

Only the generators the hot path's tests and ``bench.py`` need are provided; the reference's
real-data readers and mesh renderers (dataset.py:361-1331, datasets/*) are out of scope.

* ``PlaneDataset``   -- restatement of the reference generator of BASELINE config 0
                        (dataset.py:240-358): two 10x10 m half planes, per-scan random subsample.
* ``RoomBoxDataset`` -- "ASL-laser-shaped" scans of BASELINE configs 1-3.
* ``KittiLikeDataset`` -- "KITTI-360-shaped" ring scans of BASELINE config 4.

Every dataset yields ``(cloud, pose)`` like the reference's datasets do: ``cloud`` is a structured
array with fields x, y, z (+ normal_x.. for PlaneDataset) in the sensor frame, ``pose`` a 4x4 float64.
"""
from __future__ import annotations

import numpy as np
from numpy.lib.recfunctions import unstructured_to_structured, merge_arrays

__all__ = ['PlaneDataset', 'RoomBoxDataset', 'KittiLikeDataset', 'create_dataset', 'add_depth_noise', 'Forwarding',
           'TransformingDataset', 'FilteredDataset', 'NoisyPoseDataset', 'NoisyDepthDataset', 'noisy_dataset', 'euler_matrix']


def _structured(xyz, normals=None):
    cloud = unstructured_to_structured(np.ascontiguousarray(xyz), names=['x', 'y', 'z'])
    if normals is not None:
        nrm = unstructured_to_structured(np.ascontiguousarray(normals),
                                         names=['normal_x', 'normal_y', 'normal_z'])
        cloud = merge_arrays([cloud, nrm], flatten=True)
    return cloud


class _Seq:
    def __len__(self):
        return len(self.ids)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __str__(self):
        return self.name


class PlaneDataset(_Seq):
    """Ground-plane measurements from several view points (reference dataset.py:320-358, 240-317)."""

    def __init__(self, name='plane', n_pts=10_000, n_poses=2, height=2.0,
                 size=([-10.0, 10.0], [-10.0, 10.0], [-10.0, 10.0])):
        self.name, self.n_pts, self.n_poses, self.height, self.size = name, n_pts, n_poses, height, size
        self.ids = range(n_poses)
        # dataset.py:335-344: legacy global seeding, two half planes, z = 0, normals +z.
        np.random.seed(135)
        pts = np.zeros((n_pts, 3), dtype=np.float64)
        pts[:, :2] = np.concatenate([np.random.uniform(0, size[0][1], size=(n_pts // 2, 2)),
                                     np.random.uniform(0, size[1][1], size=(n_pts // 2, 2))
                                     + np.array([size[0][0], 0])])
        self.pts = pts
        self.normals = np.zeros_like(pts)
        self.normals[:, 2] = 1.0

    def cloud_pose(self, i):
        rng = np.random.default_rng(i)                       # dataset.py:288-294
        pose = np.eye(4)
        for p in range(2):
            pose[p, 3] = rng.uniform(low=0.6 * self.size[p][0], high=0.6 * self.size[p][1])
        pose[2, 3] = self.height
        return pose

    def local_cloud(self, i):
        rng = np.random.default_rng(i)                       # dataset.py:269-286
        sel = rng.choice(range(self.n_pts), size=self.n_pts // self.n_poses, replace=False)
        pose = self.cloud_pose(i)
        R, t = pose[:3, :3], pose[:3, 3]
        xyz = (self.pts[sel] - t) @ R                        # inverse rigid transform, row vectors
        nrm = self.normals[sel] @ R
        return _structured(xyz, nrm)

    def __getitem__(self, i):
        i = self.ids[i]
        return self.local_cloud(i), self.cloud_pose(i)


class RoomBoxDataset(_Seq):
    """Scans of the inside of a box room from view points along a line.

    Room half extents (10, 7, 2) m; scan s is taken at vp_s = (-4.5 + s, 0.3 sin s, 0); directions are
    uniform on the sphere (``default_rng(seed_base + s)``), range = first wall hit times
    ``1 + range_noise * N(0,1)``; the sensor frame is the world frame shifted by vp_s (R = I).
    """
    half = np.array([10.0, 7.0, 2.0])

    def __init__(self, name='room', n_pts=200_000, n_poses=10, seed_base=1000, range_noise=1e-3, dtype=np.float64):
        if n_poses > 14:
            raise ValueError('RoomBoxDataset: view point s sits at x = -4.5 + s, inside the 10 m half extent only for s <= 14')
        self.name, self.n_pts, self.n_poses = name, n_pts, n_poses
        self.seed_base, self.range_noise, self.dtype = seed_base, range_noise, dtype
        self.ids = range(n_poses)

    def cloud_pose(self, s):
        pose = np.eye(4)
        pose[:3, 3] = (-4.5 + 1.0 * s, 0.3 * np.sin(s), 0.0)
        return pose

    def local_cloud(self, s):
        rng = np.random.default_rng(self.seed_base + s)
        d = rng.normal(size=(self.n_pts, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        vp = self.cloud_pose(s)[:3, 3]
        with np.errstate(divide='ignore'):
            t = np.where(d > 0, (self.half - vp) / d, (-self.half - vp) / d)
        rng_ = np.nanmin(t, axis=1)
        rng_ = rng_ * (1.0 + self.range_noise * rng.normal(size=self.n_pts))
        return _structured((rng_[:, None] * d).astype(self.dtype))

    def __getitem__(self, i):
        s = self.ids[i]
        return self.local_cloud(s), self.cloud_pose(s)


class KittiLikeDataset(_Seq):
    """64 rings x ``n_azimuth`` rays over a ground plane and two walls."""

    def __init__(self, name='kitti_like', n_poses=10, n_rings=64, n_azimuth=2048, max_range=80.0,
                 height=1.73, wall_y=8.0, range_noise=2e-3, seed_base=2000):
        self.name, self.n_poses, self.n_rings, self.n_azimuth = name, n_poses, n_rings, n_azimuth
        self.max_range, self.height, self.wall_y = max_range, height, wall_y
        self.range_noise, self.seed_base = range_noise, seed_base
        self.ids = range(n_poses)

    def cloud_pose(self, s):
        yaw = 0.01 * s
        pose = np.eye(4)
        pose[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
        pose[:3, 3] = (1.0 * s, 0.0, self.height)
        return pose

    def local_cloud(self, s):
        rng = np.random.default_rng(self.seed_base + s)
        el = np.deg2rad(np.linspace(-24.8, 2.0, self.n_rings))
        az = np.linspace(-np.pi, np.pi, self.n_azimuth, endpoint=False) + 1e-3 * rng.normal()
        el, az = np.meshgrid(el, az, indexing='ij')
        d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1).reshape(-1, 3)
        pose = self.cloud_pose(s)
        dw = d @ pose[:3, :3].T
        o = pose[:3, 3]
        with np.errstate(divide='ignore', invalid='ignore'):
            t_ground = np.where(dw[:, 2] < 0, (0.0 - o[2]) / dw[:, 2], np.inf)
            t_wall = np.where(dw[:, 1] > 0, (self.wall_y - o[1]) / dw[:, 1],
                              np.where(dw[:, 1] < 0, (-self.wall_y - o[1]) / dw[:, 1], np.inf))
        t = np.minimum(t_ground, t_wall)
        keep = np.isfinite(t) & (t < self.max_range) & (t > 0.5)
        t = t[keep] * (1.0 + self.range_noise * rng.normal(size=int(keep.sum())))
        return _structured(t[:, None] * d[keep])

    def __getitem__(self, i):
        s = self.ids[i]
        return self.local_cloud(s), self.cloud_pose(s)


def add_depth_noise(cloud, sigma, rng):
    """Gaussian noise along the viewing ray (sensor at the origin of the sensor frame)."""
    xyz = np.stack([cloud[f] for f in 'xyz'], axis=1).astype(np.float64)
    depth = np.linalg.norm(xyz, axis=1, keepdims=True)
    xyz = xyz / depth * (depth + sigma * rng.normal(size=depth.shape))
    out = cloud.copy()
    for i, f in enumerate('xyz'):
        out[f] = xyz[:, i]
    return out


def create_dataset(name, cfg=None, **kwargs):
    """Subset of the reference's ``create_dataset`` (dataset.py:953-962): synthetic names only."""
    if name.startswith('plane'):
        return PlaneDataset(**kwargs)
    if name.startswith('room'):
        return RoomBoxDataset(**kwargs)
    if name.startswith('kitti_like'):
        return KittiLikeDataset(**kwargs)
    raise ValueError('Unsupported dataset: %s (real-data readers are out of scope).' % name)


# ---- dataset wrappers of the caller scripts (dataset.py:718-873, :933-950) ------------------------------------------------
class Forwarding:
    """A stand-in for ``target``: attribute look-ups that fail here, indexing, iteration, ``len`` and ``str`` all reach the
    wrapped object (contract of the reference's wrapper base, dataset.py:718-735)."""

    def __init__(self, target):
        self.target = target

    def __getattr__(self, name):                 # only consulted when normal look-up fails
        if name == 'target':                     # (not yet set: unpickling / copy) -- no recursion
            raise AttributeError(name)
        return getattr(self.target, name)

    __getitem__ = lambda self, key: self.target[key]
    __iter__ = lambda self: iter(self.target)
    __len__ = lambda self: len(self.target)
    __str__ = lambda self: str(self.target)


class TransformingDataset(Forwarding):
    """``target``'s (cloud, pose) items with both members passed through the hooks ``transform_cloud`` / ``transform_pose``
    (identity here; the hooks get the item's index as keyword ``item``).  Integer indexing, iteration and the by-id accessors
    ``local_cloud`` / ``cloud_pose`` all go through the same hooks (contract: dataset.py:738-762)."""

    def transform_cloud(self, cloud, **kwargs):
        return cloud

    def transform_pose(self, pose, **kwargs):
        return pose

    def _hooked(self, index, pair):
        cloud, pose = pair
        return self.transform_cloud(cloud, item=index), self.transform_pose(pose, item=index)

    def __getitem__(self, item):
        assert isinstance(item, int), item
        return self._hooked(item, self.target[item])

    def __iter__(self):
        return (self._hooked(index, pair) for index, pair in enumerate(self.target))

    def local_cloud(self, id):
        return self.transform_cloud(self.target.local_cloud(id))

    def cloud_pose(self, id):
        return self.transform_pose(self.target.cloud_pose(id))


class FilteredDataset(TransformingDataset):
    """Clouds through preproc.filtered_cloud (depth + voxel-grid pre-filters of ``cfg``; dataset.py:765-773)."""

    def __init__(self, dataset, cfg):
        super().__init__(dataset)
        self.cfg = cfg

    def transform_cloud(self, cloud, **kwargs):
        from .preproc import filtered_cloud
        return filtered_cloud(cloud, self.cfg)


def euler_matrix(ai, aj, ak):
    """tf.transformations.euler_matrix(ai, aj, ak) with its default axes 'sxyz' (rotations about the static x, y, z axes in
    that order: R = Rz(ak) Ry(aj) Rx(ai)), 4x4.  tf is a ROS package and absent here: restated from its published
    definition, not pinned against it."""
    ci, cj, ck = np.cos(ai), np.cos(aj), np.cos(ak)
    si, sj, sk = np.sin(ai), np.sin(aj), np.sin(ak)
    M = np.eye(4)
    M[0, 0], M[0, 1], M[0, 2] = cj * ck, sj * si * ck - ci * sk, sj * ci * ck + si * sk
    M[1, 0], M[1, 1], M[1, 2] = cj * sk, sj * si * sk + ci * ck, sj * ci * sk - si * ck
    M[2, 0], M[2, 1], M[2, 2] = -sj, cj * si, cj * ci
    return M


class NoisyPoseDataset(TransformingDataset):
    """Poses right-multiplied by a random rigid transform: Euler angles and translation ~ noise * N(0, 1), seeded by the pose
    itself (mode 'pose': every pose its own perturbation, the first one left alone unless ``first_noisy``) or by the
    configuration's seed (mode 'common': one perturbation for all) -- dataset.py:776-812."""

    class Mode(object):
        pose = 'pose'
        common = 'common'
        values = ('pose', 'common')

    def __init__(self, dataset, noise=0.0, mode=None, first_noisy=False):
        assert isinstance(noise, float) or len(noise) == 6
        assert mode is not None and mode in NoisyPoseDataset.Mode.values
        super().__init__(dataset)
        self.noise, self.mode, self.first_noisy = np.asarray(noise), mode, first_noisy

    def random_transform(self, seed):
        vec = self.noise * np.random.default_rng(seed).normal(size=(6,))
        T = euler_matrix(*vec[:3])
        T[:3, 3] = vec[3:]
        return T

    def transform_pose(self, pose, item=None):
        from .config import Config
        from .utils import hashable
        if self.mode == NoisyPoseDataset.Mode.pose:
            if not self.first_noisy and item == 0:
                print('No noise for first pose')
                return pose
            seed = abs(hash(hashable(pose)))
        else:
            seed = Config().random_seed
        if (self.noise != 0.0).any():
            pose = np.matmul(pose, self.random_transform(seed))
        return pose


class NoisyDepthDataset(TransformingDataset):
    """Points moved along their rays by noise * N(0, 1), seeded by the depths themselves (dataset.py:815-846)."""

    def __init__(self, dataset, noise=None):
        super().__init__(dataset)
        self.noise = noise

    def transform_cloud(self, cloud, **kwargs):
        from numpy.lib.recfunctions import structured_to_unstructured
        from .utils import hashable
        if self.noise:
            pts = structured_to_unstructured(cloud[['x', 'y', 'z']])
            dirs = pts - (structured_to_unstructured(cloud[['vp_x', 'vp_y', 'vp_z']]) if 'vp_x' in cloud.dtype.names else 0.0)
            depth = np.linalg.norm(dirs, axis=1)
            valid = depth > 0.0
            depth = depth[valid]
            dirs = dirs[valid] / depth[:, None]
            rng = np.random.default_rng(abs(hash(hashable(depth))))
            pts[valid] += dirs * self.noise * rng.normal(size=depth.shape)[:, None]
            cloud[['x', 'y', 'z']] = unstructured_to_structured(pts, names=['x', 'y', 'z'])
        return cloud


def noisy_dataset(ds, cfg):
    """Depth noise and pose noise of the configuration on top of ``ds`` (dataset.py:933-950; the depth-bias wrapper needs a
    mesh-free inverse model and is applied by the caller where wanted)."""
    if getattr(cfg, 'depth_noise', 0.0):
        print('Adding depth noise %.3g.' % cfg.depth_noise)
        ds = NoisyDepthDataset(ds, noise=cfg.depth_noise)
    if getattr(cfg, 'pose_noise_mode', None) is not None and np.any(np.asarray(getattr(cfg, 'pose_noise', 0.0)) != 0.0):
        print('Adding pose noise %s, %s.' % (cfg.pose_noise, cfg.pose_noise_mode))
        ds = NoisyPoseDataset(ds, noise=cfg.pose_noise, mode=cfg.pose_noise_mode)
    return ds
