"""Scan and pose file formats on either side of the path, host readers that hand the rest of the package
plain arrays (the cloud then goes to the device once, in ``local_feature_cloud``):

* KITTI-360 Velodyne ``.bin``: float32 ``[N,4]`` (x, y, z, intensity) with the ego-vehicle box removed
  (datasets/kitti360.py:96-109);
* ASL-laser point CSV: one header line, columns 1:4 = x, y, z (datasets/asl_laser.py:33-37), and ``.npz`` variants
  (``arr_0`` / ``cloud`` keys, asl_laser.py:40-45, fee_corridor.py:35-38);
* pose CSV ``poseId, timestamp, T00 .. T33`` (asl_laser.py:48-66), read and written.

``ScanFolderDataset`` yields ``(cloud, pose)`` like the reference's dataset classes, so it plugs into ``train()``.
"""
from __future__ import annotations

import os

import numpy as np
from numpy.lib.recfunctions import unstructured_to_structured

__all__ = ['read_kitti_bin', 'read_points_csv', 'read_points_npz', 'read_poses_csv', 'write_poses_csv',
           'ScanFolderDataset']


def read_kitti_bin(path, filter_ego_pts_depth=1.0):
    """Structured cloud (x, y, z, i) of one KITTI-360 scan; points with |x| <= d and |y| <= d (the car) are dropped."""
    cloud = np.fromfile(path, dtype=np.float32).reshape((-1, 4))
    if filter_ego_pts_depth is not None:
        d = filter_ego_pts_depth
        keep = (cloud[:, 0] < -d) | (cloud[:, 0] > d) | (cloud[:, 1] < -d) | (cloud[:, 1] > d)
        cloud = cloud[keep]
    return unstructured_to_structured(cloud, names=['x', 'y', 'z', 'i'])


def read_points_csv(path):
    """[N,3] float64 from an ASL-laser style CSV (header line, then id, x, y, z, ...)."""
    return np.genfromtxt(path, delimiter=',', skip_header=1)[:, 1:4]


def read_points_npz(path):
    data = np.load(path)
    return data['cloud'] if 'cloud' in data.files else data['arr_0']


def read_poses_csv(path):
    """(ids, [4x4 poses]) from ``poseId, timestamp, T00, ..., T33`` rows."""
    rows = np.atleast_2d(np.genfromtxt(path, delimiter=',', skip_header=1))
    ids = rows[:, 0].astype(int).tolist()
    return ids, list(rows[:, 2:].reshape((-1, 4, 4)))


def write_poses_csv(ids, poses, path, ts=None):
    ts = ids if ts is None else ts
    with open(path, 'w') as f:
        f.write('poseId, timestamp, ' + ', '.join('T%d%d' % (r, c) for r in range(4) for c in range(4)) + '\n')
        for i, t, pose in zip(ids, ts, poses):
            f.write('%s, %.9f, %s\n' % (i, t, ', '.join('%.9f' % v for v in np.asarray(pose).flatten())))


class ScanFolderDataset(object):
    """A sequence stored as ``<dir>/<id>.{bin,csv,npz}`` scans plus one pose CSV."""

    def __init__(self, cloud_dir, poses_csv, pattern='%010d.bin', name=None):
        self.cloud_dir, self.pattern, self.name = cloud_dir, pattern, name or os.path.basename(cloud_dir.rstrip('/'))
        ids, poses = read_poses_csv(poses_csv)
        have = [os.path.exists(os.path.join(cloud_dir, pattern % i)) for i in ids]
        self.ids = [i for i, h in zip(ids, have) if h]
        self.poses = [p for p, h in zip(poses, have) if h]

    def __len__(self):
        return len(self.ids)

    def local_cloud(self, i):
        path = os.path.join(self.cloud_dir, self.pattern % i)
        if path.endswith('.bin'):
            return read_kitti_bin(path)
        pts = read_points_csv(path) if path.endswith('.csv') else read_points_npz(path)
        return pts if pts.dtype.names else unstructured_to_structured(np.ascontiguousarray(pts[:, :3]), names=['x', 'y', 'z'])

    def __getitem__(self, k):
        return self.local_cloud(self.ids[k]), self.poses[k]

    def __iter__(self):
        for k in range(len(self)):
            yield self[k]

    def __str__(self):
        return self.name
