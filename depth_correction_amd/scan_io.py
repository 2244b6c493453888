"""Scan and pose file formats on either side of the path.

Direct-to-device loaders (``load_*_device``, ``DeviceScanDataset``): the file's bytes go through a pinned host buffer to
the GPU once, and ONE kernel pass (dc_cloud_from_points) applies the ego-vehicle crop, the depth pre-filter and
``DepthCloud.from_points`` -- the reference does those three on the host, array by array.  The result is a DepthCloud on
the device, ready for ``local_feature_cloud``.

Host readers with the reference's return types (structured / plain numpy arrays), for tools that want arrays:

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
           'ScanFolderDataset', 'upload', 'cloud_on_device', 'load_kitti_bin_device', 'load_points_csv_device',
           'load_points_npz_device', 'DeviceScanDataset']


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


# ---- direct-to-device loaders ---------------------------------------------------------------------------------------
def upload(array, device='cuda:0'):
    """Host array -> device tensor through a pinned staging buffer (asynchronous copy on torch's current stream)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(array))
    if torch.device(device).type != 'cuda':
        return t
    return t.pin_memory().to(device, non_blocking=True)


def cloud_on_device(raw, vps=None, dtype=None, device='cuda:0', ego_box=None, min_depth=None, max_depth=None):
    """Uploaded raw rows -> DepthCloud on the device in one kernel pass: ego-box crop (kitti360.py:101-105), depth
    pre-filter (filters.py:116-141), from_points (depth_cloud.py:592-638).  ``raw`` [N, >=3] float32 / float64 (host or
    device), ``vps`` [N,3] or None; ``dtype`` numpy / torch float type of the cloud (default: the raw dtype)."""
    import torch
    from . import ops
    from .depth_cloud import DepthCloud
    if isinstance(dtype, type) or isinstance(dtype, np.dtype):
        dtype = getattr(torch, np.dtype(dtype).name)
    raw_t = raw if isinstance(raw, torch.Tensor) else upload(raw, device)
    if vps is not None:
        vps = (vps if isinstance(vps, torch.Tensor) else upload(vps, device)).to(raw_t.dtype).contiguous()
    v, dirs, depth, _ = ops.cloud_from_points(raw_t.contiguous(), vps, dtype=dtype, ego_box=ego_box, min_depth=min_depth,
                                              max_depth=max_depth, want_zero_vps=True)
    return DepthCloud(v, dirs, depth)


def load_kitti_bin_device(path, device='cuda:0', dtype=None, filter_ego_pts_depth=1.0, min_depth=None, max_depth=None):
    """KITTI-360 Velodyne scan (float32 [N,4]) straight to a DepthCloud on the device."""
    raw = np.fromfile(path, dtype=np.float32).reshape((-1, 4))
    return cloud_on_device(raw, dtype=dtype, device=device, ego_box=filter_ego_pts_depth, min_depth=min_depth, max_depth=max_depth)


def load_points_csv_device(path, device='cuda:0', dtype=None, min_depth=None, max_depth=None):
    """ASL-laser point CSV (header line; id, x, y, z, ...) to a DepthCloud on the device."""
    return cloud_on_device(read_points_csv(path), dtype=dtype, device=device, min_depth=min_depth, max_depth=max_depth)


def load_points_npz_device(path, device='cuda:0', dtype=None, min_depth=None, max_depth=None):
    """``.npz`` scans: plain [N,3+] arrays (asl_laser.py:40-45) or structured arrays with x, y, z [, vp_x, vp_y, vp_z]
    (fee_corridor.py:35-38) to a DepthCloud on the device."""
    from numpy.lib.recfunctions import structured_to_unstructured
    arr = read_points_npz(path)
    vps = None
    if arr.dtype.names:
        if 'vp_x' in arr.dtype.names:
            vps = structured_to_unstructured(arr[['vp_x', 'vp_y', 'vp_z']])
        arr = structured_to_unstructured(arr[['x', 'y', 'z']])
    if arr.dtype not in (np.float32, np.float64):
        arr = arr.astype(np.float64)
    return cloud_on_device(arr, vps=None if vps is None else vps.astype(arr.dtype), dtype=dtype, device=device,
                           min_depth=min_depth, max_depth=max_depth)


class DeviceScanDataset(ScanFolderDataset):
    """ScanFolderDataset whose items are DepthClouds already on the device (``local_feature_cloud`` accepts them as they
    are), with the configuration's depth pre-filter fused into the load."""

    def __init__(self, cloud_dir, poses_csv, pattern='%010d.bin', name=None, device='cuda:0', dtype=None, min_depth=None,
                 max_depth=None):
        super().__init__(cloud_dir, poses_csv, pattern=pattern, name=name)
        self.device, self.dtype, self.min_depth, self.max_depth = device, dtype, min_depth, max_depth

    def local_cloud(self, i):
        path = os.path.join(self.cloud_dir, self.pattern % i)
        kw = dict(device=self.device, dtype=self.dtype, min_depth=self.min_depth, max_depth=self.max_depth)
        if path.endswith('.bin'):
            return load_kitti_bin_device(path, **kw)
        return load_points_csv_device(path, **kw) if path.endswith('.csv') else load_points_npz_device(path, **kw)
