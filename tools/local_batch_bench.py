#!/usr/bin/env python3
"""Stage times of pipeline.local_features_batch on the ten 200 k-point room scans (device synchronised between stages).
    python3 tools/local_batch_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_correction_amd import ops, pipeline
from depth_correction_amd.dataset import RoomBoxDataset

dev = torch.device('cuda:0')
ds = RoomBoxDataset(n_pts=200_000, n_poses=10, seed_base=1000, dtype=np.float32)
scans = [torch.as_tensor(np.stack([c[f] for f in 'xyz'], axis=1), device=dev) for c, _ in ds]


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


ms, _ = timed(lambda: pipeline.local_features_batch(scans, 10, dtype=torch.float32))
print('local_features_batch        %.2f ms' % ms)
ms, _ = timed(lambda: pipeline.on_streams([lambda s=s: pipeline.local_features(s, k=10, dtype=torch.float32) for s in scans], dev, 4))
print('ten local_features, 4 streams %.2f ms' % ms)
pts = torch.cat(scans).contiguous()
ms, x64 = timed(lambda: pts.double())
print('double()                     %.2f ms' % ms)
ms, _ = timed(lambda: ops.knn(x64, 10, want_dist=False))
print('knn fp64 2M (unshifted)      %.2f ms' % ms)
ms, _ = timed(lambda: ops.knn(pts, 10, want_dist=False))
print('knn fp32 2M (unshifted)      %.2f ms' % ms)
off = torch.zeros_like(x64)
off[:, 2] = torch.repeat_interleave(torch.arange(10, device=dev), 200_000).double() * 13.0
xs = x64 + off
ms, (_, nbr) = timed(lambda: ops.knn(xs, 10, want_dist=False))
print('knn fp64 2M (stacked in z)   %.2f ms' % ms)
depth = pts.norm(dim=-1, keepdim=True)
dirs = (pts / depth).contiguous()
ms, _ = timed(lambda: ops.features_fwd(pts, nbr, dirs=dirs, want=('eigvals', 'normals', 'inc_angles')))
print('features 2M                  %.2f ms' % ms)
