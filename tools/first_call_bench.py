#!/usr/bin/env python3
"""What the first call into each part of libdc_hip.so costs in a fresh process (code-object load + first launch), one small call per
translation unit, then the same calls again (steady state).    python3 tools/first_call_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_correction_amd import ops

dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
torch.cuda.synchronize()
x = torch.as_tensor(np.random.default_rng(0).normal(size=(4096, 3)).astype(np.float32), device=dev)
dirs = (x / x.norm(dim=-1, keepdim=True)).contiguous()
depth = x.norm(dim=-1, keepdim=True).contiguous()
torch.cuda.synchronize()


def timed(name, fn, out):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    out[name] = round((time.perf_counter() - t0) * 1e3, 2)
    return r


for rnd in range(2):
    t = {}
    ps = ops.PointSet(None, dirs, depth)
    xs = timed('points_fwd (dc_consistency)', lambda: ops.points_fwd(ps), t)
    nbr = timed('knn (dc_knn)', lambda: ops.knn(xs, 10, want_dist=False)[1], t)
    f = timed('features_fwd (dc_features)', lambda: ops.features_fwd(xs, nbr, dirs=dirs), t)
    m = torch.ones((len(xs),), dtype=torch.bool, device=dev)
    timed('mask_bounds (dc_filters)', lambda: ops.mask_bounds(m, f['eigvals'], 0, f['eigvals'], 1, 0.0, 0.25), t)
    timed('spatial_order', lambda: ops.spatial_order(xs), t)
    timed('block_table (dc_blocktab)', lambda: ops.block_table(nbr=nbr), t)
    timed('voxel_filter (dc_filters)', lambda: ops.voxel_filter(xs, 0.5, None, False), t)
    print('round', rnd, t, 'sum %.1f ms' % sum(t.values()))
