#!/usr/bin/env python3
"""Set-up phase of the C2 sequence (train.py:94-215: upload, ten local feature clouds, global cloud, global k-NN, masks, plan) three
times in one process, with the device synchronised between stages: the first call (code-object loads, allocator warm-up) and the
steady state.    python3 tools/setup_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t_imp = time.perf_counter()
from depth_correction_amd.dataset import RoomBoxDataset
from depth_correction_amd.pipeline import build_sequence
from depth_correction_amd import _native
ds = RoomBoxDataset(n_pts=200000, n_poses=10, seed_base=1000, dtype=np.float32)
scans_xyz = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
poses = np.stack([p for _, p in ds])
t0 = time.perf_counter()
torch.zeros(1, device='cuda:0')
torch.cuda.synchronize()
t1 = time.perf_counter()
_native.lib()
t2 = time.perf_counter()
print('torch device init %.1f ms, dlopen(libdc_hip.so) %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan, info = build_sequence(scans_xyz, poses, k=10, dtype=torch.float32, device='cuda:0', stage_times=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(rep, round((t1 - t0) * 1e3, 1), {k: round(v, 1) for k, v in info['setup_ms'].items()})
    del plan, info
