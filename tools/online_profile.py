#!/usr/bin/env python3
"""Where the host time of one online correction (online.correct_cloud on a resident 200 k-point scan) goes: cProfile over 50 calls.
    python3 tools/online_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_correction_amd.config import Config
from depth_correction_amd.dataset import RoomBoxDataset
from depth_correction_amd.model import ScaledPolynomial
from depth_correction_amd.online import correct_cloud
from depth_correction_amd.scan_io import cloud_on_device

dev = torch.device('cuda:0')
ds = RoomBoxDataset(n_pts=200_000, n_poses=1, seed_base=1000, dtype=np.float32)
raw = torch.as_tensor(np.stack([ds[0][0][f] for f in 'xyz'], axis=1), device=dev)
cfg = Config(nn_k=10, nn_r=None, device='cuda:0', float_type='float32', shadow_neighborhood_angle=0.017453,
             shadow_angle_bounds=[float(np.radians(5.0)), float('inf')], log_filters=False)
model = ScaledPolynomial(w=[1e-3, 2e-3], exponent=[2.0, 4.0], device=dev)
for _ in range(5):
    correct_cloud(raw, model, cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    out = correct_cloud(raw, model, cfg)
    torch.cuda.synchronize()
print('wall per call: %.3f ms' % ((time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    out = correct_cloud(raw, model, cfg)
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)

# ---- the host's time line of one call: when the mid-way synchronisation (the number of rows the shadow filter kept) returns, when
# the last launch is queued, when the device is done
from depth_correction_amd import ops as _ops
marks = []
_orig = _ops.scan_prefilter
def _marked(*a, **k):
    marks.append(('before scan_prefilter', time.perf_counter()))
    r = _orig(*a, **k)
    marks.append(('scan_prefilter returned (sync)', time.perf_counter()))
    return r
_ops.scan_prefilter = _marked
rows = []
for _ in range(30):
    del marks[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = correct_cloud(raw, model, cfg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append([(m[1] - t0) * 1e6 for m in marks] + [(t1 - t0) * 1e6, (t2 - t0) * 1e6])
med = np.median(np.array(rows), axis=0)
print('host time line (us, medians of 30): before scan_prefilter %.0f | its synchronisation returned %.0f | all launches queued %.0f | device done %.0f' % tuple(med))
