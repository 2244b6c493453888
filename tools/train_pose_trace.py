#!/usr/bin/env python3
"""train() with per-pose corrections on the C2 workload, for a kernel trace (rocprofv3 --kernel-trace --stats -- python3 tools/train_pose_trace.py)."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_correction_amd.config import Config, PoseCorrection
from depth_correction_amd.dataset import RoomBoxDataset
from depth_correction_amd.train import train

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = Config(nn_k=10, nn_r=None, min_depth=0.0, max_depth=float('inf'), grid_res=0.0, vp_dispersion_bounds=[], lr=1e-3,
             float_type='float32', device='cuda:0', loop_batch=64, model_kwargs={'w': [1e-3, 2e-3], 'exponent': [2.0, 4.0]})
cfg.pose_correction = PoseCorrection.pose
cfg.loop_graph = os.environ.get('DC_LOOP_GRAPH', '1') != '0'
cfg.n_opt_iters, cfg.log_dir = n_it, tempfile.mkdtemp()
seq = [(c, p) for c, p in RoomBoxDataset(n_pts=200_000, n_poses=10, seed_base=1000, dtype=np.float32)]
import time
import torch
for n in ([n_it] if len(sys.argv) < 3 else [int(v) for v in sys.argv[1:]]):
    cfg.n_opt_iters, cfg.log_dir = n, tempfile.mkdtemp()
    buf = io.StringIO()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        train(cfg, train_datasets=[seq], val_datasets=[])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    notes = [ln for ln in buf.getvalue().splitlines() if not ln.startswith('It. ')]
    print('%d iterations: %.3f s' % (n, dt), [ln for ln in notes if 'captur' in ln], file=sys.stderr)
