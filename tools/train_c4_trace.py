#!/usr/bin/env python3
"""train() on the C4 shape (ring scans, pre-filters, radius neighbourhoods, point-to-plane ICP, model + per-pose corrections), for a
kernel trace or a wall-clock check:  python3 tools/train_c4_trace.py [iterations ...]"""
import contextlib
import io
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_correction_amd.config import Config, Loss, PoseCorrection
from depth_correction_amd.dataset import KittiLikeDataset
from depth_correction_amd.preproc import filtered_cloud
from depth_correction_amd.train import train

cfg = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0,
             max_depth=25.0, vp_dispersion_bounds=[], lr=1e-3, device='cuda:0', loop_batch=64,
             model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]})
seq = [(filtered_cloud(cloud, cfg), pose) for cloud, pose in KittiLikeDataset(n_poses=10)]
for n in [int(v) for v in sys.argv[1:]] or [200]:
    cfg.n_opt_iters, cfg.log_dir = n, tempfile.mkdtemp()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        train(cfg, train_datasets=[seq], val_datasets=[])
    torch.cuda.synchronize()
    print('%d iterations: %.3f s' % (n, time.perf_counter() - t0), file=sys.stderr)
