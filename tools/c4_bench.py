#!/usr/bin/env python3
"""BASELINE config 4 shape on one GPU through the reference-API train(): KITTI-360-like ring scans (64 x 2048 rays),
depth 5-25 m + 0.2 m voxel pre-filters, radius neighbourhoods (0.4 m), point-to-plane ICP loss, model AND per-pose
corrections optimised with torch.optim.Adam.  Prints the wall-clock per training iteration (median over the loop).

    python3 tools/c4_bench.py [--scans 10] [--iters 30] [--point-to-point]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scans', type=int, default=10)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--point-to-point', action='store_true')
    args = ap.parse_args()
    from depth_correction_amd.config import Config, Loss, PoseCorrection
    from depth_correction_amd.dataset import KittiLikeDataset
    from depth_correction_amd.preproc import filtered_cloud
    from depth_correction_amd.train import TrainCallbacks, train
    cfg = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0,
                 max_depth=25.0, vp_dispersion_bounds=[], n_opt_iters=args.iters, lr=1e-3, device='cuda:0',
                 log_dir=tempfile.mkdtemp(), model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]})
    cfg.loss_kwargs['icp_point_to_plane'] = not args.point_to_point
    ds = KittiLikeDataset(n_poses=args.scans)
    t0 = time.perf_counter()
    seq = [(filtered_cloud(cloud, cfg), pose) for cloud, pose in ds]
    t_filter = time.perf_counter() - t0
    stamps = []

    class CB(TrainCallbacks):
        def iteration_started(self, it):
            torch.cuda.synchronize()
            stamps.append(time.perf_counter())

    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        train(cfg, callbacks=CB(cfg), train_datasets=[seq], val_datasets=[])
    torch.cuda.synchronize()
    stamps.append(time.perf_counter())
    it_ms = np.diff(stamps) * 1e3
    print(json.dumps({'workload': 'C4 shape: %d scans x 64 x 2048 rays, %d points after the pre-filters, icp_loss (%s), pose + model'
                                  % (args.scans, sum(len(c) for c, _ in seq), 'point to point' if args.point_to_point else 'point to plane'),
                      'train_iteration_ms_median': float(np.median(it_ms[2:])), 'first_iterations_ms': it_ms[:2].round(2).tolist(),
                      'prefilter_s': t_filter, 'setup_s': stamps[0] - t0 - t_filter}))


if __name__ == '__main__':
    main()
