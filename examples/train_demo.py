#!/usr/bin/env python3
"""Drop-in usage of the reference's training entry points on an MI355X (what scripts/train_demo and
scripts/model_poses_learning_icp drive in the reference), on the synthetic datasets of this repo.

    python examples/train_demo.py                 # room-box scans, min_eigval_loss, model only      (config 2 shape)
    python examples/train_demo.py --icp           # KITTI-like rings, point-to-plane ICP, model + poses (config 4 shape)

Unmodified reference callers can instead do
    import depth_correction_amd; depth_correction_amd.install_as('depth_correction')
and keep their `from depth_correction.xxx import yyy` lines.
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from depth_correction_amd.config import Config, Loss, PoseCorrection            # noqa: E402
from depth_correction_amd.dataset import KittiLikeDataset, RoomBoxDataset       # noqa: E402
from depth_correction_amd.preproc import filtered_cloud                          # noqa: E402
from depth_correction_amd.train import train                                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--icp', action='store_true')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--points', type=int, default=20000)
    ap.add_argument('--scans', type=int, default=6)
    args = ap.parse_args()
    cfg = Config(n_opt_iters=args.iters, lr=1e-3, log_dir=tempfile.mkdtemp(prefix='dc_amd_'), float_type='float32',
                 model_kwargs={'w': [0.0, 0.0], 'exponent': [2.0, 4.0]})
    if args.icp:
        cfg.from_dict(dict(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2,
                           min_depth=5.0, max_depth=25.0, vp_dispersion_bounds=[]))
        ds = KittiLikeDataset(n_poses=args.scans, n_rings=32, n_azimuth=1024)
        seq = [(filtered_cloud(cloud, cfg), pose) for cloud, pose in ds]
    else:
        cfg.from_dict(dict(nn_k=10, nn_r=None, grid_res=0.0, min_depth=0.0, max_depth=float('inf'), vp_dispersion_bounds=[]))
        ds = RoomBoxDataset(n_pts=args.points, n_poses=args.scans, dtype=np.float32)
        seq = list(ds)
    best = train(cfg, train_datasets=[seq], val_datasets=[seq])
    print('best model state:', best.model_state_dict if best is not None else None)


if __name__ == '__main__':
    main()
