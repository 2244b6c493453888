#!/usr/bin/env python3
"""Wall-clock per iteration of the reference-API train() itself (train.py:220-322) with its default callbacks, on the C2
workload (10 x 200k-point room scans, K = 10, ScaledPolynomial, min-eigenvalue loss, model weights only) and on the C4 shape
(ring scans, pre-filters, radius neighbourhoods, point-to-plane ICP, model + per-pose corrections).

    python3 tools/train_bench.py [--c2-iters 2000] [--c4-iters 2000] [--loop-batch 64] [--dist]

--dist: a one-rank RCCL group with the collectives forced (DC_FORCE_DIST=1): train() then takes its SHARDED loops -- what one rank
of a multi-GPU run executes per iteration, the all-reduce included (keys get the suffix _dist)."""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed_train(cfg, train_ds, n_short, n_long):
    """ms per iteration from two runs of different length (set-up cancels)."""
    from depth_correction_amd.train import train
    res = {n_short: [], n_long: []}
    for n in (n_short, n_short, n_long, n_short, n_long):          # the first run also pays the one-time costs of the process
        c = cfg.copy()
        c.n_opt_iters, c.log_dir = n, tempfile.mkdtemp()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            train(c, train_datasets=train_ds, val_datasets=[])
        torch.cuda.synchronize()
        res[n].append(time.perf_counter() - t0)
    return (min(res[n_long]) - min(res[n_short][1:])) / (n_long - n_short) * 1e3, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--c2-iters', type=int, default=2000)
    ap.add_argument('--c4-iters', type=int, default=2000)
    ap.add_argument('--loop-batch', type=int, default=64)
    ap.add_argument('--skip-c4', action='store_true')
    ap.add_argument('--skip-batch-1', action='store_true')
    ap.add_argument('--dist', action='store_true')
    args = ap.parse_args()
    sfx = ''
    if args.dist:
        import torch.distributed as dist
        os.environ['DC_FORCE_DIST'] = '1'
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        saved_fd = os.dup(1)
        os.dup2(2, 1)                                                   # (RCCL's banner goes to stderr)
        try:
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', rank=0, world_size=1, device_id=torch.device('cuda:0'))
            dist.all_reduce(torch.zeros((1,), device='cuda:0'))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        sfx = '_dist'
    from depth_correction_amd.config import Config, Loss, PoseCorrection
    from depth_correction_amd.dataset import KittiLikeDataset, RoomBoxDataset
    from depth_correction_amd.preproc import filtered_cloud
    out = {'loop_batch': args.loop_batch}
    # ---- C2 through train()
    cfg = Config(nn_k=10, nn_r=None, min_depth=0.0, max_depth=float('inf'), grid_res=0.0, vp_dispersion_bounds=[], lr=1e-3,
                 float_type='float32', device='cuda:0', loop_batch=args.loop_batch,
                 model_kwargs={'w': [1e-3, 2e-3], 'exponent': [2.0, 4.0]})
    ds = RoomBoxDataset(n_pts=200_000, n_poses=10, seed_base=1000, dtype=np.float32)
    seq = [(c, p) for c, p in ds]
    for batch in (args.loop_batch,) if args.skip_batch_1 else (args.loop_batch, 1):
        cfg.loop_batch = batch
        ms, raw = timed_train(cfg, [seq], 50, 50 + (args.c2_iters if batch > 1 else min(args.c2_iters, 400)))
        out['c2_train_iteration_ms' + sfx + ('' if batch > 1 else '_loop_batch_1')] = ms
    # ---- C2 with per-pose corrections (scripts/model_poses_learning:71): the map-consistency loss, model + poses optimised
    from depth_correction_amd.plan import KernelTimer
    cfgp = cfg.copy()
    cfgp.pose_correction, cfgp.loop_batch = PoseCorrection.pose, args.loop_batch
    with KernelTimer(every=16) as kt:
        ms, raw = timed_train(cfgp, [seq], 50, 50 + args.c2_iters)
        out['c2_pose_train_kernels'] = kt.kernels()
    out['c2_pose_train_iteration_ms' + sfx] = ms
    if not args.skip_c4:
        cfg4 = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0,
                      max_depth=25.0, vp_dispersion_bounds=[], lr=1e-3, device='cuda:0', loop_batch=args.loop_batch,
                      model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]})
        seq4 = [(filtered_cloud(cloud, cfg4), pose) for cloud, pose in KittiLikeDataset(n_poses=10)]
        out['c4_points'] = int(sum(len(c) for c, _ in seq4))
        for batch in (args.loop_batch,) if args.skip_batch_1 else (args.loop_batch, 1):
            cfg4.loop_batch = batch
            ms, raw = timed_train(cfg4, [seq4], 30, 30 + (args.c4_iters if batch > 1 else min(args.c4_iters, 200)))
            out['c4_train_iteration_ms' + sfx + ('' if batch > 1 else '_loop_batch_1')] = ms
    print(json.dumps(out))
    if args.dist:
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
