#!/usr/bin/env python3
"""Workloads for profiling the kernels bench.py's timed loop does not launch (rocprofv3 runs this script once per workload so
that a kernel name stands for one problem size):

    python3 tools/profile_paths.py --what knn2m | knn200k | c1 | pose | c4 | online | radius | radius25 [--reps N]

knn2m / knn200k: dc_knn_build (knn_query_kernel<10>) on the 2 M-point global cloud / one 200 k-point scan;  c1: features_fwd_kernel
<float, 3> on a 200 k-point scan (BASELINE config 1);  pose: the C2 sequence with pose gradients (general path: points_fwd, fixed-K
forward, pose-mode backward);  c4: the C4-shaped train() loop (p2plane_pair_kernel, pose_correct_kernel);  online: correct_cloud."""
import argparse
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

os.environ.setdefault('DC_ENABLE_ABLATIONS', '1')          # this tool flips the library's A-B switches (dc_set_option)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--what', required=True)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--step-var', type=int, default=None, help='dc_set_option(6, v) before the workload (A-B runs)')
    args = ap.parse_args()
    from depth_correction_amd import ops
    from depth_correction_amd.dataset import RoomBoxDataset, KittiLikeDataset
    dev = torch.device('cuda:0')
    if args.step_var is not None:
        from depth_correction_amd import _native as nv
        nv.check(nv.lib().dc_set_option(6, args.step_var), 'dc_set_option')
    if args.what in ('knn2m', 'knn200k', 'c1', 'pose', 'online'):
        ds = RoomBoxDataset(n_pts=200_000, n_poses=10, seed_base=1000, dtype=np.float32)
        scans = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
        poses = np.stack([p for _, p in ds])
    if args.what == 'knn2m':
        xyz = np.concatenate([s.astype(np.float64) + p[:3, 3] for s, p in zip(scans, poses)]).astype(np.float32)
        x = torch.as_tensor(xyz, device=dev)
        for _ in range(args.reps):
            ops.knn(x, 10, want_dist=False)
    elif args.what == 'knn200k':
        x = torch.as_tensor(scans[0], device=dev)
        for _ in range(args.reps):
            ops.knn(x, 10, want_dist=False)
    elif args.what == 'c1':
        x = torch.as_tensor(scans[0], device=dev)
        _, idx = ops.knn(x, 10)
        dirs = x / x.norm(dim=-1, keepdim=True)
        for _ in range(args.reps):
            ops.features_fwd(x, idx, dirs=dirs)
    elif args.what == 'pose':
        from depth_correction_amd.pipeline import build_sequence
        plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32, device=dev)
        w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
        e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
        out = torch.zeros((2 + 4 + 12 * plan.n_scans,), dtype=torch.float64, device=dev)
        P = plan.poses12(info['poses'])
        for _ in range(args.reps):
            plan.eval_native(w, e, P, out, want_grad=True, want_pose=True)
    elif args.what == 'c4':
        from depth_correction_amd.config import Config, Loss, PoseCorrection
        from depth_correction_amd.preproc import filtered_cloud
        from depth_correction_amd.train import train
        cfg = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0,
                     max_depth=25.0, vp_dispersion_bounds=[], n_opt_iters=args.reps, lr=1e-3, device='cuda:0', log_dir=tempfile.mkdtemp(),
                     loop_graph=False, model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]})
        seq = [(filtered_cloud(cloud, cfg), pose) for cloud, pose in KittiLikeDataset(n_poses=10)]
        with contextlib.redirect_stdout(io.StringIO()):
            train(cfg, train_datasets=[seq], val_datasets=[])
    elif args.what == 'online':
        from depth_correction_amd.config import Config
        from depth_correction_amd.model import ScaledPolynomial
        from depth_correction_amd.online import correct_cloud
        from depth_correction_amd.scan_io import cloud_on_device
        cfg = Config(nn_k=10, nn_r=None, device='cuda:0', float_type='float32', shadow_neighborhood_angle=0.017453,
                     shadow_angle_bounds=[float(np.radians(5.0)), float('inf')], log_filters=False)
        model = ScaledPolynomial(w=[1e-3, 2e-3], exponent=[2.0, 4.0], device=dev)
        raw = torch.as_tensor(scans[0], device=dev)
        import time
        lat = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            correct_cloud(raw, model, cfg)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        print('online latency ms: median %.4f  min %.4f  (of %d, the first %.3f)' % (np.median(lat[1:]), min(lat), len(lat), lat[0]))
    elif args.what in ('radius', 'radius25'):
        # ball neighbourhoods (the reference's default): ten room scans, voxel grid 0.2 m, r = 0.4 m / 0.25 m
        from depth_correction_amd.filters import filter_grid
        from depth_correction_amd.pipeline import build_sequence
        from depth_correction_amd.plan import SequenceTrainer
        ds = RoomBoxDataset(n_pts=200_000, n_poses=10, seed_base=1000, dtype=np.float32)
        rng = np.random.default_rng(135)
        kept = [filter_grid(np.stack([c[f] for f in 'xyz'], axis=1), 0.2, keep='random', rng=rng) for c, _ in ds]
        poses = np.stack([p for _, p in ds])
        plan, info = build_sequence(kept, poses, k=None, r=0.4 if args.what == 'radius' else 0.25, dtype=torch.float32, device=dev)
        tr = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3, chained=True)
        for _ in range(args.reps):
            tr.step()
        tr.flush()
    else:
        raise SystemExit('unknown workload ' + args.what)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
