#!/usr/bin/env python3
"""Pose-mode evaluation (loss, dL/dw and dL/d[R|t] of every scan: what train() runs with pose corrections, train.py:300-312) on the C2
sequence: microseconds per evaluation, with and without the per-block scan grouping of the plan's layout.

    python3 tools/pose_bench.py [--reps 200] [--no-group]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

os.environ.setdefault('DC_ENABLE_ABLATIONS', '1')          # this tool flips the library's A-B switches (dc_set_option)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--points', type=int, default=200_000)
    ap.add_argument('--variants', default='grouped,ungrouped')
    ap.add_argument('--option', default='', help='dc_set_option pairs, e.g. 7=1 (three-kernel path)')
    ap.add_argument('--heavy-first', type=int, default=1, help='ball neighbourhoods: blocks with the longest rows first (SequencePlan.heavy_first)')
    ap.add_argument('--radius', default='', help='grid:radius in metres: ball neighbourhoods on voxel-filtered scans instead of k = 10')
    args = ap.parse_args()
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import KernelTimer
    dev = torch.device('cuda:0')
    for kv in filter(None, args.option.split(',')):
        from depth_correction_amd import _native
        _native.lib().dc_set_option(*[int(v) for v in kv.split('=')])
    ds = RoomBoxDataset(n_pts=args.points, n_poses=10, seed_base=1000, dtype=np.float32)
    scans = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    out = {}
    ref = None
    for name in args.variants.split(','):
        if args.radius:
            from depth_correction_amd.filters import filter_grid
            grid, r = (float(v) for v in args.radius.split(':'))
            rng = np.random.default_rng(135)
            kept = [filter_grid(s_, grid, keep='random', rng=rng) for s_ in scans]
            plan, info = build_sequence(kept, poses, k=None, r=r, dtype=torch.float32, device=dev, scan_group=(name == 'grouped'),
                                        degree_group=False, heavy_first=bool(args.heavy_first))
        else:
            plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32, device=dev, scan_group=(name == 'grouped'))
        w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
        e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
        res = torch.zeros((2 + 4 + 12 * plan.n_scans,), dtype=torch.float64, device=dev)
        P = plan.poses12(info['poses'])
        for _ in range(50):
            plan.eval_native(w, e, P, res, want_grad=True, want_pose=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with KernelTimer(every=1) as kt:
            a.record()
            for _ in range(args.reps):
                plan.eval_native(w, e, P, res, want_grad=True, want_pose=True)
            b.record()
            torch.cuda.synchronize()
            ks = kt.read()
            names = kt.kernels()
        a2, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a2.record()
        for _ in range(args.reps):
            plan.eval_native(w, e, P, res, want_grad=True, want_pose=True)
        b2.record()
        torch.cuda.synchronize()
        out[name] = {'us_per_evaluation': a2.elapsed_time(b2) * 1e3 / args.reps,
                     'us_per_evaluation_with_every_launch_stamped': a.elapsed_time(b) * 1e3 / args.reps,
                     'kernels_us': {k: round(v[0] * 1e3, 2) for k, v in ks.items()}, 'kernels': names}
        r = res.cpu().numpy().copy()
        if ref is None:
            ref = r
        else:
            out[name]['max_rel_diff_vs_first'] = float(np.abs(r - ref).max() / np.abs(ref).max())
        del plan, info
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
