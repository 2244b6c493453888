#!/usr/bin/env python3
"""Ball neighbourhoods (the reference's default: nn_type = ball, nn_r, config.py:187-189) on the ten room scans after a voxel
filter: radius-search time, neighbourhood statistics, and microseconds per chained optimisation step (SequenceTrainer).

    python3 tools/radius_bench.py [--cases 0.2:0.4,0.2:0.25,0.1:0.25] [--steps 200]        (grid:radius in metres)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='0.2:0.4,0.2:0.25,0.1:0.25')
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--points', type=int, default=200_000)
    ap.add_argument('--degree-sort', type=int, default=0)
    ap.add_argument('--heavy-first', type=int, default=1, help='blocks with the longest rows first in the grid (SequencePlan.heavy_first)')
    args = ap.parse_args()
    from depth_correction_amd import ops
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd.filters import filter_grid
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import KernelTimer, SequenceTrainer
    dev = torch.device('cuda:0')
    ds = RoomBoxDataset(n_pts=args.points, n_poses=10, seed_base=1000, dtype=np.float32)
    scans = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    out = {}
    for case in args.cases.split(','):
        grid, r = (float(v) for v in case.split(':'))
        rng = np.random.default_rng(135)
        kept = [filter_grid(s, grid, keep='random', rng=rng) for s in scans]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan, info = build_sequence(kept, poses, k=None, r=r, dtype=torch.float32, device=dev, degree_sort=bool(args.degree_sort),
                                    heavy_first=bool(args.heavy_first))
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        nbr = info['neighbors']
        deg = (nbr >= 0).sum(1)
        x0 = info['points0']
        # lanes of a wavefront run to its longest row: useful share of the slots the wavefronts walk
        dl = deg[plan.order] if plan.order is not None else deg
        dpad = torch.zeros(((dl.numel() + 63) // 64) * 64, dtype=dl.dtype, device=dl.device)
        dpad[:dl.numel()] = dl
        wave_eff = float(dl.sum()) / float((dpad.reshape(-1, 64).max(1).values * 64).sum())
        for _ in range(2):
            ops.radius_neighbors(x0, r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.radius_neighbors(x0, r)
        torch.cuda.synchronize()
        search_ms = (time.perf_counter() - t0) * 1e3
        tr = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3, chained=True)
        for _ in range(50):
            tr.step()
        tr.flush()
        torch.cuda.synchronize()
        with KernelTimer(every=4) as kt:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tr.step()
            sums = tr.flush()
            torch.cuda.synchronize()
            step_us = (time.perf_counter() - t0) / args.steps * 1e6
            ks, names = kt.read(), kt.kernels()
        out[case] = {'points': plan.n, 'kmax': int(nbr.shape[1]), 'mean_degree': float(deg.double().mean()),
                     'pairs': int(deg.sum()), 'step_us': step_us, 'chained': bool(tr.chained),
                     'kernel_us': {k: round(v[0] * 1e3, 1) for k, v in ks.items()}, 'kernel': names.get('consistency_fwd'),
                     'wavefront_slot_efficiency': wave_eff, 'ps_per_pair': step_us * 1e6 / float(deg.sum()), 'radius_search_ms': search_ms, 'setup_ms': setup_ms,
                     'fused_table': plan.fwd_table is not None, 'max_rows_per_block': None if plan.fwd_table is None else plan.fwd_table.max_rows, 'loss': float(sums[0] / sums[1])}
        del plan, info, tr
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
