#!/usr/bin/env python3
"""Per-kernel timings of the fused path on the C2 workload (or a smaller one): builds one sequence plan, then times
dc_points_fwd / dc_consistency_fwd / dc_consistency_bwd in isolation and the whole dc_sequence_step with HIP events.

    python3 tools/kbench.py [--scans 10] [--points 200000] [--k 10] [--reps 200] [--opt IDX=VAL ...] [--pose-grad]

`--opt` sets library ablation switches (dc_set_option) before timing, so variants can be compared by running the
script several times in the same gpurun call."""
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
    ap.add_argument('--scans', type=int, default=10)
    ap.add_argument('--points', type=int, default=200_000)
    ap.add_argument('--k', type=int, default=10)
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--dtype', default='float32')
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--pose-grad', action='store_true')
    ap.add_argument('--bwd-layout', default='runs')
    ap.add_argument('--degree-sort', action='store_true')
    ap.add_argument('--tag', default='')
    ap.add_argument('--no-basis', action='store_true')
    args = ap.parse_args()
    from depth_correction_amd import _native as nv, ops
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import SequenceTrainer
    dev = torch.device('cuda', 0)
    dtype = getattr(torch, args.dtype)
    ds = RoomBoxDataset(n_pts=args.points, n_poses=args.scans, seed_base=1000,
                        dtype=np.float32 if dtype == torch.float32 else np.float64)
    scans_xyz = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    plan, info = build_sequence(scans_xyz, poses, k=args.k, dtype=dtype, device=dev, bwd_layout=args.bwd_layout,
                                degree_sort=args.degree_sort, basis=not args.no_basis)
    for o in args.opt:
        i, v = o.split('=')
        nv.check(nv.lib().dc_set_option(int(i), int(v)), 'dc_set_option')
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
    P = plan.poses12(info['poses'])

    # kernel-only durations: the library stamps every 2nd launch of each hot kernel with the dispatch's own start / end
    from depth_correction_amd.plan import KernelTimer
    want_pose = args.pose_grad
    out = torch.zeros((2 + 2 * 2 + 12 * plan.n_scans,), dtype=torch.float64, device=dev)
    if want_pose:
        step = lambda: plan.eval_native(w, e, P, out, want_grad=True, want_pose=True)
    else:
        tr = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3)
        step = tr.step
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with KernelTimer(every=2) as timer:
        a.record()
        for _ in range(args.reps):
            step()
        b.record()
        torch.cuda.synchronize()
        res = {'tag': args.tag}
        for name, (ms, cnt) in timer.read().items():
            res[name] = ms * 1e3
        res['kernels'] = {k: v.split('<')[0] for k, v in timer.kernels().items()}
    res['step'] = a.elapsed_time(b) / args.reps * 1e3
    res = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}
    res['n'] = plan.n
    print(json.dumps(res))


if __name__ == '__main__':
    main()
