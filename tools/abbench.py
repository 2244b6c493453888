#!/usr/bin/env python3
"""A-B timing of forms of the one-pass step kernel in ONE process (interleaved rounds, cdna_hip_programming.md rule 24):
builds the C2 sequence once, then for every `--var` value (dc_set_option(6, v)) times chained steps and compares the
evaluation's sums with the first variant's.

    python3 tools/abbench.py --var 0 1 3 7 [--rounds 5] [--steps 200] [--scans 10] [--points 200000] [--k 10]"""
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
    ap.add_argument('--var', type=int, nargs='+', default=[0])
    ap.add_argument('--opt', type=int, default=6, help='which dc_set_option switch the variants are values of')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--scans', type=int, default=10)
    ap.add_argument('--points', type=int, default=200_000)
    ap.add_argument('--k', type=int, default=10)
    ap.add_argument('--dtype', default='float32')
    ap.add_argument('--no-chain', action='store_true')
    ap.add_argument('--no-wave-pack', action='store_true')
    args = ap.parse_args()
    from depth_correction_amd import _native as nv
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import SequenceTrainer, KernelTimer
    dev = torch.device('cuda', 0)
    dtype = getattr(torch, args.dtype)
    ds = RoomBoxDataset(n_pts=args.points, n_poses=args.scans, seed_base=1000,
                        dtype=np.float32 if dtype == torch.float32 else np.float64)
    scans_xyz = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    plan, info = build_sequence(scans_xyz, poses, k=args.k, dtype=dtype, device=dev, wave_pack=False if args.no_wave_pack else None)
    print(json.dumps({'skipped_wavefronts': plan.skipped_wavefronts, 'max_rows': int(plan.fwd_table.max_rows), 'rows_active': plan.fwd_rows_active, 'loss_table_max_rows': None if plan.fwd_table_loss is None else int(plan.fwd_table_loss.max_rows)}))
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device=dev)
    e = torch.tensor([2.0, 4.0], dtype=torch.float64, device=dev)
    P12 = plan.poses12(info['poses'])
    setv = lambda v: nv.check(nv.lib().dc_set_option(args.opt, int(v)), 'dc_set_option')
    # ---- the evaluation's sums per variant
    sums = {}
    for v in args.var:
        setv(v)
        out = torch.zeros((2 + 4 + 12 * plan.n_scans,), dtype=torch.float64, device=dev)
        plan.eval_native(w, e, P12, out)
        sums[v] = out[:4].cpu().numpy().copy()
    base = sums[args.var[0]]
    # ---- warm the clocks, then interleaved rounds
    setv(args.var[0])
    tr = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3, chained=not args.no_chain)
    for _ in range(600):
        tr.step()
    tr.flush()
    torch.cuda.synchronize()
    times = {v: [] for v in args.var}
    ktimes = {v: [] for v in args.var}
    for r in range(args.rounds):
        for v in args.var:
            setv(v)
            tr = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3, chained=not args.no_chain)
            for _ in range(20):
                tr.step()
            tr.flush()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with KernelTimer(every=8) as timer:
                a.record()
                for _ in range(args.steps):
                    tr.step()
                tr.flush()
                b.record()
                torch.cuda.synchronize()
                k = timer.read()
            times[v].append(a.elapsed_time(b) / args.steps * 1e3)
            ktimes[v].append(k['consistency_fwd'][0] * 1e3 if 'consistency_fwd' in k else float('nan'))
    setv(0)
    for v in args.var:
        rel = np.abs(sums[v] - base) / np.maximum(np.abs(base), 1e-300)
        print(json.dumps({'var': v, 'step_us_median': round(float(np.median(times[v])), 2), 'step_us_min': round(float(np.min(times[v])), 2),
                          'kernel_us_median': round(float(np.median(ktimes[v])), 2), 'kernel_us_min': round(float(np.min(ktimes[v])), 2),
                          'rel_diff_vs_first': [float('%.3g' % x) for x in rel], 'sums': [float(x) for x in sums[v]], 'n': plan.n}))


if __name__ == '__main__':
    main()
