#!/usr/bin/env python3
"""A-B timing of dc_features_fwd (BASELINE config 1: one 200 k-point scan, k = 10, all DepthCloud features written): the tiled
kernel against the general run-time-k kernel in one process, plus their largest output differences.

    python3 tools/features_bench.py [--n 200000] [--k 10] [--reps 300] [--dtype float32|float64|both]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

os.environ.setdefault('DC_ENABLE_ABLATIONS', '1')          # this tool flips the library's A-B switches (dc_features_set_tiled)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=200_000)
    ap.add_argument('--k', type=int, default=10)
    ap.add_argument('--reps', type=int, default=300)
    ap.add_argument('--dtype', default='both')
    args = ap.parse_args()
    from depth_correction_amd import ops
    from depth_correction_amd._native import lib
    from depth_correction_amd.dataset import RoomBoxDataset
    dev = torch.device('cuda:0')
    ds = RoomBoxDataset(n_pts=args.n, n_poses=1, seed_base=1000, dtype=np.float32)
    xyz = np.stack([ds[0][0][f] for f in 'xyz'], axis=1)
    out = {}
    for dt in ([torch.float32, torch.float64] if args.dtype == 'both' else [getattr(torch, args.dtype)]):
        x = torch.as_tensor(xyz, device=dev).to(dt).contiguous()
        _, idx = ops.knn(x, args.k)
        dirs = (x / x.norm(dim=-1, keepdim=True)).contiguous()
        res = {}
        for name, on in (('general', 0), ('tiled', 1)):
            lib().dc_features_set_tiled(on)
            f = ops.features_fwd(x, idx, dirs=dirs, want_saved=True)
            res[name] = {k: v.clone() for k, v in f.items() if v is not None}
            us_all = timed(lambda: ops.features_fwd(x, idx, dirs=dirs), args.reps)
            us_set = timed(lambda: ops.features_fwd(x, idx, dirs=dirs, want=('eigvals', 'normals', 'inc_angles')), args.reps)
            out.setdefault(str(dt), {})[name] = {'all_outputs_us': round(us_all, 2), 'setup_outputs_us': round(us_set, 2)}
        lib().dc_features_set_tiled(1)
        # the same scan with its points in Morton order (what a scan in sensor order -- ring by ring -- looks like to the caches:
        # neighbours are near in memory; RoomBoxDataset's rays come in random order, every gather its own cache line)
        order = ops.spatial_order(x).long()
        xs, ds_ = x[order].contiguous(), dirs[order].contiguous()
        _, idx_s = ops.knn(xs, args.k)
        out[str(dt)]['tiled_morton_ordered_scan'] = {
            'all_outputs_us': round(timed(lambda: ops.features_fwd(xs, idx_s, dirs=ds_), args.reps), 2),
            'setup_outputs_us': round(timed(lambda: ops.features_fwd(xs, idx_s, dirs=ds_, want=('eigvals', 'normals', 'inc_angles')), args.reps), 2)}
        diffs = {}
        for k in res['general']:
            a, b = res['general'][k].double(), res['tiled'][k].double()
            if k == 'eigvecs':                                       # sign is arbitrary
                s = torch.sign((a * b).sum(dim=1, keepdim=True))
                b = b * s
            scale = a.abs().max().item() or 1.0
            diffs[k] = float((a - b).abs().max().item() / scale)
        out[str(dt)]['max_abs_diff_over_scale'] = diffs
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
