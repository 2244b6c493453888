#!/usr/bin/env python3
"""k-NN build times (dc_knn_build incl. grid set-up) for the shell budget values given: one 200k-point scan and the 2 M-point
global cloud, k = 10; and online.correct_cloud on the scan.   python3 tools/knn_bench.py --budget -1 1 2 3"""
import argparse, json, os, sys, time
os.environ.setdefault('DC_ENABLE_ABLATIONS', '1')          # this tool flips the library's A-B switches (dc_knn_set_shell_budget)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--budget', type=int, nargs='+', default=[1000])
    ap.add_argument('--k', type=int, default=10)
    args = ap.parse_args()
    from depth_correction_amd import ops, _native as nv
    from depth_correction_amd.dataset import RoomBoxDataset
    dev = torch.device('cuda:0')
    ds = RoomBoxDataset(n_pts=200_000, n_poses=10, seed_base=1000, dtype=np.float32)
    scans = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    x1 = torch.as_tensor(scans[0], device=dev)
    x2 = torch.as_tensor(np.concatenate([s.astype(np.float64) + p[:3, 3] for s, p in zip(scans, poses)]).astype(np.float32), device=dev)
    ref = {}
    for b in args.budget:
        nv.check(nv.lib().dc_knn_set_shell_budget(b), 'budget')
        res = {'budget': b}
        for name, x in (('n200k', x1), ('n2m', x2)):
            for _ in range(3):
                d, i = ops.knn(x, args.k)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.knn(x, args.k, want_dist=False); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res[name + '_ms'] = round(float(np.median(ts)), 3)
            if name not in ref:
                ref[name] = (d.clone(), i.clone())
            res[name + '_same'] = bool(torch.equal(ref[name][0], d) and torch.equal(ref[name][1], i))
        print(json.dumps(res))
    nv.check(nv.lib().dc_knn_set_shell_budget(1000), 'budget')


if __name__ == '__main__':
    main()
