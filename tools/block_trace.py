#!/usr/bin/env python3
"""Schedule of ONE launch of a one-pass step kernel, from the diagnostic build of the library (csrc/dc_consistency.hip with
-DDC_BLOCK_TRACE: every block records {XCC | HW_ID, start, end} on the 100 MHz constant clock):

    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDC_BLOCK_TRACE -Iinclude -c depth_correction_amd/csrc/dc_consistency.hip -o build/dc_consistency_trace.o
    hipcc -shared -fPIC --offload-arch=gfx950 -o build/libdc_hip_trace.so build/dc_consistency_trace.o <the other objects of depth_correction_amd/lib/obj>
    python3 tools/block_trace.py [--radius 0.2:0.25]

Prints per launch: its length, the distribution of block lengths, blocks and busy time per CU, when the CUs ran dry."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

os.environ.setdefault('DC_ENABLE_ABLATIONS', '1')          # this tool flips the library's A-B switches (dc_set_option)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--radius', default='', help='grid:radius (ball neighbourhoods on voxel-filtered scans); default: C2, k = 10')
    ap.add_argument('--heavy-first', type=int, default=1)
    ap.add_argument('--points', type=int, default=200_000)
    ap.add_argument('--forwards', type=int, default=0, help='1: dc_set_option(8, 1), every launch of the chain walks the blocks forwards')
    ap.add_argument('--lib', default=os.path.join(ROOT, 'build', 'libdc_hip_trace.so'))
    args = ap.parse_args()
    from depth_correction_amd import _native
    _native.lib_path = lambda: args.lib
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import SequenceTrainer
    dev = torch.device('cuda:0')
    ds = RoomBoxDataset(n_pts=args.points, n_poses=10, seed_base=1000, dtype=np.float32)
    scans = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    if args.radius:
        from depth_correction_amd.filters import filter_grid
        grid, r = (float(v) for v in args.radius.split(':'))
        rng = np.random.default_rng(135)
        kept = [filter_grid(s, grid, keep='random', rng=rng) for s in scans]
        plan, info = build_sequence(kept, poses, k=None, r=r, dtype=torch.float32, device=dev, heavy_first=bool(args.heavy_first))
    else:
        plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32, device=dev)
    _native.lib().dc_set_option(8, int(args.forwards))
    tr = SequenceTrainer([plan], [1e-3, 2e-3], [2.0, 4.0], [info['poses']], lr=1e-3, chained=True)
    for _ in range(300):
        tr.step()
    tr.flush()
    torch.cuda.synchronize()
    cap = 1 << 16
    buf = torch.zeros((cap, 4), dtype=torch.int64, device=dev)
    fn = _native.lib().dc_debug_block_trace
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    out = []
    for rep in range(3):
        for _ in range(20):
            tr.step()
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
        tr.step()                                   # ONE traced launch, in the middle of a chain
        torch.cuda.synchronize()
        assert fn(ctypes.c_void_p(0)) == 0
        tr.step()
        tr.flush()
        torch.cuda.synchronize()
        t = buf.cpu().numpy()
        t = t[t[:, 3] == 1]
        hw, xcc = t[:, 0] & 0xffffffff, (t[:, 0] >> 32) & 0xf
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
        t0, t1 = t[:, 1].astype(np.float64) * 0.01, t[:, 2].astype(np.float64) * 0.01          # microseconds
        begin, end = t0.min(), t1.max()
        dur = t1 - t0
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        ids = np.unique(cuid)
        busy = np.array([dur[cuid == c].sum() for c in ids])
        cnt = np.array([(cuid == c).sum() for c in ids])
        last = np.array([t1[cuid == c].max() for c in ids]) - begin
        first = np.array([t0[cuid == c].min() for c in ids]) - begin
        q = lambda a: [round(float(v), 2) for v in np.percentile(a, [0, 10, 50, 90, 100])]
        # where the dispatcher puts block g of the grid: the same CU as block g - 256 / g - 8?  (rows of the trace are blockIdx.x)
        full = buf.cpu().numpy()
        g = np.nonzero(full[:, 3] == 1)[0]
        cu_of = np.full(int(g.max()) + 1, -1, dtype=np.int64)
        cu_of[g] = cuid
        xcc_of = np.full(int(g.max()) + 1, -1, dtype=np.int64)
        xcc_of[g] = xcc
        ok = lambda a, b: float(np.mean(a[(a >= 0) & (b >= 0)] == b[(a >= 0) & (b >= 0)]))
        place = {'same_cu_as_block_minus_256': round(ok(cu_of[256:], cu_of[:-256]), 3), 'same_cu_as_block_minus_512': round(ok(cu_of[512:], cu_of[:-512]), 3),
                 'xcc_is_block_mod_8': round(float(np.mean(xcc_of[g] == (g - int(g.min())) % 8)), 3),
                 'first_traced_block': int(g.min()), 'cu_of_first_40': [int(v) for v in cu_of[g.min():g.min() + 40]]}
        first_round = np.sort(t0)[min(len(t0) - 1, 1535)]
        out.append({'block_us_started_in_first_round_p50': round(float(np.median(dur[t0 <= first_round])), 2),
                    'block_us_started_later_p50': round(float(np.median(dur[t0 > first_round])), 2) if (t0 > first_round).any() else None,
                    'blocks_traced': int(len(t)), 'launch_us': round(float(end - begin), 2), 'cus_seen': int(len(ids)),
                    'block_us_p0_10_50_90_100': q(dur), 'blocks_per_cu_p0_10_50_90_100': q(cnt),
                    'cu_first_start_us': q(first), 'cu_last_end_us': q(last),
                    'cu_busy_block_us_sum': q(busy),
                    'blocks_started_after_half': int((t0 - begin > (end - begin) / 2).sum()),
                    'mean_concurrency': round(float(dur.sum() / (end - begin)), 1),
                    'placement': place, 'share_of_slot_time_idle_at_the_end': round(float(((end - begin) - last).sum() / (len(ids) * (end - begin))), 3)})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
