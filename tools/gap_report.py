#!/usr/bin/env python3
"""Gaps between consecutive kernels of the timed loop, from a rocprofv3 --kernel-trace CSV:
    python3 tools/gap_report.py <dir with *_kernel_trace.csv>
prints, for the dc_sequence_step loop, the median duration of each hot kernel and the median idle time before it."""
import csv, glob, os, sys
import numpy as np

path = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
names = ('consistency_step', 'consistency_fwd_basis', 'consistency_bwd_basis', 'reduce_eval', 'points_fwd_kernel',
         'consistency_fwd_fixed', 'consistency_bwd_runs')
stat = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows[:-1], rows[1:]):
    key = next((k for k in names if k in n1), None)
    prev = next((k for k in names if k in n0), None)
    if key is None or prev is None:
        continue
    stat.setdefault(key, []).append((e1 - s1, s1 - e0))
for k, v in stat.items():
    a = np.array(v[len(v) // 4:])          # skip the warm-up quarter
    print('%-26s n=%4d  duration %.2f us   idle before %.2f us' % (k, len(a), np.median(a[:, 0]) / 1e3, np.median(a[:, 1]) / 1e3))
