#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: mean counter value per launch for every dc:: kernel.

    python3 tools/pmc_summary.py <raw_dir with pmc*/..._counter_collection.csv> <out.json>

Per kernel: {"counters": {name: mean per launch}, "launches": n, and derived figures following
MI355X_MICROARCH.md: "hbm_bytes" = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE is in KiB and tallies the 128-B
requests of wide coalesced reads at 64 B on gfx950: doubled, an upper bound for gather parts of the pattern),
"valu_insts_per_wave", "valu_active_frac" = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both in quad-cycles), ...}."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(raw, out):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in sorted(glob.glob(os.path.join(raw, 'pmc*', '**', '*counter_collection.csv'), recursive=True)):
        per_dispatch = defaultdict(float)
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                name = row['Kernel_Name']
                if 'dc::' not in name:
                    continue
                # a counter of one dispatch may come as several rows (one per instance / dimension): sum them
                per_dispatch[(name, row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
        for (name, _, counter), v in per_dispatch.items():
            a = acc[name][counter]
            a[0] += v
            a[1] += 1
    res = {}
    for name, counters in acc.items():
        short = name.split('(')[0].replace('void ', '')
        c = {k: v[0] / v[1] for k, v in counters.items()}
        d = {'counters': c, 'launches': max(v[1] for v in counters.values())}
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            d['hbm_bytes'] = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
        if c.get('SQ_WAVES') and 'SQ_INSTS_VALU' in c:
            d['valu_insts_per_wave'] = c['SQ_INSTS_VALU'] / c['SQ_WAVES']
        if c.get('SQ_WAVE_CYCLES'):
            for k in ('SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY'):
                if k in c:
                    d[k.lower() + '_frac_of_wave_cycles'] = c[k] / c['SQ_WAVE_CYCLES']
        res[short] = d
    with open(out, 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for name, d in sorted(res.items()):
        if any(s in name for s in ('consistency', 'points_fwd')):
            print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k != 'counters'})


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
