#!/usr/bin/env python3
"""profiles/<run>/pmc_summary.json -> profiles/traffic.json: per kernel instantiation and problem size the measured HBM
bytes per launch ((2 * FETCH_SIZE + WRITE_SIZE) * 1024, MI355X_MICROARCH.md) and VALU instructions per point, which
bench.py looks up by the exact instantiation name the library reports for its launches (dc_profiler_kernel).

    python3 tools/make_traffic_table.py profiles/r02_final 2000000 [commit]
"""
import json
import os
import sys


def main(run_dir, n, commit=''):
    with open(os.path.join(run_dir, 'pmc_summary.json')) as f:
        summary = json.load(f)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_path = os.path.join(root, 'profiles', 'traffic.json')
    table = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            table = {k: v for k, v in json.load(f).items() if isinstance(v, dict)}
    for kernel, d in summary.items():
        name = kernel.replace('dc::', '').strip()
        if not any(s in name for s in ('consistency_fwd', 'consistency_bwd', 'consistency_step', 'points_fwd_kernel<float, q32')):
            continue
        if 'hbm_bytes' not in d or 'valu_insts_per_wave' not in d:
            continue
        table['%s/N%d' % (name, int(n))] = {
            'hbm_bytes': d['hbm_bytes'], 'valu_insts_per_point': d['valu_insts_per_wave'],
            'source': '%s/pmc_summary.json%s' % (os.path.relpath(run_dir, root), (' @ ' + commit) if commit else ''),
            'launches': d.get('launches')}
    table['_comment'] = ('HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes (the factor 2 '
                         'is the gfx950 correction for wide coalesced reads, an upper bound for gather parts); VALU instructions per '
                         'point = SQ_INSTS_VALU / SQ_WAVES.  Keys: kernel instantiation as dc_profiler_kernel reports it / N.')
    with open(out_path, 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print('\n'.join(k for k in sorted(table) if not k.startswith('_')))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4]))
