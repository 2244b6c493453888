#!/usr/bin/env python3
"""profiles/<run>/<workload>/{kernel_stats.csv, pmc_summary.json} (tools/profile_paths.sh) -> profiles/<run>/summary.json and
rows of profiles/traffic.json for the kernels bench.py's timed loop does not launch: duration (kernel trace), HBM bytes per
launch ((2 FETCH_SIZE + WRITE_SIZE) * 1024), VALU instructions per wavefront, per workload.

    python3 tools/make_paths_table.py profiles/r03_paths [commit]"""
import csv
import json
import os
import sys

WANT = {   # workload -> (kernel name fragments, units per launch as text, unit count)
    'c4': (['p2plane_seq_fused_kernel', 'p2plane_seq_kernel', 'p2plane_reduce_all_kernel', 'pose_train_finish_kernel', 'pose_correct_kernel'], '9 scan pairs of ~15 k correspondences', 131980),
    'pose': (['consistency_step_pose_kernel', 'reduce_eval_kernel', 'consistency_bwd_runs_kernel<float, dc::q32, false, true>', 'consistency_fwd_fixed_kernel', 'points_fwd_kernel<float, dc::q32, 4>'], 'N = 2 000 000 points', 2000000),
    'knn2m': (['knn_group_kernel', 'knn_query_kernel<10>', 'knn_tail_kernel'], 'N = 2 000 000 queries', 2000000),
    'knn200k': (['knn_group_kernel', 'knn_query_kernel<10>', 'knn_tail_kernel'], 'N = 200 000 queries', 200000),
    'radius': (['consistency_step_ragged_q32_kernel', 'consistency_step_basis_slots_kernel', 'radius_kernel', 'radius_sort_rows_kernel', 'radius_group_kernel'], '282 303 points, 52.1 M (point, neighbour) pairs (grid 0.2 m, r = 0.4 m)', 52089021),
    'radius25': (['consistency_step_ragged_q32_kernel', 'consistency_step_basis_slots_kernel', 'radius_kernel', 'radius_sort_rows_kernel', 'radius_group_kernel'], '282 303 points, 20.3 M (point, neighbour) pairs (grid 0.2 m, r = 0.25 m)', 20290615),
    'c1': (['features_fwd_tile_kernel<float, 3, 10>', 'features_fwd_kernel<float, 3>'], 'N = 200 000 points', 200000),
    'online': (['shadow_group_kernel', 'knn_group_kernel', 'knn_tail_kernel', 'features_fwd_tile_kernel', 'features_fwd_kernel', 'correct_depth_kernel', 'mask_bounds_multi_kernel', 'cell_keys_kernel', 'sorted_cells_kernel', 'compact_place_kernel', 'scan_direct_kernel',
               'to_points_kernel'], 'one 200 000-point scan', 200000),
}


def main(run_dir, commit=''):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for w, (frags, what, units) in WANT.items():
        d = os.path.join(run_dir, w)
        if not os.path.isdir(d):
            continue
        with open(os.path.join(d, 'pmc_summary.json')) as f:
            pmc = json.load(f)
        with open(os.path.join(d, 'kernel_stats.csv'), newline='') as f:
            stats = list(csv.DictReader(f))
        for frag in frags:
            rows = [r for r in stats if frag in r['Name']]
            pm = [(k, v) for k, v in pmc.items() if frag in k]
            if not rows:
                continue
            r = rows[0]
            e = {'workload': w, 'units': what, 'launches': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3}
            if pm:
                k, v = pm[0]
                c = v['counters']
                if 'hbm_bytes' in v:
                    e['hbm_bytes'] = v['hbm_bytes']
                    e['hbm_GBps'] = v['hbm_bytes'] / (e['avg_us'] * 1e-6) / 1e9
                    e['hbm_frac_of_8TBps'] = e['hbm_GBps'] / 8000.0
                if 'valu_insts_per_wave' in v:
                    e['valu_insts_per_wave'] = v['valu_insts_per_wave']
                    waves = c.get('SQ_WAVES', 0.0)
                    e['waves'] = waves
                    # wave-instruction slots of the chip in the kernel's duration: 1024 SIMDs x 2.4 GHz / 4
                    e['valu_frac_of_issue_peak'] = v['valu_insts_per_wave'] * waves / (e['avg_us'] * 1e-6 * 1024 * 2.4e9 / 4)
                for key in ('sq_wait_any_frac_of_wave_cycles', 'sq_wait_inst_any_frac_of_wave_cycles', 'sq_active_inst_valu_frac_of_wave_cycles'):
                    if key in v:
                        e[key] = v[key]
                if c.get('SQ_LDS_IDX_ACTIVE') and c.get('SQ_BUSY_CYCLES'):
                    e['lds_bank_conflict_share'] = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']
            e['units_per_s'] = units / (e['avg_us'] * 1e-6)
            out['%s @ %s' % (frag, w)] = e
    with open(os.path.join(run_dir, 'summary.json'), 'w') as f:
        json.dump({'source': '%s%s' % (os.path.relpath(run_dir, root), (' @ ' + commit) if commit else ''), 'kernels': out}, f, indent=1, sort_keys=True)
    tpath = os.path.join(root, 'profiles', 'traffic.json')
    table = json.load(open(tpath)) if os.path.exists(tpath) else {}
    table['_paths'] = {'source': os.path.relpath(run_dir, root) + ((' @ ' + commit) if commit else ''), 'kernels': out}
    with open(tpath, 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
    for k, e in sorted(out.items()):
        print('%-70s %9.1f us  hbm %6.0f GB/s (%.2f)  valu/wave %7.0f (%.2f of issue)' % (
            k[:70], e['avg_us'], e.get('hbm_GBps', float('nan')), e.get('hbm_frac_of_8TBps', float('nan')),
            e.get('valu_insts_per_wave', float('nan')), e.get('valu_frac_of_issue_peak', float('nan'))))


if __name__ == '__main__':
    main(sys.argv[1], *(sys.argv[2:3]))
