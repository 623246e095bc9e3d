#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<dir>) into the small files committed under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r1c r1_final gather
writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and updates profiles/pmc_summary.json[kind].
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd.build import kernel_source_hash  # noqa: E402


def main():
    src, tag, kind = sys.argv[1], sys.argv[2], sys.argv[3]
    n_envs = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    out_dir = os.path.join(ROOT, 'profiles')
    os.makedirs(out_dir, exist_ok=True)
    WINDOW = 50  # launches of the timed region of tools/profile_round.sh (the settle steps before it are left out of every average)
    stats = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))
    kernel_us = kernel_us_all = None
    trace = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_trace.csv'))
    if trace:  # average duration of the last WINDOW k_step dispatches (the stats file averages over the settle steps too)
        d = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(trace[0])) if 'k_step' in r['Kernel_Name']]
        d = sorted(d)[-WINDOW:]
        if d:
            kernel_us = sum(e - s for s, e in d) / len(d) / 1e3
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        for r in rows:
            if 'k_step' in r['Name']:
                kernel_us_all = float(r['AverageNs']) / 1e3
        if kernel_us is None:
            kernel_us = kernel_us_all
        with open(os.path.join(out_dir, f'{tag}_kernel_stats.csv'), 'w') as f:
            w = csv.writer(f)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
            for r in rows:
                name = r['Name'] if len(r['Name']) <= 120 else r['Name'][:117] + '...'
                w.writerow([name, r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs'], r['StdDev']])
    counters = {}
    for d in sorted(glob.glob(os.path.join(src, 'pmc_*'))):
        for f in glob.glob(os.path.join(d, '*', '*_counter_collection.csv')):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'k_step' in r['Kernel_Name']:
                    acc[r['Counter_Name']].append((int(r.get('Dispatch_Id', len(acc[r['Counter_Name']]))), float(r['Counter_Value'])))
            for k, v in acc.items():
                v = [x for _, x in sorted(v)][-WINDOW:]  # dispatch order: the settled window
                counters[k] = {'launches': len(v), 'mean_per_launch': sum(v) / len(v), 'mean_per_wave': sum(v) / len(v) / n_envs}
    meta = {'source': src, 'kernel': f'k_step<{kind}>', 'envs_per_launch': n_envs, 'source_sha256': kernel_source_hash(),
            'command': 'rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --steps 50 --warmup 300 --no-cpu-baseline; '
                       'every figure = mean over the last 50 launches (the settled window)',
            'kernel_us_window': kernel_us, 'kernel_us_all_launches': kernel_us_all,
            'units': 'FETCH_SIZE / WRITE_SIZE in KiB per launch as rocprofv3 reports them (RAW).  On gfx950 FETCH_SIZE reports half of the bytes '
                     '(MI355X_MICROARCH.md, HBM) -- calibrated on these kernels\' own access pattern, 4-B-per-lane record loads and small rows '
                     'with known byte counts (tools/traffic_calib.sh, profiles/r5_traffic_calibration.txt): FETCH_SIZE x 0.50, WRITE_SIZE x 1.000 -- '
                     'so pmc_summary.json and the bench line use 2 x FETCH_SIZE + WRITE_SIZE; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles',
            'counters': counters}
    json.dump(meta, open(os.path.join(out_dir, f'{tag}_pmc.json'), 'w'), indent=1)
    if 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
        path = os.path.join(out_dir, 'pmc_summary.json')
        summ = json.load(open(path)) if os.path.exists(path) else {}
        summ[kind] = {'tag': tag, 'envs_per_launch': n_envs, 'source_sha256': kernel_source_hash(),
                      # the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes; confirmed for this access pattern: tools/traffic_calib.sh)
                      'fetch_bytes_per_env': 2.0 * counters['FETCH_SIZE']['mean_per_launch'] * 1024 / n_envs,
                      'fetch_bytes_per_env_raw_counter': counters['FETCH_SIZE']['mean_per_launch'] * 1024 / n_envs,
                      'write_bytes_per_env': counters['WRITE_SIZE']['mean_per_launch'] * 1024 / n_envs}
        if 'SQ_INSTS_VALU' in counters:  # one wave per env: per-wave counters are per-env counters
            summ[kind]['valu_insts_per_env'] = counters['SQ_INSTS_VALU']['mean_per_wave']
        other = ['SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR']
        if 'SQ_INSTS_VALU' in counters and all(k in counters for k in other):  # everything a wave issues (s_nop / s_waitcnt are not counted by these)
            summ[kind]['insts_per_env'] = counters['SQ_INSTS_VALU']['mean_per_wave'] + sum(counters[k]['mean_per_wave'] for k in other)
            summ[kind]['salu_insts_per_env'] = counters['SQ_INSTS_SALU']['mean_per_wave']
        if 'SQ_WAVE_CYCLES' in counters:  # quad-cycles -> cycles (MI355X_MICROARCH.md, cycle constants)
            summ[kind]['wave_cycles_per_env'] = counters['SQ_WAVE_CYCLES']['mean_per_wave'] * 4
        if kernel_us is not None:
            summ[kind]['kernel_us_profiled'] = kernel_us
        json.dump(summ, open(path, 'w'), indent=1)
    print(json.dumps({k: round(v['mean_per_wave'], 1) for k, v in counters.items()}))


if __name__ == '__main__':
    main()
