#!/usr/bin/env python3
"""Post-process gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the tracked evidence
under profiles/: <tag>_kernel_stats.csv, <tag>_pmc_summary.csv, <tag>_pmc_traffic.json, <tag>_bench.json.

    python tools/summarize_profile.py r01e
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def main(tag):
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    # bench JSON line
    bj = os.path.join(src, 'bench.json')
    if os.path.exists(bj):
        line = [ln for ln in open(bj).read().splitlines() if ln.startswith('{')][-1]
        json.dump(json.loads(line), open(os.path.join(dst, f'{tag}_bench.json'), 'w'), indent=1)
    # kernel stats (names truncated so that the csv stays readable)
    for sub, out in (('stats', f'{tag}_kernel_stats.csv'), ('stats_combat', f'{tag}_combat_kernel_stats.csv'),
                     ('stats_actor', f'{tag}_actor_kernel_stats.csv'), ('stats_collect', f'{tag}_collect_loop_kernel_stats.csv')):
        ks = find(os.path.join(src, sub, '**', '*kernel_stats.csv'))
        if ks:
            rows = list(csv.reader(open(ks)))
            with open(os.path.join(dst, out), 'w', newline='') as f:
                w = csv.writer(f)
                for r in rows:
                    w.writerow([c[:100] for c in r])
    for extra in ('wg_timeline_n1e6.json', 'bench_driver_cmd.json', 'cold_start.json', 'n_sweep.json', 'bench_combat_e12500.json', 'bench_combat_e1e5.json', 'bench_tracking.json', 'bench_control.json'):
        if os.path.exists(os.path.join(src, extra)):
            lines = open(os.path.join(src, extra)).read().splitlines()
            contract = [ln for ln in lines if ln.startswith('{')]
            details = [ln[len('BENCH_DETAILS '):] for ln in lines if ln.startswith('BENCH_DETAILS ')]
            if extra.startswith('bench') and contract:      # bench.py: the contract line (last) + everything measured on the BENCH_DETAILS line
                open(os.path.join(dst, f'{tag}_{extra}'), 'w').write(contract[-1] + '\n')
                if details:
                    json.dump(json.loads(details[-1]), open(os.path.join(dst, f'{tag}_{extra[:-5]}_details.json'), 'w'), indent=1)
            else:
                shutil.copy(os.path.join(src, extra), os.path.join(dst, f'{tag}_{extra}'))
    # PlanningEnv: kernel stats per controller numerics, and the PMC passes over the persistent kernel
    for sub in ('stats_planning_i8', 'stats_planning_fp32', 'stats_planning_i8_n1e4'):
        ks = find(os.path.join(src, sub, '**', '*kernel_stats.csv'))
        if ks:
            rows = list(csv.reader(open(ks)))
            with open(os.path.join(dst, f'{tag}_{sub[6:]}_kernel_stats.csv'), 'w', newline='') as f:
                w = csv.writer(f)
                for r in rows:
                    w.writerow([c[:110] for c in r])
        lg = os.path.join(src, sub + '.log')
        if os.path.exists(lg):
            shutil.copy(lg, os.path.join(dst, f'{tag}_{sub[6:]}.log'))
    prow = []
    for d in sorted(glob.glob(os.path.join(src, 'planning_pmc_*'))):
        cc = find(os.path.join(d, '**', '*counter_collection.csv')) if os.path.isdir(d) else None
        if not cc:
            continue
        nm = os.path.basename(d).split('_')[2]
        acc = {}
        for r in csv.DictReader(open(cc)):
            if 'planning_persistent_kernel' not in r.get('Kernel_Name', ''):
                continue
            a = acc.setdefault(r['Counter_Name'], {})
            a[r['Dispatch_Id']] = a.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
        for key, a in acc.items():
            vals = list(a.values())
            prow.append((nm, key, len(vals), sum(vals) / len(vals)))
    if prow:
        with open(os.path.join(dst, f'{tag}_planning_pmc.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['controller_numerics', 'counter', 'launches', 'mean_per_launch (summed over the chip; the persistent kernel, n = 8192)'])
            for row in sorted(prow):
                w.writerow([row[0], row[1], row[2], f'{row[3]:.6g}'])
    # PMC passes: mean per launch / per wave for the dominant kernel
    summary, traffic = [], {}
    for d in sorted(glob.glob(os.path.join(src, 'pmc_*'))):
        if not os.path.isdir(d):
            continue
        cc = find(os.path.join(d, '**', '*counter_collection.csv'))
        if not cc:
            continue
        acc = {}
        for r in csv.DictReader(open(cc)):
            if 'f16_env_kernel' not in r.get('Kernel_Name', '') or 'true, true' not in r['Kernel_Name']:
                continue
            key = r['Counter_Name']
            a = acc.setdefault(key, {})
            a.setdefault(r['Dispatch_Id'], 0.0)
            a[r['Dispatch_Id']] += float(r['Counter_Value'])
            grid = int(r.get('Grid_Size', 0) or 0)
            acc[key]['_waves'] = grid / 64 if grid else None
        for key, a in acc.items():
            waves = a.pop('_waves', None)
            vals = list(a.values())
            mean = sum(vals) / len(vals)
            summary.append((key, len(vals), mean, mean / waves if waves else float('nan')))
            if key in ('FETCH_SIZE', 'WRITE_SIZE'):
                traffic[key] = mean
    if summary:
        with open(os.path.join(dst, f'{tag}_pmc_summary.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['counter', 'launches', 'mean_per_launch', 'mean_per_wave'])
            for row in sorted(summary):
                w.writerow([row[0], row[1], f'{row[2]:.6g}', f'{row[3]:.4g}'])
    # correction factors measured on this access pattern (tools/calibrate_counters.sh -> profiles/r03_counter_calibration.json); the
    # guide's x2 / x1 when no calibration file exists
    rf, wf, cal_src = 2.0, 1.0, 'MI355X_MICROARCH.md (FETCH_SIZE x2, WRITE_SIZE uncalibrated)'
    cal = os.path.join(dst, 'r03_counter_calibration.json')
    if os.path.exists(cal):
        c = json.load(open(cal))['env_kernel_factors']
        rf, wf, cal_src = float(c['read']), float(c['write']), 'profiles/r03_counter_calibration.json (known-byte-count kernels in the env kernel\'s access pattern)'
    if 'FETCH_SIZE' in traffic and 'WRITE_SIZE' in traffic:
        n = json.load(open(os.path.join(dst, f'{tag}_bench.json')))['config']['aircraft_per_gpu'] if os.path.exists(os.path.join(dst, f'{tag}_bench.json')) else 1000000
        json.dump({'round': int(tag[1:3]) if tag[1:3].isdigit() else None, 'kernel': 'f16_env_kernel<0, 0, true, true, 128, 2, false, 3> (pair variant, three waves per SIMD)', 'n': n, 'task': 'heading', 'FETCH_SIZE_KB': traffic['FETCH_SIZE'],
                   'WRITE_SIZE_KB': traffic['WRITE_SIZE'],
                   'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh), mean over the '
                           'cached-kernel launches. MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 128-B requests at '
                           '64 B -> doubled; WRITE_SIZE taken as is (uncalibrated).',
                   'read_factor': rf, 'write_factor': wf, 'factor_source': cal_src,
                   'read_bytes_per_launch': rf * traffic['FETCH_SIZE'] * 1024.0, 'write_bytes_per_launch': wf * traffic['WRITE_SIZE'] * 1024.0,
                   'traffic_bytes_per_launch': (rf * traffic['FETCH_SIZE'] + wf * traffic['WRITE_SIZE']) * 1024.0},
                  open(os.path.join(dst, f'{tag}_pmc_traffic.json'), 'w'), indent=1)
    print('profiles/ updated for', tag, ':', sorted(f for f in os.listdir(dst) if f.startswith(tag)))


if __name__ == '__main__':
    main(sys.argv[1])
