#!/usr/bin/env python3
"""gpurun_out/calib (tools/calibrate_counters.sh) -> profiles/r03_counter_calibration.json: per access pattern of
tools/microbench/counter_calib.hip the bytes a launch is known to move against what rocprofv3's FETCH_SIZE / WRITE_SIZE
(KB, mean over the launches) report, and the correction factor = known / reported.

    python tools/summarize_calibration.py [tag]
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def known_bytes(n):
    return {
        'rd_dword<19>': (19 * 4 * n, 0), 'wr_dword<19>': (0, 19 * 4 * n), 'rd_dword<14>': (14 * 4 * n, 0), 'wr_dword<14>': (0, 14 * 4 * n),
        'rd_u8': (3 * n, 0), 'wr_u8': (0, 3 * n), 'rd_i64': (8 * n, 0), 'wr_i64': (0, 8 * n), 'rd_act': (16 * n, 0),
        'wr_obs': (0, 88 * n), 'rd_vec16': (16 * n, 0), 'mix': (159 * n, 235 * n),
    }


def collect(d):
    cc = sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True))
    out = {}
    if not cc:
        return out
    for r in csv.DictReader(open(cc[0])):
        name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').strip()
        out.setdefault(name, {}).setdefault(r['Dispatch_Id'], 0.0)
        out[name][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: sorted(v.values()) for k, v in out.items()}


def main(tag='r03'):
    src = os.path.join(ROOT, 'gpurun_out', 'calib')
    log = open(os.path.join(src, 'FETCH_SIZE.log')).read()
    n = int(re.search(r'n=(\d+)', log).group(1))
    fetch, write = collect(os.path.join(src, 'FETCH_SIZE')), collect(os.path.join(src, 'WRITE_SIZE'))
    rows = {}
    for name, (rb, wb) in known_bytes(n).items():
        f, w = fetch.get(name), write.get(name)
        med = lambda v: v[len(v) // 2] * 1024.0 if v else None  # noqa: E731  (KB -> bytes; median over the launches)
        fm, wm = med(f), med(w)
        rows[name] = {'known_read_bytes': rb, 'known_write_bytes': wb, 'FETCH_SIZE_bytes': fm, 'WRITE_SIZE_bytes': wm,
                      'read_factor': (rb / fm) if (rb and fm) else None, 'write_factor': (wb / wm) if (wb and wm) else None}
    res = {'n': n, 'tool': 'tools/microbench/counter_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only)',
           'unit_note': 'rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; factor = known bytes / reported bytes (multiply a reading by it)',
           'patterns': rows}
    mix = rows['mix']
    res['env_kernel_factors'] = {'read': mix['read_factor'], 'write': mix['write_factor'],
                                 'note': "the 'mix' kernel reproduces f16_env_kernel<.., CACHED>'s per-aircraft access pattern; these two factors "
                                         'replace the guide\'s x2 / x1 in tools/summarize_profile.py'}
    json.dump(res, open(os.path.join(ROOT, 'profiles', f'{tag}_counter_calibration.json'), 'w'), indent=1)
    for k, v in rows.items():
        print(f"{k:14s} read x{v['read_factor'] if v['read_factor'] else float('nan'):.3f}  write x{v['write_factor'] if v['write_factor'] else float('nan'):.3f}")


if __name__ == '__main__':
    main(*sys.argv[1:])
