#!/usr/bin/env python3
"""A/B timing of kernel builds INSIDE ONE gpurun session (box-to-box spread is +-3-5 %): every tools/microbench/libs/*.so
(or the names given) runs the headline workload in its own process through NPF16_LIB, interleaved over `--rounds` rounds.

    python tools/microbench/ab_libs.py [--rounds 2] [--n 1000000] [--steps 100] [name ...]

Build variants in the build container, e.g.
    NPF16_EXTRA_FLAGS="-DNPF16_EXP=1" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
    cp neuralplane_amd/csrc/libneuralplane_hip.so tools/microbench/libs/nonoise.so           (libs/ is git-ignored)
Timing-only switches: NPF16_EXP bit 1 = no observation noise, 2 = no Overload re-evaluation, 4 = no MLP evaluation, 8 = hardware
sin/cos/pow; NPF16_DIVC=0 IEEE divisions / 9 multiply by the reciprocal; NPF16_PAIR_STAGGER, NPF16_PAIR_GROUPS.
"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIBS = os.path.join(ROOT, 'tools', 'microbench', 'libs')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('names', nargs='*')
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--n', type=int, default=1_000_000)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--task', default='heading')
    ap.add_argument('--extra', default='')
    args = ap.parse_args()
    names = args.names or sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(LIBS, '*.so')))
    res = {n: [] for n in names}
    for _ in range(args.rounds):
        for n in names:
            env = dict(os.environ, NPF16_LIB=os.path.join(LIBS, n + '.so'))
            cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--headline-only', '--steps', str(args.steps), '--warmup', '5', '--n', str(args.n),
                   '--task', args.task] + args.extra.split()
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            # bench.py prints everything it measured on the BENCH_DETAILS line and the small contract line last (round 6)
            line = [ln[len('BENCH_DETAILS '):] for ln in r.stdout.splitlines() if ln.startswith('BENCH_DETAILS ')] or [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not line:
                res[n].append(None)
                print(n, 'FAILED', r.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            res[n].append((d['roofline']['kernel_avg_ms'], d['roofline']['kernel_median_ms'], d['value'], d['cold_start']['kernel_avg_ms']))
            print(f"{n:28s} kernel avg {res[n][-1][0]:.4f} median {res[n][-1][1]:.4f} ms  value {res[n][-1][2]:.3e}  cold {res[n][-1][3]:.4f}", flush=True)
    print('--- summary (best median over rounds) ---')
    base = None
    for n in names:
        ok = [x for x in res[n] if x]
        if not ok:
            continue
        best = min(x[1] for x in ok)
        base = base or best
        print(f'{n:28s} {best:.4f} ms  {100 * (best / base - 1):+.1f} % vs {names[0]}')


if __name__ == '__main__':
    main()
