#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of the HIP extension (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py [extra hipcc flags ...]        # compiles every translation unit (neuralplane_amd/build.py::SOURCES) to a scratch .so

A kernel that starts spilling shows up here (ScratchSize > 0) before it shows up in the HBM traffic counters.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from neuralplane_amd import build
    src = [os.path.join(build.CSRC, s) for s in build.SOURCES]
    with tempfile.TemporaryDirectory() as td:
        cmd = [build._hipcc()] + build.FLAGS + sys.argv[1:] + ['-Rpass-analysis=kernel-resource-usage', '-o', os.path.join(td, 'x.so')] + src
        r = subprocess.run(cmd, cwd=build.CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.exit(r.stdout)
    blocks = re.split(r'remark: [^\n]*Function Name: ', r.stdout)[1:]
    keys = [('VGPRs', 'vgpr'), ('AGPRs', 'agpr'), ('SGPRs', 'sgpr'), (r'ScratchSize \[bytes/lane\]', 'scratch'),
            (r'Occupancy \[waves/SIMD\]', 'occ'), (r'LDS Size \[bytes/block\]', 'lds')]
    print(' '.join(f'{k:>7}' for _, k in keys) + '  kernel')
    for b in blocks:
        name = b.split('\n')[0].strip().split(' ')[0]
        vals = []
        for pat, _ in keys:
            m = re.search(pat + r': (\d+)', b)
            vals.append(m.group(1) if m else '?')
        short = subprocess.run(['c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip() or name
        short = re.sub(r'\(npf16::KArgs\)|\(npf16::CombatArgs\)', '', short)
        print(' '.join(f'{v:>7}' for v in vals) + '  ' + short[:120])


if __name__ == '__main__':
    main()
