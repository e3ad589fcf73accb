#!/usr/bin/env python3
"""EXPERIMENT (CPU only, test infrastructure): the oracle with the aero MLPs' bias added LAST (F16O_MODE_BIAS_LAST — the order of a GEMM with a bias
epilogue; never the shipped spec, never compared with the kernels) against the reference-CPU fixtures, beside the shipped spec: the closed-loop
1000-step fixture, the open-loop 1000-step fixture and the PID-flown 2600-step fixture (tools/parity_report.py).  Round 6:
profiles/r06h_bias_last_vs_reference_cpu.json — 2-4 x closer to the reference (closed loop max 1.27e-5 -> 7.0e-6; PID-flown t = 2000 max 6.3e-5 -> 1.6e-5).
    python tools/microbench/bias_last_report.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.f16_oracle import MODE_BIAS_LAST, Oracle  # noqa: E402
from tools import parity_report as pr  # noqa: E402


class BiasLastEngine(pr.OracleEngine):
    name = 'oracle, bias-last experiment'

    def __init__(self, task, n, overrides=None):
        super().__init__(task, n, overrides)
        self.o.mode = MODE_BIAS_LAST

    @staticmethod
    def xdot(s, u):
        return Oracle('heading', mode=MODE_BIAS_LAST).nlplant(np.hstack([s, u]).astype(np.float32))


def main():
    out = {}
    for name, cls in (('shipped_spec', pr.OracleEngine), ('bias_last_experiment', BiasLastEngine)):
        cl = pr.closed_loop_report(cls)
        tr = pr.trajectory_report(cls, 'heading', 256, 1000, (1, 10, 100, 426, 1000))
        pid = pr.trajectory_report(cls, 'heading', 64, 2600, (1000, 2000, 2500), fixture='traj_pid_heading_N64_T2600.npz')
        out[name] = {'closed_loop_1000': cl['at'][-1], 'open_loop_1000': tr['at'][-1], 'open_loop_mask_diffs': tr['first_mask_differences'],
                     'pid_2600': pid['at'], 'pid_mask_diffs': pid['first_mask_differences']}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
