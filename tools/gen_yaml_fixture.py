#!/usr/bin/env python3
"""Record the key/value sets of the reference's scenario and PID YAMLs as a data fixture (build container only).

    python tools/gen_yaml_fixture.py        # reads /root/reference, writes tests/golden/ref_yaml_configs.json

The shipped YAMLs under neuralplane_amd/envs/configs are re-written files (own comments and order) that must carry the same
keys and values as the reference's (envs/configs/*.yaml, algorithms/pid/config/*.yaml): tests/test_host_logic_cpu.py compares
them against this fixture, so a typo in a constant cannot hide behind "same kernel, different scenario".
"""
import json
import os
import sys

import yaml

REF = os.environ.get('NP_REFERENCE', '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENARIOS = ('heading', 'control', 'tracking', 'selfplay')
PIDS = ('rollcontroller', 'pitchcontroller', 'yawcontroller')


def main():
    out = {'scenarios': {}, 'pid': {}}
    for name in SCENARIOS:
        with open(os.path.join(REF, 'envs', 'configs', name + '.yaml'), encoding='utf-8') as f:
            out['scenarios'][name] = yaml.load(f, Loader=yaml.FullLoader)
    for name in PIDS:
        with open(os.path.join(REF, 'algorithms', 'pid', 'config', name + '.yaml'), encoding='utf-8') as f:
            out['pid'][name] = yaml.load(f, Loader=yaml.FullLoader)
    dst = os.path.join(ROOT, 'tests', 'golden', 'ref_yaml_configs.json')
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote', dst)


if __name__ == '__main__':
    sys.exit(main())
