#!/usr/bin/env python3
"""The reference's own headline benchmark, protocol as written (envs/measure_env.py:65-78,112-128): for N = 10^0 .. 10^6 (and 10^7)
build ControlEnv(num_envs=N, 'heading', 'F16', random_seed=0), run 500 back-to-back env.step calls with the constant action of its
INIT_U (clamped to (1, 0, 0, 0)), wall-clock the loop (no warm-up; a device synchronisation at the end, which the reference's implicit
syncs make unnecessary there).  Prints one JSON line; the reference's published times (envs/measure_env/time_neuralplane.npy, the
authors' CUDA run) are quoted beside ours."""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
from neuralplane_amd.envs.control_env import ControlEnv

REFERENCE_PUBLISHED_S = {1: 19.645, 10: 20.344, 100: 18.846, 1000: 18.086, 10000: 18.249, 100000: 21.128, 1000000: 105.163}  # seconds per 500 steps


def main():
    rows = []
    for e in range(8):
        n = 10 ** e
        env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0')
        a = torch.tensor([1.0, 0.0, 0.0, 0.0], device='cuda').repeat(n, 1)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(500):
            env.step(a)
        torch.cuda.synchronize()
        el = time.time() - t0
        mem = torch.cuda.memory_allocated() / 2 ** 20
        ref = REFERENCE_PUBLISHED_S.get(n)
        rows.append({'n': n, 'seconds_per_500_steps': el, 'aircraft_steps_per_s': n * 500 / el, 'us_per_step': 1e6 * el / 500,
                     'device_memory_mb': mem, 'reference_published_seconds': ref, 'speedup_vs_published': (ref / el) if ref else None})
        del env, a
        torch.cuda.empty_cache()
    print(json.dumps({'protocol': 'envs/measure_env.py: 500 steps, constant action, no warm-up', 'rows': rows}))


if __name__ == '__main__':
    main()
