"""PlanningEnv.step with the fp32 controller against the block-fixed-point one, per schedule: ms per macro-step (back to back, 20 steps).
    python tools/microbench/planning_i8_bench.py [n ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuralplane_amd.actor import NUM_FLOATS, FusedActor  # noqa: E402
from neuralplane_amd.envs.planning_env import PlanningEnv  # noqa: E402

w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
modes = os.environ.get('MODES', 'auto').split(',')
for n in [int(x) for x in sys.argv[1:]] or [4096, 8192, 10000, 12288, 16384, 20000, 65536, 262144]:
    line = f'n = {n}:'
    for numerics in ('fp32', 'i8'):
        for mode in modes:
            env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cuda:0', controller=FusedActor(w, 'cuda:0', numerics=numerics))
            env.loop_mode = mode
            a = torch.rand(n, 3, device='cuda') * 2 - 1
            K = 20 if n <= 65536 else 4
            try:
                for _ in range(3):
                    env.step(a)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(K):
                    env.step(a)
                torch.cuda.synchronize()
                line += f'  {numerics}/{mode} {1e3 * (time.perf_counter() - t0) / K:.3f} ms'
            except RuntimeError as e:
                line += f'  {numerics}/{mode} n/a'
            del env
    print(line, flush=True)
