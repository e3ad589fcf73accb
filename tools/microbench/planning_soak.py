"""Soak of np_planning_inner_loop's row groups: many macro-steps, compared with the launch-by-launch path (a missing dependency
between the groups' streams would show as a mismatch).   python tools/microbench/planning_soak.py [n ...]"""
import sys, torch
import numpy as np
sys.path.insert(0, '.')
from neuralplane_amd.envs.planning_env import PlanningEnv
from neuralplane_amd.actor import FusedActor, NUM_FLOATS

w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
for n in [int(x) for x in sys.argv[1:]] or [10_037, 20_011, 50_001]:
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=3, device='cuda:0', controller=FusedActor(w, 'cuda:0')) for _ in range(2)]
    envs[0].use_inner_loop = False
    g = torch.Generator(device='cuda').manual_seed(n)
    bad = 0
    for k in range(120):
        a = torch.rand((n, 3), generator=g, device='cuda') * 2 - 1
        outs = [e.step(a) for e in envs]
        if k % 10 == 9:
            same = all(torch.equal(x, y) for x, y in zip(outs[0][:5], outs[1][:5])) and torch.equal(envs[0].model.s, envs[1].model.s) \
                and torch.equal(envs[0].ego_rnn_states, envs[1].ego_rnn_states)
            bad += 0 if same else 1
    print(f'n={n}: 120 macro-steps, mismatching checkpoints {bad}, terminations equal {envs[0].termination_counts() == envs[1].termination_counts()}', flush=True)
    del envs
