"""Soak of the persistent PlanningEnv kernel's schedules (np_planning.hip: static / guests / queue / dual) against the launch-by-launch path:
many macro-steps on odd sizes, every tenth compared bit for bit (a race in the coherent imports / exports, the progress words, the park or the
window nets would show as a mismatch; a lost wake-up as a hang — run under `timeout`).
    [NUMERICS=i8|fp32] python tools/microbench/planning_soak_modes.py [steps] [mode:n ...]        # default: 400 steps of the list below"""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, '.')
from neuralplane_amd.envs.planning_env import PlanningEnv
from neuralplane_amd.actor import FusedActor, NUM_FLOATS

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cases = sys.argv[2:] or ['auto:4000', 'auto:8192', 'auto:8193', 'auto:9001', 'auto:10000', 'auto:12288', 'auto:12289', 'auto:14001', 'auto:16384',
                         'queue:10037', 'queue:6000', 'guests:11111', 'persistent:5555', 'dual:15001']
w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
for case in cases:
    mode, n = case.split(':'); n = int(n)
    if mode == 'dual' and os.environ.get('NUMERICS', 'i8') == 'i8':
        continue   # the dual workgroups serve the fp32 controller only (NUMERICS=fp32 runs this case)
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=3, device='cuda:0', controller=FusedActor(w, 'cuda:0', numerics=os.environ.get('NUMERICS', 'i8'))) for _ in range(2)]
    envs[0].loop_mode = 'launches'
    envs[1].loop_mode = mode
    if mode in ('queue', 'guests', 'persistent'):
        envs[1].loop_waves = 8
    g = torch.Generator(device='cuda').manual_seed(n)
    bad = 0
    t0 = time.perf_counter()
    for k in range(steps):
        a = torch.rand((n, 3), generator=g, device='cuda') * 2 - 1
        outs = [e.step(a) for e in envs]
        if k % 10 == 9:
            same = all(torch.equal(x, y) for x, y in zip(outs[0][:5], outs[1][:5])) and torch.equal(envs[0].model.s, envs[1].model.s) \
                and torch.equal(envs[0].ego_rnn_states, envs[1].ego_rnn_states)
            bad += 0 if same else 1
    torch.cuda.synchronize()
    print(f'{os.environ.get("NUMERICS", "i8")} {mode} n={n} (fallbacks {envs[1].loop_fallbacks}): {steps} macro-steps in {time.perf_counter() - t0:.1f} s, mismatching checkpoints {bad}, '
          f'terminations equal {envs[0].termination_counts() == envs[1].termination_counts()}', flush=True)
    del envs
