"""Two PROCESSES run PlanningEnv's guest schedule (a grid of one eight-wave workgroup per CU each, spinning on progress words) on ONE GPU at
the same time: neither grid can be fully resident.  Forward progress then rests on the argument in np_planning.hip's header (a workgroup
waits only for a lower-indexed one, dispatched before it).  Each process checks its results against the launch-by-launch path.
    timeout 300 python tools/microbench/planning_two_procs.py [n] [steps] [procs]"""
import subprocess, sys, time

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    import numpy as np
    sys.path.insert(0, '.')
    from neuralplane_amd.envs.planning_env import PlanningEnv
    from neuralplane_amd.actor import FusedActor, NUM_FLOATS
    n, steps, mode, rank = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=3 + rank, device='cuda:0', controller=FusedActor(w, 'cuda:0')) for _ in range(2)]
    envs[0].loop_mode, envs[1].loop_mode = 'launches', mode
    envs[1].loop_waves = 8
    g = torch.Generator(device='cuda').manual_seed(n + rank)
    bad = 0
    t0 = time.perf_counter()
    for k in range(steps):
        a = torch.rand((n, 3), generator=g, device='cuda') * 2 - 1
        o1 = envs[1].step(a)          # the persistent kernel first: the two processes' grids overlap in time
        o0 = envs[0].step(a)
        if k % 10 == 9:
            same = all(torch.equal(x, y) for x, y in zip(o0[:5], o1[:5])) and torch.equal(envs[0].model.s, envs[1].model.s)
            bad += 0 if same else 1
    torch.cuda.synchronize()
    print(f'process {rank}: {mode} n={n}: {steps} macro-steps in {time.perf_counter() - t0:.1f} s, mismatching checkpoints {bad}', flush=True)
    sys.exit(1 if bad else 0)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
procs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for mode in ('guests', 'queue'):
    ps = [subprocess.Popen([sys.executable, __file__, 'child', str(n), str(steps), mode, str(r)]) for r in range(procs)]
    rcs = [p.wait() for p in ps]
    print(f'{mode}: {procs} processes, exit codes {rcs}', flush=True)
    if any(rcs):
        sys.exit(1)
