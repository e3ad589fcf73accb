import sys, time, torch
sys.path.insert(0, '.')
from neuralplane_amd.envs.control_env import ControlEnv
for task, n, steps in (('heading', 200_001, 12000), ('tracking', 131_073, 6000), ('control', 1_000_000, 1500)):
    envs = {}
    for v in ('pair', 'throughput'):
        e = ControlEnv(num_envs=n, config=task, model='F16', random_seed=3, device='cuda:0'); e._batch.set_kernel_variant(v); e.reset(); envs[v] = e
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    t0 = time.time(); bad = 0
    for t in range(steps):
        a = torch.rand((n, 4), generator=g, device='cuda') * 2.4 - 1.2
        ra, rb = envs['pair'].step(a), envs['throughput'].step(a)
        if t % 500 == 499 or t == steps - 1:
            same = all(torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y) for x, y in zip(ra[:5], rb[:5]))
            same = same and torch.equal(envs['pair'].model.s.view(torch.int32), envs['throughput'].model.s.view(torch.int32))
            bad += 0 if same else 1
    ca, cb = envs['pair'].termination_counts(), envs['throughput'].termination_counts()
    print(task, n, steps, 'steps: mismatching checkpoints', bad, 'counters equal', ca == cb, 'terminations', sum(ca.values()), '%.1f s' % (time.time() - t0))
