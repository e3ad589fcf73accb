"""Soak of SingleCombat's dual family (np_combat_lat.hip) against the pair / latency kernels: thousands of env.steps on BASELINE config 5's
per-GPU sizes and odd ones, compared bit for bit every 250 steps (state, controller state, blood, observation, reward, flags)."""
import sys, time, torch
sys.path.insert(0, '.')
from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for E, variants in ((12_500, ('dual8', 'pair')), (25_000, ('dual4', 'pair')), (8_193, ('dual8', 'latency')), (16_385, ('dual4', 'dual8')), (31_999, ('dual4', 'throughput'))):
    envs = []
    for v in variants:
        e = SingleCombatEnv(num_envs=E, config='selfplay', random_seed=11, device='cuda:0'); e._batch.set_kernel_variant(v); e.reset(); envs.append(e)
    g = torch.Generator(device='cuda'); g.manual_seed(E)
    t0 = time.time(); bad = 0
    for t in range(steps):
        a = torch.rand((2 * E, 4), generator=g, device='cuda') * 2.4 - 1.2
        ra, rb = envs[0].step(a), envs[1].step(a)
        if t % 250 == 249 or t == steps - 1:
            eq = lambda x, y: torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
            same = all(eq(x, y) for x, y in zip(ra[:5], rb[:5])) and eq(envs[0].s, envs[1].s) and eq(envs[0].blood, envs[1].blood) and eq(envs[0]._batch.pid, envs[1]._batch.pid)
            bad += 0 if same else 1
    torch.cuda.synchronize()
    print(f'{E} engagements, {variants[0]} vs {variants[1]}: {steps} env.steps, mismatching checkpoints {bad}, counters equal '
          f'{envs[0]._batch.termination_counts() == envs[1]._batch.termination_counts()}, {time.time() - t0:.1f} s', flush=True)
    del envs
