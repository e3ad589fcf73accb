"""Latency regime: microseconds per env.step for small batches, the three kernel variants (bit-identical results)."""
import sys, time, torch
sys.path.insert(0, '.')
from neuralplane_amd.envs.control_env import ControlEnv
sizes = [int(x) for x in sys.argv[1:]] or [256, 4096, 16384, 32768, 49152, 65536, 98304, 131072]
for n in sizes:
    row = []
    for variant in ('latency8', 'latency', 'throughput', 'pair'):
        env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0')
        env._batch.set_kernel_variant(variant)
        env.reset()
        a = torch.rand(n, 4, device='cuda') * 2 - 1
        for _ in range(50): env.step(a)
        env._batch.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); K = 1000
        for _ in range(K): env.step(a)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ms, cnt = env._batch.get_timing()
        row.append(f'{variant}: {dt/K*1e6:.1f} us/step wall ({t_host/K*1e6:.1f} host), kernel {ms*1e3:.1f} us')
        del env
    print(f'N={n}: ' + ' | '.join(row))
