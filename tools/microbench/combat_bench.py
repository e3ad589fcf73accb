"""SingleCombat macro-step throughput / latency.  usage: combat_bench.py [E ...] (engagements)"""
import os, sys, time, torch
sys.path.insert(0, '.')
from neuralplane_amd.core import F16CombatBatch
from neuralplane_amd.envs.utils.utils import parse_config
sizes = [int(x) for x in sys.argv[1:]] or [100_000, 500_000]
for E in sizes:
    for variant in (os.environ['COMBAT_VARIANTS'].split(',') if os.environ.get('COMBAT_VARIANTS') else (('latency', 'throughput', 'pair') if E <= 65536 else ('throughput', 'pair'))):
        b = F16CombatBatch(E, parse_config('selfplay'), 'cuda:0', seed=1)
        b.set_kernel_variant(variant)
        b.reset()
        a = torch.rand(2 * E, 4, device='cuda') * 2 - 1
        t0 = time.time()
        while time.time() - t0 < 0.3:     # clock-governor ramp (bench.py's prelude)
            for _ in range(8): b.step(a)
            torch.cuda.synchronize()
        b.set_timing(True)
        torch.cuda.synchronize(); t0 = time.time()
        K = 50 if E > 65536 else 300
        for _ in range(K): b.step(a)
        torch.cuda.synchronize(); dt = (time.time() - t0) / K
        ms, cnt = b.get_timing()
        print(f'E={E} {variant}: {dt*1e3:.3f} ms/env.step wall, kernel {ms:.3f} ms; aircraft-FDM-steps/s = {2*E*5/dt:.3e}; env-steps/s (pairs) = {E/dt:.3e}')
        del b
