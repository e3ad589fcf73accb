"""Fused low-level controller alone: microseconds per call and the rate of its 302 KFLOP per row."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from neuralplane_amd.actor import FusedActor, NUM_FLOATS
fa = FusedActor(np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32), 'cuda:0')
sizes = [int(x) for x in sys.argv[1:]] or [64, 1024, 10000, 16384, 65536, 262144, 1048576]   # e.g. `actor_bench.py 262144` under rocprofv3
for n in sizes:
    obs = torch.randn(n, 22, device='cuda'); h = torch.zeros(n, 1, 128, device='cuda'); m = torch.ones(n, 1, device='cuda')
    for _ in range(5): a, _, h = fa(obs, h, m)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 50
    for _ in range(K): a, _, h = fa(obs, h, m)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f'n={n}: {dt*1e6:.1f} us per call, {n*301.6e3/dt/1e12:.1f} TFLOP/s')
