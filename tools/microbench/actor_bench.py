"""Fused low-level controller alone: microseconds per call and the rate of its 302 KFLOP per row, for the automatic tiling
and for each tiling forced (NP_ACTOR_TILE).  Kernel time by HIP events around 50 back-to-back calls on the current stream."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from neuralplane_amd.actor import FusedActor, NUM_FLOATS
fa = FusedActor(np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32), 'cuda:0')
sizes = [int(x) for x in sys.argv[1:]] or [64, 1024, 4096, 8192, 10000, 16384, 32768, 65536, 262144, 1048576]   # e.g. `actor_bench.py 262144` under rocprofv3
for n in sizes:
    obs = torch.randn(n, 22, device='cuda'); h0 = torch.zeros(n, 1, 128, device='cuda'); m = torch.ones(n, 1, device='cuda')
    out = [(torch.empty(n, 4, device='cuda'), torch.empty(n, 1, 128, device='cuda')) for _ in range(2)]
    line = f'n={n:8d}:'
    for tile in ('', '32', '64'):
        if tile: os.environ['NP_ACTOR_TILE'] = tile
        else: os.environ.pop('NP_ACTOR_TILE', None)
        h = h0
        for it in range(6): _, _, h = fa(obs, h, m, out=out[it & 1])
        torch.cuda.synchronize()
        K = 50 if n <= 262144 else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(K): _, _, h = fa(obs, h, m, out=out[it & 1])
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / K
        line += f'  {tile or "auto":>4}: {dt*1e6:8.1f} us {n*301.6e3/dt/1e12:6.1f} TFLOP/s'
    print(line, flush=True)
