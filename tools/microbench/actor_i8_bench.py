"""Stand-alone controller call: fp32 kernels against the block-fixed-point kernel (np_actor_i8.h), microseconds per call, back to back.
    python tools/microbench/actor_i8_bench.py [n ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuralplane_amd.actor import NUM_FLOATS, FusedActor  # noqa: E402

w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
for n in [int(x) for x in sys.argv[1:]] or [8192, 10000, 16384, 65536, 262144]:
    obs = torch.randn(n, 22, device='cuda')
    m = torch.ones(n, 1, device='cuda')
    out = {}
    for numerics in ('fp32', 'i8'):
        fa = FusedActor(w, 'cuda:0', numerics=numerics)
        h = [torch.zeros(n, 1, 128, device='cuda'), torch.zeros(n, 1, 128, device='cuda')]
        act = torch.empty(n, 4, device='cuda')
        for k in range(20):
            fa(obs, h[k & 1], m, out=(act, h[(k + 1) & 1]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        for k in range(K):
            fa(obs, h[k & 1], m, out=(act, h[(k + 1) & 1]))
        torch.cuda.synchronize()
        out[numerics] = 1e6 * (time.perf_counter() - t0) / K
    print(f'n = {n}: fp32 {out["fp32"]:.1f} us, i8 {out["i8"]:.1f} us per call ({out["fp32"] / out["i8"]:.2f} x)', flush=True)
