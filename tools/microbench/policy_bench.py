#!/usr/bin/env python3
"""FusedPolicy.get_actions (np_policy_act): GPU time per call (HIP events around back-to-back calls) and host enqueue time per call,
beside the same networks in eager torch (tools/collect_loop.py TorchPolicy).   python tools/microbench/policy_bench.py [n ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuralplane_amd.policy import FusedPolicy  # noqa: E402
from tools.collect_loop import TorchPolicy  # noqa: E402


def timed(fn, reps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, host


def main():
    dev = 'cuda:0'
    sizes = [int(x) for x in sys.argv[1:]] or [1000, 3000, 8192, 10000, 16384, 32768, 100000]
    torch.manual_seed(0)
    tp = TorchPolicy().to(dev).eval()
    fp = FusedPolicy(tp.state_dicts(), dev, numerics='fp32')
    f8 = FusedPolicy(tp.state_dicts(), dev, numerics='i8')
    for n in sizes:
        obs = torch.randn((n, 22), device=dev)
        ha, hc = torch.randn((n, 128), device=dev) * 0.3, torch.randn((n, 128), device=dev) * 0.3
        m = torch.ones((n, 1), device=dev)
        eps = torch.randn((n, 4), device=dev)
        g_f, h_f = timed(lambda: fp.get_actions(obs, ha, hc, m), 200)
        g_n, h_n = timed(lambda: fp.get_actions(obs, ha, hc, m, noise=eps), 200)
        g_v, _ = timed(lambda: fp.get_values(obs, hc, m), 200)
        g_8, _ = timed(lambda: f8.get_actions(obs, ha, hc, m, noise=eps), 200)
        g_v8, _ = timed(lambda: f8.get_values(obs, hc, m), 200)
        g_t, h_t = timed(lambda: tp.get_actions(obs, ha, hc, m), 50)
        print(f'n = {n:7d}: get_actions {g_f:7.1f} us GPU / {h_f:5.1f} us host enqueue fp32 chains (given noise: {g_n:7.1f} / {h_n:5.1f}; i8 numerics {g_8:7.1f}); get_values {g_v:6.1f} us (i8 {g_v8:6.1f}); '
              f'eager torch {g_t:7.1f} us GPU / {h_t:6.1f} us host   -> {g_t / g_f:4.1f} x', flush=True)


if __name__ == '__main__':
    main()
