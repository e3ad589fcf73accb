#!/usr/bin/env python3
"""Per-launch kernel durations from process start (why is the driver's `--steps 20 --warmup 5` window slower than a 200-step one?).

    python tools/microbench/cold_start.py [--n 1000000] [--steps 120] [--prelude none|spin|steps] [--trace]

Prints one JSON line: the HIP-event duration of every np_f16_step launch in launch order, wall time per step, and with --trace
the effective shader clock (shader-clock counter delta / 100 MHz counter delta) of the first and last workgroups of some launches.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=1_000_000)
    ap.add_argument('--steps', type=int, default=120)
    ap.add_argument('--prelude', default='none', choices=['none', 'spin', 'sleep'])
    ap.add_argument('--trace', action='store_true')
    ap.add_argument('--sync-every', type=int, default=0, help='torch.cuda.synchronize() every k steps (0: only at the end)')
    args = ap.parse_args()
    t_start = time.perf_counter()
    from neuralplane_amd.envs.control_env import ControlEnv
    dev = torch.device('cuda', 0)
    n = args.n
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device=str(dev))
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    pool = [torch.rand((n, 4), generator=g, device=dev) * 2 - 1 for _ in range(8)]
    b = env._batch
    if args.prelude == 'spin':   # ~100 ms of unrelated GPU work first (clock ramp?)
        x = torch.rand((4096, 4096), device=dev)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:
            x = x @ x
            x = x / x.abs().max()
        torch.cuda.synchronize()
    elif args.prelude == 'sleep':
        torch.cuda.synchronize()
        time.sleep(2.0)
    trace = None
    if args.trace:
        import ctypes as C
        from neuralplane_amd import _lib
        cap = (n + 63) // 64
        trace = torch.zeros((cap, 6), dtype=torch.int64, device=dev)
        _lib.check(b.lib.np_f16_set_trace(b._ctx, C.c_void_p(trace.data_ptr()), cap))
    env.reset()
    b.set_timing(True)
    torch.cuda.synchronize()
    t_ready = time.perf_counter()
    walls = []
    clocks = []
    for i in range(args.steps):
        env.step(pool[i % 8])
        if args.sync_every and (i + 1) % args.sync_every == 0:
            torch.cuda.synchronize()
            walls.append(time.perf_counter())
        if trace is not None and i in (0, 1, 2, 5, 10, 20, 50, args.steps - 1):
            torch.cuda.synchronize()
            t = trace.cpu()
            wgs = (n + 127) // 128
            t = t[:wgs]
            dc = (t[:, 2] - t[:, 0]).double()
            dr = (t[:, 4] - t[:, 3]).double()
            mhz = (dc / dr * 100.0)
            span_us = float((t[:, 4].max() - t[:, 3].min()).item()) / 100.0
            clocks.append({'launch': i, 'mhz_median': float(mhz.median()), 'mhz_min': float(mhz.min()), 'mhz_max': float(mhz.max()),
                           'wg_us_median': float((dr / 100.0).median()), 'span_us': span_us,
                           'delay_cycles_max': int((t[:, 1] - t[:, 0]).max())})
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    samples = b.get_timing_samples()
    out = {'n': n, 'steps': args.steps, 'prelude': args.prelude, 'startup_s': t_ready - t_start, 'wall_ms_per_step': 1e3 * (t_end - t_ready) / args.steps,
           'kernel_ms': [round(x, 4) for x in samples], 'clocks': clocks}
    k = samples
    if len(k) >= 25:
        out['mean_5_25'] = sum(k[5:25]) / 20
        out['mean_last20'] = sum(k[-20:]) / 20
        out['median_all'] = sorted(k)[len(k) // 2]
    print(json.dumps(out))


if __name__ == '__main__':
    main()
