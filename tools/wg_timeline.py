#!/usr/bin/env python3
"""Per-workgroup timeline of one np_f16_step launch (np_f16_set_trace, include/neuralplane_amd.h).

    python tools/wg_timeline.py [--n 1000000] [--prelude 300] [--task heading] [--out gpurun_out/timeline.json] [--raw x.npy]

Runs `prelude` untimed steps (the GPU's clock governor needs ~50-100 ms of load to reach its steady state), then traces one
launch and prints: span, effective shader clock, workgroups resident over time, duration of a workgroup by start order,
load balance across XCDs / CUs, and the start-up and tail losses in workgroup-slot-microseconds.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(t, wgs_per_slotset=None):
    """t: int64 [wgs, 6] trace records -> dict of summary numbers (times in us, 100 MHz counter)."""
    c0, c1, c2, r0, r1, hw = (t[:, k].astype(np.int64) for k in range(6))
    start = (r0 - r0.min()) / 100.0
    end = (r1 - r0.min()) / 100.0
    dur = end - start
    mhz = (c2 - c0) / np.maximum(r1 - r0, 1) * 100.0
    xcc = (hw >> 32) & 0xF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    span = float(end.max())
    # resident workgroups over time (1 us bins)
    bins = np.arange(0.0, span + 1.0, 1.0)
    active = np.zeros(len(bins))
    for s, e in zip(start, end):
        active[int(s):int(e) + 1] += 1
    peak = active.max()
    full = active >= 0.95 * peak
    t_full_first = float(bins[np.argmax(full)])
    t_full_last = float(bins[len(full) - 1 - np.argmax(full[::-1])])
    busy_area = float(dur.sum())                      # workgroup-us actually occupied
    ideal_span = busy_area / peak                     # if `peak` slots were busy from 0 to the end
    order = np.argsort(start)
    q = len(order) // 8
    by_start = [float(np.median(dur[order[k * q:(k + 1) * q]])) for k in range(8)] if q else []
    cu_key = xcc * 1000 + se * 100 + cu
    per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
    per_xcc = np.bincount(xcc.astype(np.int64), minlength=8)
    return {
        'workgroups': int(len(t)), 'span_us': span, 'effective_mhz_median': float(np.median(mhz)),
        'wg_us_median': float(np.median(dur)), 'wg_us_p5': float(np.percentile(dur, 5)), 'wg_us_p95': float(np.percentile(dur, 95)),
        'peak_resident_wgs': int(peak), 'mean_resident_wgs': float(busy_area / span),
        't_95pct_full_first_us': t_full_first, 't_95pct_full_last_us': t_full_last,
        'startup_us': t_full_first, 'tail_us': span - t_full_last,
        'slot_efficiency': ideal_span / span,
        'wg_us_median_by_start_octile': by_start,
        'delay_us_max': float(((c1 - c0) / np.maximum(mhz, 1.0)).max()),
        'cus_seen': int(len(per_cu)), 'wgs_per_cu_min': int(per_cu.min()), 'wgs_per_cu_max': int(per_cu.max()),
        'wgs_per_xcc': per_xcc.tolist(), 'simd_hist_of_wave0': np.bincount(simd.astype(np.int64), minlength=4).tolist(),
        'last_start_us': float(start.max()), 'first_end_us': float(end.min()),
        'wgs_per_cu_hist': np.bincount(per_cu.astype(np.int64)).tolist() if per_cu.max() < 64 else None,
        'idle_cus': int(256 - len(per_cu)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=1_000_000)
    ap.add_argument('--prelude', type=int, default=300)
    ap.add_argument('--task', default='heading')
    ap.add_argument('--out', default=None)
    ap.add_argument('--raw', default=None)
    ap.add_argument('--variant', default=None, help='pin a kernel variant (auto when omitted)')
    args = ap.parse_args()
    from neuralplane_amd import _lib
    from neuralplane_amd.envs.control_env import ControlEnv
    dev = torch.device('cuda', 0)
    n = args.n
    env = ControlEnv(num_envs=n, config=args.task, model='F16', random_seed=0, device=str(dev))
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    pool = [torch.rand((n, 4), generator=g, device=dev) * 2 - 1 for _ in range(8)]
    b = env._batch
    if args.variant:
        b.set_kernel_variant(args.variant)
    env.reset()
    for i in range(args.prelude):
        env.step(pool[i % 8])
    cap = (n + 63) // 64
    trace = torch.zeros((cap, 6), dtype=torch.int64, device=dev)
    _lib.check(b.lib.np_f16_set_trace(b._ctx, C.c_void_p(trace.data_ptr()), cap))
    b.set_timing(True)
    for i in range(3):
        env.step(pool[i % 8])
    torch.cuda.synchronize()
    ms = b.get_timing_samples()
    _lib.check(b.lib.np_f16_set_trace(b._ctx, None, 0))
    t = trace.cpu().numpy()
    t = t[t[:, 3] != 0]
    out = analyse(t)
    out.update({'n': n, 'task': args.task, 'variant': args.variant or 'auto', 'prelude_steps': args.prelude, 'kernel_ms_traced_launches': ms})
    if args.raw:
        np.save(args.raw, t)
    txt = json.dumps(out)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
