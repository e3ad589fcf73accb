"""Mid-size batches (the sizes the reference trains at, 3 000 - 262 144): kernel microseconds per env.step by HIP events attached to
each dispatch, and wall microseconds per back-to-back step, for every kernel variant (bit-identical results).

    python tools/microbench/mid_n.py [--variants auto,latency8,latency,latency2,pair,throughput] [--task heading] [--out f.json] [n ...]

NPF16_PAIR_WAVES=2|3 (process-wide) pins the pair variant's build; run the script once per setting.
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, '.')
from neuralplane_amd.envs.control_env import ControlEnv

ap = argparse.ArgumentParser()
ap.add_argument('sizes', nargs='*', type=int)
ap.add_argument('--variants', default="auto,latency8,latency,latency2,pair,throughput")
ap.add_argument('--task', default='heading')
ap.add_argument('--steps', type=int, default=400)
ap.add_argument('--out', default='')
args = ap.parse_args()
sizes = args.sizes or [3000, 10000, 30000, 49152, 65536, 81920, 100000, 131072, 196608, 262144]
rows = []
for n in sizes:
    for variant in args.variants.split(','):
        env = ControlEnv(num_envs=n, config=args.task, model='F16', random_seed=0, device='cuda:0')
        env._batch.set_kernel_variant(variant)
        env.reset()
        a = torch.rand(n, 4, device='cuda') * 2 - 1
        t_end = time.perf_counter() + 0.15          # prelude: clocks up
        while time.perf_counter() < t_end:
            env.step(a)
        torch.cuda.synchronize()
        K = args.steps
        t0 = time.perf_counter()
        for _ in range(K):
            env.step(a)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / K * 1e6
        env._batch.set_timing(True)
        for _ in range(K):
            env.step(a)
        torch.cuda.synchronize()
        smp = sorted(env._batch.get_timing_samples())
        env._batch.set_timing(False)
        med, mn = smp[len(smp) // 2] * 1e3, smp[0] * 1e3
        rows.append({'n': n, 'variant': variant, 'pair_waves_env': os.environ.get('NPF16_PAIR_WAVES', ''), 'kernel_us_median': med, 'kernel_us_min': mn,
                     'wall_us_per_step': wall, 'aircraft_steps_per_s_wall': n / wall * 1e6, 'aircraft_steps_per_s_kernel': n / med * 1e6})
        print(f"N={n:7d} {variant:10s} kernel {med:7.1f} us (min {mn:6.1f})  wall {wall:7.1f} us  {n / med * 1e6:.3e} /s kernel", flush=True)
        del env
if args.out:
    json.dump({'task': args.task, 'steps': args.steps, 'rows': rows}, open(args.out, 'w'), indent=1)
