#!/usr/bin/env python3
"""bench.py — aircraft-steps/sec of the fused F-16 env.step on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n AIRCRAFT_PER_GPU] [--task heading|control|tracking|combat]

`--gpus N` with N > 1 works as a plain command: when no launcher environment is present (WORLD_SIZE unset) the script
re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank
per GPU over RCCL; launched by a driver that way already, it just reads RANK / LOCAL_RANK / WORLD_SIZE.

Workload = BASELINE.json configs[1]: F-16 Heading, N = 1e6 aircraft per GPU (weak scaling: the batch shards embarrassingly,
no data-path collective), envs/configs/heading.yaml constants (Euler, dt = 0.02, noise_scale = 0.01), per-step uniform
random actions from a fixed seed so that terminations / auto-resets occur at a realistic rate.  A "step" is one
`ControlEnv.step(action)` = one fused HIP kernel launch; inputs (state, actions) are resident in HBM before the timed region.
`--task tracking` is configs[3]'s per-GPU shape, `--task control` configs[2]'s stand-in, `--task combat` configs[4]
(SingleCombat 1v1, engagements sharded by env, the opponent-observation all-gather inside the stepped loop).

Protocol (the reference's own benchmark is envs/measure_env.py:65-78: back-to-back env.step, 500 steps, mean):
  1. `cold_start`: reset, W warm-up steps, K timed steps — exactly the requested protocol from an idle GPU.  Reported, not `value`:
     the MI355X clock governor needs ~50-100 ms of sustained load to reach its steady shader clock (2.07 -> 2.39 GHz measured with
     the in-kernel counters, tools/microbench/cold_start.py), so an 8 ms window right after start-up measures the ramp.
  2. `prelude`: untimed env.steps for --prelude-ms (default 300 ms) of GPU load, count reported.
  3. W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides, MAX over ranks -> `value`.
Rank 0 prints `BENCH_DETAILS {...}` (everything measured) and then, LAST, the one contract JSON line (< 4 KB: metric, value,
ms_per_step, config, roofline, cpu_baseline).  `--full` adds the side modes to the details (DESIGN.md §6 has the roofline arithmetic).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic work per aircraft-step, F-16 Heading, Euler
ALGO_BYTES = 278.0      # HBM: read s48+u20+tgt12+step_count8+flags3+action16, write s48+u20+step_count8+obs88+reward4+flags3
ALGO_FLOP = 33.8e3      # aero eval 23 770 + force-side re-evaluation 9 160 + ~900 non-MLP (FMA = 2)
EXEC_FLOP = 25.7e3      # what the kernel executes: 14 force-side coefficients are carried over from the previous step's Overload check
ALGO_FLOP_RK4 = 105e3   # 4 x 23 770 + 9 160 + ~900
# SingleCombat: per aircraft and FDM step the same two aero evaluations + ~300 FLOP of PID stack / terminations / pairwise geometry
ALGO_FLOP_COMBAT_FDM = 34.1e3
ALGO_BYTES_COMBAT = 500.0  # per aircraft and env.step (5 FDM steps): state 136 + controller 44 + action 16 in, the same + obs 60 + reward/flags out
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 vector peak == fp32 (f32-input) MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
METRIC = 'aircraft-steps/sec at N=1e6 F-16 Heading; 1/2/4/8 MI355X scaling'


def pmc_traffic(n, task):
    """(HBM bytes per launch, source file) from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json — bench.py
    cannot collect PMC counters itself); (None, None) unless a profile of exactly this workload exists."""
    import glob
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json'))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if int(d.get('n', -1)) == int(n) and d.get('task', 'heading') == task:
            best = (float(d['traffic_bytes_per_launch']), os.path.relpath(f, ROOT))
    return best


def usable_cpus():
    import math
    cpus = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            cpus = max(1, min(cpus, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
    return cpus


def cpu_baseline(task, budget_s=10.0, eager_budget_s=5.0):
    """The CPU oracle (oracle/, the C restatement of the reference path) timed on this box's host cores on a bounded sample of
    the same workload.  A reported baseline, never the thing shipped."""
    import numpy as np
    cpus = usable_cpus()   # libgomp's default is the machine's CPU count, which oversubscribes a container limited to a few cores
    from oracle.f16_oracle import Oracle
    o = Oracle(task, threads=cpus)  # (OMP_NUM_THREADS is too late here: torch already initialised libgomp)
    n = 4096 * max(4, cpus)
    st = Oracle.new_state(n)
    rng = np.random.RandomState(0)
    acts = [rng.uniform(-1, 1, (n, 4)).astype(np.float32) for _ in range(4)]
    o.reset(st, seed=0, call_idx=0)
    o.step(st, acts[0], seed=0, call_idx=1)  # warm
    t0 = time.perf_counter()
    steps = 0
    while True:
        o.step(st, acts[steps % 4], seed=0, call_idx=2 + steps)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 2000:
            break
    eager = None
    if task == 'heading':
        # a from-scratch eager-PyTorch formulation of the same step on the same host cores (oracle/torch_eager.py — test /
        # bench infrastructure, checked against the reference's fixtures; the reference's own files cannot travel to this box)
        try:
            from oracle.torch_eager import timed_rate
            eager = timed_rate(n=100_000, budget_s=eager_budget_s, threads=cpus)
        except Exception as e:   # the baseline is context, never a reason to lose the bench line
            eager = {'error': repr(e)}
    return {'value': n * steps / el, 'unit': 'aircraft-steps/s', 'cores': int(o.threads), 'kind': 'port',
            'torch_eager': eager,
            'sample': f'F-16 {task}, N={n} aircraft x {steps} steps, oracle/f16_oracle.c (OpenMP, fp32 scalar, '
                      f'same numerics spec), {el:.1f} s',
            # context only: the reference's own eager-PyTorch path cannot travel to this box (its source never leaves the build
            # container); measured there and published by its authors
            'reference_context': {'pytorch_cpu_8_vcpu_N1e5': 4.7e5, 'published_pytorch_cuda_N1e6': 4.75e6, 'unit': 'aircraft-steps/s'}}


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: one rank per GPU under torch.distributed.run on this node."""
    argv = ['--aircraft' if a == '--n' else a for a in sys.argv[1:]]   # torch.distributed.run's parser abbreviates --n
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this driver (RCCL / cross-process tensors)
    env.setdefault('OMP_NUM_THREADS', '1')
    return subprocess.call(cmd, env=env)


class Timer:
    """cold window / prelude / timed region around a `step(i)` callable; `batch` gives the HIP-event kernel times."""

    def __init__(self, step, batch, dev, dist, backend, per_launch=False):
        """per_launch: keep per-dispatch events on the timed launches too — for steps that hold more than the one kernel
        (policies, the opponent exchange), where an event pair around the region would time all of it."""
        self.step, self.batch, self.dev, self.dist, self.backend, self.per_launch = step, batch, dev, dist, backend, per_launch

    def barrier(self):
        import torch
        torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize(self.dev)

    def window(self, warmup, steps, i0=0):
        """W untimed steps, then K timed ones between two barriers.
        -> (elapsed_s max over ranks, Samples, next index)

        Kernel time is measured twice, both with HIP events on the launch stream (= torch's current stream):
          * per launch (start / stop events attached to each dispatch, np_f16_set_timing) on the W warm-up launches and the
            launches of the prelude before them — median / min / max, the quantity rocprofv3 reports per kernel;
          * in the timed region: a per-dispatch pair on launch 1 and ONE event pair spanning launches 2..K;
            (d_1 + span) / K = average launch duration there, the gaps between back-to-back launches (~1.3 us) included.
            Launches 2..K carry no per-dispatch events: each
            pair costs ~4.5 us of dispatch time (tools/microbench/launch_gap.py: 0.3074 vs 0.3029 ms per step at N = 1e6), which
            would be the benchmark measuring its own instrumentation."""
        import torch
        from neuralplane_amd import sharding
        i = i0
        self.batch.set_timing(True)
        for _ in range(warmup):
            self.step(i)
            i += 1
        torch.cuda.synchronize(self.dev)
        samples = Samples(getattr(self, 'recent', []) + list(self.batch.get_timing_samples()))
        self.recent = []
        if not self.per_launch:
            self.batch.set_timing(False)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.barrier()
        t0 = time.perf_counter()
        if self.per_launch:
            for _ in range(steps):
                self.step(i)
                i += 1
        else:
            # launch 1 of K carries its own event pair (the GPU idles for the host's first-call latency before it, which is
            # not kernel time); ONE pair spans launches 2..K.  average launch duration = (d_1 + span_2..K) / K
            self.batch.set_timing(True)
            if steps > 0:
                self.step(i)
                i += 1
            self.batch.set_timing(False)
            ev0.record()
            for _ in range(max(0, steps - 1)):
                self.step(i)
                i += 1
            ev1.record()
        host_enqueue = time.perf_counter() - t0   # this rank's host time to enqueue the K steps (before it waits for the GPU)
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.per_launch:
            samples = Samples(self.batch.get_timing_samples())
            self.batch.set_timing(False)
        else:
            first = list(self.batch.get_timing_samples())
            samples.region_avg_ms = (sum(first) + ev0.elapsed_time(ev1)) / max(1, steps)
            samples.region_launches = steps
        red_dev = self.dev if self.backend == 'nccl' else 'cpu'
        # per-rank diagnostics of the SAME region (read the launch skew of an N-process run off these): each rank's own wall time
        # between the two barriers and its host enqueue time per step
        self.rank_elapsed_s = sharding.gather_over_ranks(elapsed, self.dist, red_dev)
        self.rank_host_us_per_step = [1e6 * v / max(1, steps) for v in sharding.gather_over_ranks(host_enqueue, self.dist, red_dev)]
        elapsed = sharding.max_over_ranks(elapsed, self.dist, red_dev)
        return elapsed, samples, i

    def prelude(self, seconds, i0=0, est_step_s=None, max_steps=20000):
        """Untimed steps for about `seconds` of GPU load (the clock governor's ramp).  The COUNT is fixed up front from a step-time
        estimate that is identical on every rank (the cold window's max-over-ranks time): a step may contain collectives (--task
        combat), so all ranks must run the same number of them.  -> (steps run, wall seconds, next index)"""
        import torch
        i, t0 = i0, time.perf_counter()
        if seconds <= 0:
            return 0, 0.0, i
        n = max_steps if not est_step_s else max(8, min(max_steps, int(seconds / est_step_s + 0.999)))
        for k in range(n):
            if est_step_s and k == n - 64 and self.batch is not None:
                self.batch.set_timing(True)      # per-dispatch events on the last 64 launches: the median / min / max beside the average
            self.step(i)
            i += 1
            if k % 8 == 7:
                torch.cuda.synchronize(self.dev)   # bounds the launch queue; ~10 us of idle per 8 launches
            if not est_step_s and time.perf_counter() - t0 >= seconds:   # single-process callers without an estimate: time-based
                break
        torch.cuda.synchronize(self.dev)
        el = time.perf_counter() - t0
        if est_step_s and n >= 64 and self.batch is not None:
            self.recent = list(self.batch.get_timing_samples())
            self.batch.set_timing(False)
        return i - i0, el, i


class Samples(list):
    """per-launch kernel durations (ms) of the launches just before a timed region + the region's own average launch duration"""
    region_avg_ms = None
    region_launches = 0


def stats(samples):
    """kernel_avg_ms = the timed region's (t_last_end - t_first_start) / K from ONE HIP event pair on the launch stream;
    median / min / max = per-dispatch HIP events on the launches right before the region (warm-up, end of the prelude)."""
    region = getattr(samples, 'region_avg_ms', None)
    if not samples and not region:
        return {'kernel_avg_ms': 0.0, 'kernel_median_ms': 0.0, 'kernel_min_ms': 0.0, 'kernel_max_ms': 0.0, 'launches_timed': 0}
    s = sorted(samples) or [region]
    return {'kernel_avg_ms': region if region else sum(s) / len(s), 'kernel_median_ms': s[len(s) // 2], 'kernel_min_ms': s[0],
            'kernel_max_ms': s[-1], 'launches_timed': getattr(samples, 'region_launches', 0) or len(s),
            'launches_sampled_individually': len(samples),
            'kernel_avg_source': '(per-dispatch HIP event pair on timed launch 1 + one HIP event pair spanning timed launches 2..K on the launch stream) / K, inter-launch gaps included'
                                 if region else 'per-dispatch HIP events'}


def shader_mhz(batch, step, n, dev):
    """Effective shader clock of one extra (untimed) launch: in-kernel shader-clock counter / 100 MHz counter (np_f16_set_trace)."""
    import ctypes as C
    import torch
    from neuralplane_amd import _lib
    cap = (n + 63) // 64
    trace = torch.zeros((cap, 6), dtype=torch.int64, device=dev)
    _lib.check(batch.lib.np_f16_set_trace(batch._ctx, C.c_void_p(trace.data_ptr()), cap))
    step(0)
    torch.cuda.synchronize(dev)
    _lib.check(batch.lib.np_f16_set_trace(batch._ctx, None, 0))
    t = trace[trace[:, 3] != 0]
    if t.shape[0] == 0:
        return None
    mhz = (t[:, 2] - t[:, 0]).double() / (t[:, 4] - t[:, 3]).clamp(min=1).double() * 100.0
    return float(mhz.median().item())


def side_mode(make_env, make_actions, dev, steps, warmup, prelude_s):
    """A mode reported beside the headline: same protocol (prelude, warm-up, K steps between synchronisations), one GPU."""
    import torch
    env = make_env()
    acts = make_actions()
    env.reset()
    tm = Timer(lambda i: env.step(acts[i % len(acts)]), env._batch, dev, None, 'nccl')
    _, _, i = tm.prelude(prelude_s)
    el, samples, _ = tm.window(warmup, steps, i)
    del env
    torch.cuda.empty_cache()
    return el, samples


def group_fields(dist, args, world):
    """world_size / backend / rccl_ranks as the LIVE process group reports them (dist.get_world_size(), dist.get_backend()), next to
    what was asked for; a mismatch is an error, not a footnote."""
    from neuralplane_amd import sharding
    live_world, live_backend = sharding.live_group(dist)
    if live_world != world or (world > 1 and live_backend != args.backend):
        raise SystemExit(f'process group mismatch: asked for {world} rank(s) on {args.backend}, the live group has {live_world} on {live_backend}')
    return {'world_size': live_world, 'backend': live_backend, 'rccl_ranks': live_world if live_backend == 'nccl' else 0,
            'group_source': 'torch.distributed.get_world_size() / get_backend() of the initialised group' if live_backend else 'single process, no group'}


def expected_weak_scaling(host_us, step_ms, kernel_ms, world):
    """What the 1 -> 8 GPU curve of a weak-scaling config (Heading / Control / Tracking: rows shard, no data-path collective) should
    look like, from THIS run's numbers: every rank enqueues its own launches from its own process, so the aggregate is N x the
    single-GPU rate unless the host cannot keep its GPU fed (host enqueue time per step >= kernel time) — plus the skew of the closing
    barrier, one step at worst over the K timed ones."""
    if not kernel_ms:
        return None
    fed = host_us * 1e-3 < kernel_ms
    per_gpu_ms = max(step_ms, host_us * 1e-3)
    return {'host_enqueue_us_per_step_rank0': host_us, 'kernel_ms': kernel_ms, 'ms_per_step': step_ms, 'host_keeps_gpu_fed': bool(fed),
            'measured_at_world_size': world,
            'predicted_efficiency_8_gpus': [0.95, 1.0] if fed else [kernel_ms / per_gpu_ms * 0.95, kernel_ms / per_gpu_ms],
            'predicted_speedup_8_gpus': [7.6, 8.0] if fed else [8 * kernel_ms / per_gpu_ms * 0.95, 8 * kernel_ms / per_gpu_ms],
            'note': 'weak scaling, one process per GPU, no collective inside the timed steps (RCCL: the two barriers and one MAX all-reduce '
                    'around them); the 5 % allowance is barrier skew and the clock spread between GPUs of a node (+-3-4 % between '
                    'boxes of this pool, DESIGN.md).  A prediction, unmeasured until the driver has an 8-GPU node.'}


def run_env(args, rank, local_rank, world, dev, dist):
    import torch
    from neuralplane_amd import sharding
    from neuralplane_amd.envs.control_env import ControlEnv
    n = args.n  # weak scaling: every GPU simulates args.n aircraft, global rows [rank*n, (rank+1)*n)
    row0, n_local = sharding.shard_rows(world * n, world, rank)
    assert n_local == n
    env = ControlEnv(num_envs=n, config=args.task, model='F16', random_seed=0, device=str(dev), row0=row0,
                     aero_1d_tables=args.aero_1d_tables, solver=args.solver)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    if args.actions == 'random':
        pool = [torch.rand((n, 4), generator=g, device=dev) * 2 - 1 for _ in range(8)]
    else:  # the reference benchmark's clamped constant action (measure_env.py:12-16,68-72)
        pool = [torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).repeat(n, 1)]
    b = env._batch
    tm = Timer(lambda i: env.step(pool[i % len(pool)]), b, dev, dist, args.backend)

    env.reset()
    cold_el, cold_samples, i = tm.window(args.warmup, args.steps)            # 1. the requested protocol from an idle GPU
    p_steps, p_sec, i = tm.prelude(args.prelude_ms * 1e-3, i, est_step_s=cold_el / max(1, args.steps))   # 2. clock-governor ramp, untimed, reported
    elapsed, samples, i = tm.window(args.warmup, args.steps, i)              # 3. W warm-up + EXACTLY K timed steps -> value
    mhz = shader_mhz(b, lambda k: env.step(pool[0]), n, dev) if rank == 0 else None
    mem_mb = torch.cuda.max_memory_allocated(dev) / 2 ** 20   # env state + cache + outputs of one step + the 8-entry action pool
    fin = bool(torch.isfinite(env.model.s).all().item())       # sanity of the timed region: states finite
    if rank != 0:
        return None
    st, cst = stats(samples), stats(cold_samples)
    kern_s = st['kernel_avg_ms'] * 1e-3
    flop = ALGO_FLOP_RK4 if args.solver == 'rk4' else ALGO_FLOP
    ach_tflops = n * flop / kern_s / 1e12 if kern_s > 0 else 0.0
    exe_tflops = n * EXEC_FLOP / kern_s / 1e12 if kern_s > 0 and args.solver != 'rk4' else None
    ach_gbs = n * ALGO_BYTES / kern_s / 1e9 if kern_s > 0 else 0.0
    traffic, traffic_src = pmc_traffic(n, args.task)
    value = world * n * args.steps / elapsed
    out = {
        'metric': METRIC if args.task == 'heading' else f'aircraft-steps/sec, F-16 {args.task}',
        'value': value, 'unit': 'aircraft-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'F-16 {args.task}, N={n} aircraft per GPU, {args.solver or "euler"} dt=0.02, noise_scale per YAML, '
                               f'{args.actions} actions, one fused HIP kernel per env.step',
                   'aircraft_per_gpu': n, 'sharding': f'rows split over {world} GPU(s), no data-path collective',
                   'cross_step_coefficient_reuse': True, 'aero_1d_tables': bool(b.aero_1d_tables)},
        **group_fields(dist, args, world),
        'per_rank': {'host_enqueue_us_per_step': tm.rank_host_us_per_step, 'elapsed_ms_per_step': [1e3 * v / args.steps for v in tm.rank_elapsed_s],
                     'note': 'timed region only; elapsed = each rank\'s own wall time between the two barriers (value uses the max)'},
        'expected_scaling': expected_weak_scaling(tm.rank_host_us_per_step[0], 1e3 * elapsed / args.steps, st['kernel_avg_ms'], world),
        'prelude': {'steps': p_steps, 'seconds': p_sec, 'timed': False,
                    'why': 'clock-governor ramp: the shader clock reaches its steady state only after ~50-100 ms of load '
                           '(cold_start below is the same K-step window taken from an idle GPU)'},
        'cold_start': {'value': world * n * args.steps / cold_el, 'ms_per_step': 1e3 * cold_el / args.steps, **cst,
                       'note': 'reset + W warm-up + K timed steps from an idle GPU, before the prelude; not the headline'},
        'roofline': {'bound': 'valu', 'achieved': ach_tflops, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': ach_tflops / PEAK_FP32_TFLOPS,
                     'traffic': traffic, 'traffic_source': traffic_src,
                     'traffic_unit': 'HBM bytes per launch: rocprofv3 PMC passes FETCH_SIZE x 2.0 + WRITE_SIZE x 1.0 (factors calibrated on this '
                                     'access pattern, profiles/r03_counter_calibration.json) read from the committed profile named in '
                                     'traffic_source (not measured by this run)',
                     'algorithmic_bytes_per_launch': n * ALGO_BYTES,
                     'algorithmic_flop_per_aircraft_step': flop,
                     'executed_flop_per_aircraft_step': EXEC_FLOP if exe_tflops is not None else None,
                     'executed': exe_tflops, 'executed_frac': exe_tflops / PEAK_FP32_TFLOPS if exe_tflops is not None else None,
                     'kernel': 'f16_env_kernel<task,solver,STEP>', **st, 'effective_shader_mhz': mhz,
                     # the clock the governor settles at differs between boxes of the pool (2.15 - 2.37 GHz seen in round 6) while the kernel's CYCLES
                     # per launch do not (6.3 - 6.5e5): the same fraction against the peak AT THE MEASURED CLOCK (157.3 TFLOP/s is quoted at 2.4 GHz)
                     'kernel_shader_kcycles': st['kernel_avg_ms'] * mhz if mhz else None,
                     'frac_at_measured_clock': (ach_tflops / (PEAK_FP32_TFLOPS * mhz / 2400.0)) if mhz else None,
                     'note': 'fp32 VECTOR (VALU) roof: 157.3 TFLOP/s is both the fp32 vector peak and the f32-input MFMA peak on '
                             'gfx950; the kernel issues no MFMA.  achieved = algorithmic FLOP x N / average launch duration (HIP '
                             'events on the launch stream: one pair around the K timed launches; per-dispatch pairs on the launches before them for the median); executed = the FLOP the kernel '
                             'really performs (cross-step coefficient reuse)'},
        'roofline_hbm': {'bound': 'hbm', 'achieved': ach_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                         'frac': ach_gbs / PEAK_HBM_GBS, 'note': '278 algorithmic B per aircraft-step; not the binding roof'},
        'state_finite': fin,
        # the reference publishes 245.5 MB allocated after its N = 1e6 run (envs/measure_env/gpu_memory_neuralplane.npy)
        'device_memory_mb': mem_mb,
    }
    if world > 1 or args.headline_only or not args.full:
        return out
    del env, tm
    torch.cuda.empty_cache()
    ps = min(args.prelude_ms, 150) * 1e-3
    rand_pool = lambda m: (lambda: [torch.rand((m, 4), generator=g, device=dev) * 2 - 1 for _ in range(4)])  # noqa: E731
    modes = out['optional_modes'] = {}
    if not b.aero_1d_tables and args.solver != 'rk4':
        # optional numerics mode (never `value`): the 22 single-input aero nets through their exact piecewise-linear tables
        k2 = min(args.steps, 100)
        el2, s2 = side_mode(lambda: ControlEnv(num_envs=n, config=args.task, model='F16', random_seed=0, device=str(dev), row0=row0,
                                               aero_1d_tables=1), lambda: pool, dev, k2, 5, ps)
        ms2 = stats(s2)['kernel_avg_ms']
        # executed per aircraft-step in this mode: the multi-input nets of the cached integrator evaluation (14 nets, 9 350 FLOP) and of the
        # Overload re-evaluation (9 nets, 5 910), 22 + 14 table lookups (one fma each behind a 6-step search) and ~900 of non-MLP arithmetic
        exec_tab = 9350 + 5910 + 2 * 36 + 900
        modes['aero_1d_tables'] = {'value': n * k2 / el2, 'unit': 'aircraft-steps/s', 'steps': k2, **stats(s2),
                                   'executed_flop_per_aircraft_step': exec_tab,
                                   'executed_tflops': n * exec_tab / (ms2 * 1e-3) / 1e12 if ms2 > 0 else 0.0,
                                   'executed_frac_of_fp32_peak': n * exec_tab / (ms2 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if ms2 > 0 else 0.0,
                                   'kernel': 'pair variant (round 3): the 20 multi-input nets on the two-set bodies, the 22 single-input nets as per-lane table lookups',
                                   'note': 'not the headline: changes the rounding of 22 of the 42 aero coefficients by ~1e-5 rel '
                                           '(tests: masks identical to the reference, HIP == oracle bit-exact)'}
    if args.solver != 'rk4':
        # the other integrator of the reference (`solver: rk4`, torchdiffeq's 3/8 rule: 4 aero evaluations per step)
        k4 = min(args.steps, 50)
        el4, s4 = side_mode(lambda: ControlEnv(num_envs=n, config=args.task, model='F16', random_seed=0, device=str(dev), row0=row0,
                                               solver='rk4'), lambda: pool, dev, k4, 3, ps)
        ms4 = stats(s4)['kernel_avg_ms']
        modes['solver_rk4'] = {'value': n * k4 / el4, 'unit': 'aircraft-steps/s', 'steps': k4, **stats(s4),
                               'algorithmic_tflops': n * ALGO_FLOP_RK4 / (ms4 * 1e-3) / 1e12 if ms4 > 0 else 0.0,
                               'note': '4 x 23.8 KFLOP aero evaluations + the Overload re-evaluation = 105 KFLOP per aircraft-step '
                                       '(SURVEY 8d); reference parity of rk4 is unpinned (no artefact exercises it), HIP == oracle bit-exact'}
    if n == 1_000_000:
        # the same kernel on a batch that amortises launch, start-up and the tail of the grid: the kernel's asymptotic rate
        nb, k5 = 10_000_000, 20
        el5, s5 = side_mode(lambda: ControlEnv(num_envs=nb, config=args.task, model='F16', random_seed=0, device=str(dev)),
                            lambda: [torch.rand((nb, 4), generator=g, device=dev) * 2 - 1], dev, k5, 2, ps)
        ms5 = stats(s5)['kernel_avg_ms']
        modes['batch_1e7'] = {'value': nb * k5 / el5, 'unit': 'aircraft-steps/s', 'steps': k5, **stats(s5),
                              'fp32_roof_frac': nb * ALGO_FLOP / (ms5 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if ms5 > 0 else 0.0,
                              'note': 'N = 1e7 aircraft on one GPU (3 GB of state + observations of the 288 GB): same kernel, same numerics'}
    # BASELINE.json configs[0] size (N = 256 = 4 waves): the latency regime — microseconds per env.step, not a roofline.  Wall time
    # is taken WITHOUT the per-launch HIP events (two event records between consecutive 20 us kernels are a measurable part of the
    # gap); the kernel duration comes from a second window with them.
    envs = ControlEnv(num_envs=256, config=args.task, model='F16', random_seed=0, device=str(dev))
    a_s = [torch.rand((256, 4), generator=g, device=dev) * 2 - 1 for _ in range(4)]
    envs.reset()
    for i in range(200):
        envs.step(a_s[i % 4])
    torch.cuda.synchronize(dev)
    k6 = 2000
    t1 = time.perf_counter()
    for i in range(k6):
        envs.step(a_s[i % 4])
    torch.cuda.synchronize(dev)
    el6 = time.perf_counter() - t1
    tm6 = Timer(lambda i: envs.step(a_s[i % 4]), envs._batch, dev, None, 'nccl', per_launch=True)   # launch-bound: per-dispatch events
    _, s6, _ = tm6.window(20, 500)
    modes['latency_n256'] = {'value': 1e6 * el6 / k6, 'unit': 'us per env.step (wall, back-to-back launches, no timing events)', 'steps': k6,
                             'kernel_avg_us': 1e3 * stats(s6)['kernel_avg_ms'], 'kernel_median_us': 1e3 * stats(s6)['kernel_median_ms'],
                             'aircraft_steps_per_s': 256 * k6 / el6,
                             'note': 'latency8 variant: eight waves share a tile of 64 aircraft (two per SIMD), split the net evaluations, the serial fp64 '
                                     'chains and the observation noise; the GPU is otherwise idle (reference training sizes: 3 000-10 000 '
                                     'envs, scripts/train_heading.sh)'}
    del envs, tm6
    modes['singlecombat_1v1'] = combat_mode(dev, 100_000, min(args.steps, 100), 5, ps)
    modes['planning_tracking_n8192'] = planning_mode(dev, g, 8_192, 20)
    modes['planning_tracking_n1e4'] = planning_mode(dev, g, 10_000, 20)
    modes['planning_tracking_n262144'] = planning_mode(dev, g, 262_144, 4)
    # SURVEY 8(f) N1 end to end: the reference's collect loop (policy forward -> env.step -> buffer insert -> compute_returns) at its own
    # training sizes, device-resident / with a HIP graph / through the numpy contract (tools/collect_loop.py)
    from tools.collect_loop import collect_loop_report
    modes['collect_loop_n3000'] = collect_loop_report(3_000, 200, dev)
    modes['collect_loop_n1e4'] = collect_loop_report(10_000, 200, dev)
    # last: it wants an idle GPU
    modes['reference_protocol'] = reference_protocol_mode(dev)
    return out


# the reference's published totals for 500 env.steps (envs/measure_env/time_neuralplane.npy, producer envs/measure_env.py:65-78)
REF_PUBLISHED_S_500 = {10_000: 18.25, 100_000: 21.13, 1_000_000: 105.16}


def reference_protocol_mode(dev):
    """The paper's benchmark as the paper ran it (/root/reference/envs/measure_env.py:65-78,114,129): a fresh ControlEnv('heading', F16,
    seed 0), the constant action its script builds (INIT_U's controls, which env.step clamps to (1, 0, 0, 0)), 500 env.steps, NO warm-up,
    no clock prelude, started after two seconds of idle GPU; time.time() around the loop — plus one device synchronisation at the end,
    which the reference's script lacks (its loop returns when the last step is enqueued; ours would too, 30 x earlier)."""
    import torch
    from neuralplane_amd.envs.control_env import ControlEnv
    out = {'protocol': 'N aircraft, F-16 Heading, constant clamped action (1, 0, 0, 0), 500 steps, no warm-up, no prelude, idle GPU before; '
                       'seconds for the 500 steps incl. a final device synchronisation', 'unit': 's per 500 env.steps'}
    for n in (1_000_000, 100_000, 10_000):
        env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device=str(dev))
        a = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).repeat(n, 1)
        torch.cuda.synchronize(dev)
        time.sleep(2.0)                       # let the clock governor fall back: the reference's run starts on an idle GPU
        t0 = time.time()
        for _ in range(500):
            env.step(a)
        torch.cuda.synchronize(dev)
        el = time.time() - t0
        out[f'n{n}'] = {'seconds_500_steps': el, 'reference_published_seconds_500_steps': REF_PUBLISHED_S_500[n],
                        'aircraft_steps_per_s': n * 500 / el, 'ratio_to_published': REF_PUBLISHED_S_500[n] / el}
        del env
    out['note'] = ('the published column is the reference on its authors\' unnamed NVIDIA GPU (BASELINE.md section 1): other hardware, quoted for '
                   'orientation; at N <= 1e5 this run is host-bound (one launch per step, ~16 us of Python per env.step)')
    return out


def combat_mode(dev, E, steps, warmup, prelude_s):
    """BASELINE.json configs[4] on one GPU (reported beside the headline): SingleCombat 1v1, E engagements = 2E aircraft, one
    launch per env.step = 5 FDM steps behind the attitude PID stack."""
    import torch
    from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
    cenv = SingleCombatEnv(num_envs=E, config='selfplay', random_seed=0, device=str(dev))
    cenv.reset()
    g2 = torch.Generator(device='cpu').manual_seed(5)
    cpool = [(torch.rand((2 * E, 4), generator=g2) * 2 - 1).to(dev) for _ in range(4)]
    tm = Timer(lambda i: cenv.step(cpool[i % 4]), cenv._batch, dev, None, 'nccl', per_launch=True)
    _, _, i = tm.prelude(prelude_s)
    el, samples, _ = tm.window(warmup, steps, i)
    st = stats(samples)
    ks = st['kernel_avg_ms'] * 1e-3
    ach = 2 * E * 5 * ALGO_FLOP_COMBAT_FDM / ks / 1e12 if ks > 0 else 0.0
    out = {'value': E * steps / el, 'unit': 'engagement-steps/s', 'aircraft_fdm_steps_per_s': 2 * E * 5 * steps / el, 'steps': steps,
           'engagements': E, **st,
           'roofline': {'bound': 'valu', 'achieved': ach, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_FP32_TFLOPS,
                        'algorithmic_flop_per_engagement_step': 2 * 5 * ALGO_FLOP_COMBAT_FDM,
                        'hbm_gbs': 2 * E * ALGO_BYTES_COMBAT / ks / 1e9 if ks > 0 else 0.0,
                        'note': '2 aircraft x 5 FDM steps x 34.1 KFLOP (the two aero evaluations of an FDM step + ~300 FLOP of PID '
                                'stack, terminations and pairwise geometry) per engagement-step; ~0.5 KB of HBM per aircraft and env.step'},
           'note': 'one f16_combat_kernel launch per SingleCombatEnv.step (pairwise reset, 5 x {PID stack, FDM step, '
                   'terminations}, pairwise obs/reward/blood); HIP == oracle bit-exact (tests/test_gpu_combat_parity.py)'}
    del cenv
    torch.cuda.empty_cache()
    return out


def planning_mode(dev, g, npl, k7, numerics='i8'):
    """BASELINE.json configs[3] as the reference ships it (hierarchical Tracking): PlanningEnv.step = 50 x {frozen PPOActor-
    architecture controller as ONE fused MFMA kernel, fused env step that also writes the controller's next observation}, at the
    batch size of the reference's own training script (n = 1e4) and at a size that fills the GPU (262 144); random-init controller
    weights of that architecture."""
    import numpy as np
    import torch
    from neuralplane_amd.actor import FusedActor, NUM_FLOATS
    from neuralplane_amd.envs.planning_env import PlanningEnv
    ctrl = FusedActor(np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32), str(dev), numerics=numerics)
    penv = PlanningEnv(num_envs=npl, config='tracking', model='F16', random_seed=0, device=str(dev), controller=ctrl)
    ap = torch.rand((npl, 3), generator=g, device=dev) * 2 - 1
    for _ in range(3):
        penv.step(ap)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(k7):
        penv.step(ap)
    torch.cuda.synchronize(dev)
    el7 = time.perf_counter() - t1
    ms = 1e3 * el7 / k7
    # the same macro-step through the round-3 path (2 x 50 launches, row groups on their own streams), and the env kernels' share of it
    penv.loop_mode = 'launches'
    for _ in range(2):
        penv.step(ap)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(k7):
        penv.step(ap)
    torch.cuda.synchronize(dev)
    ms_launches = 1e3 * (time.perf_counter() - t1) / k7
    penv._batch.set_timing(True)
    penv.step(ap)
    torch.cuda.synchronize(dev)
    env_ms = sum(penv._batch.get_timing_samples())
    penv._batch.set_timing(False)
    penv.loop_mode = 'auto'
    tiles, cus = (npl + 31) // 32, torch.cuda.get_device_properties(dev).multi_processor_count
    auto = ('persistent kernel, eight waves per 32-row tile, one workgroup per tile' if tiles <= cus else
            'persistent kernel, guest schedule (every CU owns a tile and hosts one block of iterations of a guest tile)' if tiles - cus <= cus // 2 else
            'persistent kernel, dual workgroups (two tiles per eight-wave workgroup, one FDM step for both)' if tiles <= 2 * cus else
            'launch by launch (np_actor_forward + np_f16_step per iteration, row groups on their own streams)')
    flop_actor, flop_env = 2 * 151_000.0, ALGO_FLOP     # PPOActor.forward: 151 K multiply-adds per aircraft and call; one FDM step
    per_macro = 50 * (flop_actor + flop_env)
    ach = npl * per_macro / (ms * 1e-3) / 1e12
    i8 = numerics == 'i8'
    # the integer work of the block-fixed-point controller: nine limb products per multiply-add of the six quantised layers (148 K of the 151 K)
    i8_ops = 2 * 9 * 148_000.0 * 50
    out = {'value': ms, 'unit': 'ms per PlanningEnv.step (50 inner FDM steps + 50 controller calls)', 'steps': k7, 'aircraft': npl,
           'controller_numerics': 'block fixed point on the i8 matrix pipe (np_actor_i8.h; the CPU restatement f16_actor_i8.inc)' if i8 else 'fp32 fmaf chains (np_actor.h)',
           'aircraft_fdm_steps_per_s': npl * 50 * k7 / el7, 'inner_loop': auto if not (i8 and 'dual' in auto) else 'launch by launch (the dual workgroups serve the fp32 controller only)',
           'launches_per_macro_step': 3 if tiles <= 2 * cus and not (i8 and 'dual' in auto) else 2 + 50 * 2,
           'launch_by_launch': {'ms': ms_launches, 'env_kernels_summed_ms': env_ms,
                                'note': 'NP_PLANNING_MODE=launches: the round-3 path, 102 launches; env_kernels_summed_ms = sum of the 50 inner-step kernel '
                                        'durations (kernels of different row groups overlap: a sum, not wall time)'},
           'roofline': {'bound': 'valu + latency: with the controller\'s matrix work on the i8 pipe (15 % of ITS peak) what binds is the fp32 vector work around it — '
                                 'epilogues, LayerNorms, gates, the FDM step — and the dependent-issue latency of one wave per SIMD; the fp32 pipe is the roof quoted'
                        if i8 else 'mfma+valu (fp32: the f32-input MFMA and the vector ALU share one 157.3 TFLOP/s pipe, tools/microbench/mfma_coissue.hip)',
                        'achieved': ach, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s (fp32-EQUIVALENT: the algorithmic FLOP of the fp32 formulation / time)', 'frac': ach / PEAK_FP32_TFLOPS,
                        'algorithmic_flop_per_aircraft_macro_step': per_macro,
                        'i8_matrix_pipe': ({'achieved_tops': npl * i8_ops / (ms * 1e-3) / 1e12, 'peak_tops': 5000.0, 'frac': npl * i8_ops / (ms * 1e-3) / 1e12 / 5000.0,
                                            'busy_fraction_pmc': planning_matrix_busy(ms) if npl == 8_192 else None,
                                            'note': 'integer operations of the nine limb products / time against the dense i8 peak; busy_fraction_pmc = SQ_VALU_MFMA_BUSY_CYCLES '
                                                    'per SIMD / the launch\'s shader cycles, from the committed rocprofv3 PMC pass (profiles/, n = 8 192), not measured by this run'}
                                           if i8 else None),
                        'note': '50 x (302 KFLOP controller forward + 33.8 KFLOP FDM step) = 16.8 MFLOP per aircraft and PlanningEnv.step; wall clock '
                                'of back-to-back macro-steps (reset + prelude launches included)'},
           'note': 'n <= 32 x CUs: ONE launch per macro-step of the persistent kernel (np_planning.hip) — a workgroup owns a 32-row tile and loops the 50 x '
                   '{controller call, inner FDM step} with the recurrent state in registers and the tile\'s observation / state / cached coefficients in LDS; '
                   "eight waves per tile: an inner step's Overload evaluation, terminations and reward run on waves 4..7 during the NEXT controller call.  "
                   'The controller call (round 5): the six Linear layers with N >= 128 as v_mfma_i32_32x32x32_i8 on integer limbs (activations sign + 22 bits per row, '
                   'weights sign + 29 bits per output feature, nine limb products in four exact class sums), every activation in the accumulator layout from the '
                   'first layer to the head, LayerNorm + quantiser fused, weight fragments prefetched one M-block ahead, 15 barriers (fp32 formulation: 1 184 dependent '
                   'K = 1 MFMA steps, 23 barriers).  Up to 1.5 tiles per CU (n = 1e4: 313 tiles on 256 CUs) the guest schedule: every CU owns a tile and hosts one '
                   'block of ~13 iterations of a guest tile between two stretches of its own (makespan 63 iterations instead of 100; tiles change CU through coherent sc1 '
                   'accesses; bounded waits).  Larger batches: launch by launch, two to four row groups on their own streams.  Bit-identical between all schedules and to '
                   'the integer CPU restatement (tests/test_gpu_actor.py)'}
    if i8 and npl <= 16_384:   # the fp32 controller on the same schedule, for the comparison
        del penv, ctrl
        torch.cuda.empty_cache()
        ctrl = FusedActor(np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32), str(dev), numerics='fp32')
        penv = PlanningEnv(num_envs=npl, config='tracking', model='F16', random_seed=0, device=str(dev), controller=ctrl)
        for _ in range(3):
            penv.step(ap)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(k7):
            penv.step(ap)
        torch.cuda.synchronize(dev)
        out['fp32_controller_ms'] = 1e3 * (time.perf_counter() - t1) / k7
    del penv, ctrl
    torch.cuda.empty_cache()
    return out


def planning_matrix_busy(ms):
    """SQ_VALU_MFMA_BUSY_CYCLES of the persistent kernel (block-fixed-point controller, n = 8 192) from the newest committed PMC summary,
    as a fraction of the launch: summed over the chip's 1 024 SIMDs / (1 024 x the launch's shader cycles at 2.39 GHz)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_planning_pmc.csv')))
    if not files:
        return None
    for r in csv.reader(open(files[-1])):
        if len(r) >= 4 and r[0] == 'i8' and r[1] == 'SQ_VALU_MFMA_BUSY_CYCLES':
            return {'value': float(r[3]) / 1024.0 / (ms * 1e-3 * 2.39e9), 'source': os.path.basename(files[-1])}
    return None


def _wall(step, batch, dev, warmup, k):
    """seconds per step, back-to-back launches, no per-dispatch timing events (each pair stretches a dispatch by ~4.5 us)"""
    import torch
    batch.set_timing(False)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(k):
        step(i)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / k


def _kernel_ms(step, batch, dev, k):
    """average duration of the context's kernel over k steps (events attached to each dispatch)"""
    import torch
    batch.set_timing(True)
    for i in range(k):
        step(i)
    torch.cuda.synchronize(dev)
    smp = batch.get_timing_samples()
    batch.set_timing(False)
    return sum(smp) / max(1, len(smp))


def run_combat(args, rank, local_rank, world, dev, dist):
    """BASELINE.json configs[4]: SingleCombat 1v1 self-play, `--engagements` in total sharded BY ENV over the ranks (strong
    scaling), with the opponent-observation exchange of the self-play runner inside the stepped loop
    (neuralplane_amd/selfplay.py; reference runner/selfplay_F16sim_runner.py:49-55,90-100)."""
    import torch
    from neuralplane_amd import sharding
    from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
    from neuralplane_amd.selfplay import OpponentExchange
    e_total = args.engagements
    env0, e_loc = sharding.shard_rows(e_total, world, rank)
    cenv = SingleCombatEnv(num_envs=e_loc, config='selfplay', random_seed=0, device=str(dev), env0=env0)
    # stand-ins for the two policies (the caller's networks are not on this path): fixed linear maps of the 15-float observation
    W_ego = torch.linspace(-1, 1, 15 * 4, device=dev).reshape(15, 4)
    W_opp = torch.linspace(1, -1, 15 * 4, device=dev).reshape(15, 4)
    ego_policy = lambda o: torch.tanh(o @ W_ego)                # noqa: E731
    opp_policy = lambda o, ids: torch.tanh(o @ W_opp)           # noqa: E731
    ex = OpponentExchange(e_loc, env0, e_total, dist, dev, opponent_policy=opp_policy, lag=args.opponent_lag)
    # the split layout (np_f16_combat_io.action_opp / obs_opp): ego / opponent halves are separate contiguous arrays the kernel reads
    # and writes itself, so the exchange and the policies work on the kernel's buffers — no split / stack / contiguous copies
    state = {'obs': cenv.reset_split()} if not args.interleaved else {'obs': cenv.reset()}

    def step(i):
        if args.interleaved:   # round-2 loop, kept for A/B: interleaved [2E, .] rows, torch split / contiguous / stack around the launch
            a = ex.actions(state['obs'], ego_policy)
            state['obs'] = cenv.step(a)[0]
        else:
            ea, oa = ex.actions_split(state['obs'][0], state['obs'][1], ego_policy)
            state['obs'] = cenv.step_split(ea, oa)[:2]

    tm = Timer(step, cenv._batch, dev, dist, args.backend, per_launch=True)
    cold_el, cold_samples, i = tm.window(args.warmup, args.steps)
    p_steps, p_sec, i = tm.prelude(args.prelude_ms * 1e-3, i, est_step_s=cold_el / max(1, args.steps))   # same count on every rank: steps hold collectives
    elapsed, samples, i = tm.window(args.warmup, args.steps, i)
    fin = bool(torch.isfinite(cenv.s).all().item())
    plain_ms = 1e3 * _wall(step, cenv._batch, dev, 10, max(50, args.steps)) if world == 1 else None   # the loop without timing events
    policy_ms = None
    if world == 1 and not args.interleaved:   # the two stand-in policies alone (4 small torch kernels): what the loop spends outside this library
        oe, oo = state['obs']
        ids = torch.arange(e_loc, device=dev)
        policy_ms = 1e3 * _wall(lambda i: (ego_policy(oe), opp_policy(oo, ids)), cenv._batch, dev, 10, 200)
    # What the 1 -> 8 GPU curve of this config should look like, from numbers measured in THIS run (single-GPU runs only): the
    # per-GPU share of an 8-rank job stepped through the same loop on this GPU (no collective in it), plus an estimate for the two
    # all-gathers of a step.  Strong scaling: the work per GPU shrinks 8x, the per-step floor (kernel latency at a small grid,
    # policy launches, exchange) does not.
    expected = None
    if world == 1 and not args.headline_only and args.full:
        share = max(64, e_total // 8)
        senv = SingleCombatEnv(num_envs=share, config='selfplay', random_seed=0, device=str(dev), env0=0)
        sex = OpponentExchange(share, 0, share, None, dev, opponent_policy=opp_policy, lag=args.opponent_lag)
        sst = {'obs': senv.reset_split()}

        def sstep(i):
            ea, oa = sex.actions_split(sst['obs'][0], sst['obs'][1], ego_policy)
            sst['obs'] = senv.step_split(ea, oa)[:2]
        s_step_ms = 1e3 * _wall(sstep, senv._batch, dev, 30, 300)          # no timing events in this window
        s_kernel_ms = _kernel_ms(sstep, senv._batch, dev, 100)
        ag_lo, ag_hi = 0.04, 0.06   # ms per all-gather on xGMI at this payload (0.75 MB / 0.2 MB per rank): launch + ring latency bound, DESIGN.md §7 — an estimate, no multi-GPU box was available
        t1 = plain_ms
        lag0 = (s_step_ms + 2 * ag_lo, s_step_ms + 2 * ag_hi)       # both collectives exposed behind the ego policy's two small kernels
        lag1 = (max(s_step_ms, 2 * ag_lo), max(s_step_ms, 2 * ag_hi))  # the exchange overlaps the env kernel of the previous step
        expected = {'per_gpu_share_engagements': share, 'kernel_ms_at_share': s_kernel_ms, 'loop_ms_at_share_one_gpu': s_step_ms,
                    'loop_overhead_ms_at_share': s_step_ms - s_kernel_ms, 'allgather_ms_each_estimate': [ag_lo, ag_hi],
                    'ms_per_step_one_gpu_full_size': t1,
                    'predicted_ms_per_step_8_gpus_lag0': list(lag0), 'predicted_speedup_8_gpus_lag0': [t1 / lag0[1], t1 / lag0[0]],
                    'predicted_ms_per_step_8_gpus_lag1': list(lag1), 'predicted_speedup_8_gpus_lag1': [t1 / lag1[1], t1 / lag1[0]],
                    'note': 'strong scaling of a latency-bound step: the kernel at the per-GPU share is a one-generation grid (no 8x from 8x '
                            'fewer rows), so >= 6x is not expected for this config; Heading / Tracking (weak scaling, no collective) are'}
        del senv, sex
    if rank != 0:
        return None
    st = stats(samples)
    ks = st['kernel_avg_ms'] * 1e-3
    ach = 2 * e_loc * 5 * ALGO_FLOP_COMBAT_FDM / ks / 1e12 if ks > 0 else 0.0
    return {
        'metric': 'engagement-steps/sec, SingleCombat 1v1 self-play (BASELINE.json configs[4])',
        'value': e_total * args.steps / elapsed, 'unit': 'engagement-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'SingleCombat 1v1, {e_total} engagements in total ({e_loc} on rank 0), 5 FDM steps per env.step, fixed '
                               f'linear stand-in policies, opponent exchange: 2 all-gathers per step on a side stream, opponent lag {args.opponent_lag}',
                   'engagements_total': e_total, 'sharding': f'envs split over {world} GPU(s); all-gather of opponent observations / actions only'},
        'aircraft_fdm_steps_per_s': 2 * e_total * 5 * args.steps / elapsed,
        **group_fields(dist, args, world),
        'per_rank': {'host_enqueue_us_per_step': tm.rank_host_us_per_step, 'elapsed_ms_per_step': [1e3 * v / args.steps for v in tm.rank_elapsed_s]},
        'exchange': {'collectives_per_step': 2 if world > 1 else 0, 'obs_bytes_gathered_per_step': e_total * 15 * 4, 'action_bytes_gathered_per_step': e_total * 4 * 4,
                     'opponent_lag': args.opponent_lag},
        'expected_scaling': expected,
        'layout': 'interleaved rows + torch split / stack (round-2 loop)' if args.interleaved else 'split ego / opponent arrays (np_f16_combat_io.action_opp / obs_opp): no copies around the launch',
        'ms_per_step_without_timing_events': plain_ms, 'loop_overhead_ms': (plain_ms - st['kernel_avg_ms']) if plain_ms is not None else None,
        'stand_in_policies_ms': policy_ms,
        'exchange_and_glue_ms': (plain_ms - st['kernel_avg_ms'] - policy_ms) if (plain_ms is not None and policy_ms is not None) else None,
        'prelude': {'steps': p_steps, 'seconds': p_sec, 'timed': False},
        'cold_start': {'value': e_total * args.steps / cold_el, 'ms_per_step': 1e3 * cold_el / args.steps, **stats(cold_samples)},
        'roofline': {'bound': 'valu', 'achieved': ach, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_FP32_TFLOPS,
                     'algorithmic_flop_per_engagement_step': 2 * 5 * ALGO_FLOP_COMBAT_FDM, 'kernel': 'f16_combat_kernel<solver,STEP>', **st,
                     'traffic': None, 'note': 'rank 0 kernel; per engagement-step 2 aircraft x 5 FDM steps x 34.1 KFLOP'},
        'state_finite': fin,
    }


DETAILS_PREFIX = 'BENCH_DETAILS '
CONTRACT_MAX_BYTES = 4000


def contract_line(out, details_path=None):
    """The ONE line the driver parses: the contract fields and nothing else (< 4 KB).  Everything else this run measured (per-rank
    times, cold start, prelude, expected scaling, optional modes, the long notes) is on the BENCH_DETAILS line printed before it
    and in gpurun_out/bench_details.json."""
    r = out['roofline']
    keep_r = ('bound', 'achieved', 'peak', 'unit', 'frac', 'executed_frac', 'kernel', 'kernel_avg_ms', 'kernel_median_ms', 'launches_timed',
              'traffic', 'traffic_source', 'algorithmic_bytes_per_launch', 'algorithmic_flop_per_aircraft_step', 'executed_flop_per_aircraft_step',
              'algorithmic_flop_per_engagement_step', 'effective_shader_mhz', 'kernel_shader_kcycles', 'frac_at_measured_clock')
    line = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                'vs_baseline', 'dtype', 'data')}
    line['config'] = {k: v for k, v in out['config'].items() if k in ('workload', 'aircraft_per_gpu', 'engagements_total', 'sharding')}
    line['roofline'] = {k: r[k] for k in keep_r if k in r}
    if 'roofline_hbm' in out:
        line['roofline']['hbm_achieved_gbs'] = out['roofline_hbm']['achieved']
        line['roofline']['hbm_frac'] = out['roofline_hbm']['frac']
    line['roofline']['kernel_avg_source'] = 'HIP events on the launch stream over the K timed launches'
    c = out.get('cpu_baseline')
    if c:
        te = c.get('torch_eager') or {}
        line['cpu_baseline'] = {'value': c['value'], 'unit': c['unit'], 'cores': c['cores'], 'kind': c['kind'], 'sample': c['sample'],
                                'torch_eager': te.get('value', te.get('aircraft_steps_per_s')) if isinstance(te, dict) else te}
    for k in ('world_size', 'backend', 'rccl_ranks', 'state_finite'):
        if k in out:
            line[k] = out[k]
    line['details'] = details_path
    txt = json.dumps(line)
    if len(txt) > CONTRACT_MAX_BYTES:   # never let the contract line grow again (round 5: 28.8 KB, unparsed)
        raise SystemExit(f'bench.py: contract line is {len(txt)} bytes (> {CONTRACT_MAX_BYTES})')
    return txt


def emit(out):
    """BENCH_DETAILS <everything> on an earlier stdout line + gpurun_out/bench_details.json; the contract line LAST."""
    path = None
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, 'bench_details.json')
        with open(path, 'w') as f:
            json.dump(out, f, indent=1)
        path = os.path.relpath(path, ROOT)
    except OSError:
        path = None
    print(DETAILS_PREFIX + json.dumps(out), flush=True)
    print(contract_line(out, path), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--n', '--aircraft', dest='n', type=int, default=1_000_000, help='aircraft per GPU')
    ap.add_argument('--task', default='heading', choices=['heading', 'control', 'tracking', 'combat'])
    ap.add_argument('--solver', default=None, choices=['euler', 'rk4'], help='default: the scenario YAML (euler)')
    ap.add_argument('--actions', default='random', choices=['random', 'constant'])
    ap.add_argument('--interleaved', action='store_true', help='--task combat: the interleaved [2E, .] action / observation rows with torch-side split / stack (round-2 loop)')
    ap.add_argument('--engagements', type=int, default=100_000, help='--task combat: engagements in TOTAL (sharded by env over the ranks)')
    ap.add_argument('--opponent-lag', type=int, default=0, choices=[0, 1],
                    help='--task combat: 0 = the opponent acts on the current observation (the reference runner); 1 = on the previous one, '
                         'so that the exchange overlaps the env kernel')
    ap.add_argument('--prelude-ms', type=float, default=300.0, help='untimed GPU load before the timed window (clock-governor ramp); 0 disables')
    ap.add_argument('--headline-only', '--no-cpu-baseline', dest='headline_only', action='store_true',
                    help='skip cpu_baseline and the optional modes reported beside the headline')
    ap.add_argument('--full', action='store_true',
                    help='also run the modes reported beside the headline (rk4, 1-D tables, N = 1e7, N = 256 latency, SingleCombat, PlanningEnv, the collect '
                         'loop, the reference protocol): minutes, reported on the BENCH_DETAILS line / gpurun_out/bench_details.json, never on the contract line')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend: 'nccl' = RCCL (default); 'gloo' lets the multi-rank path be exercised on a box with "
                         'fewer GPUs than ranks (ranks then share GPUs: a functional check, not a benchmark)')
    ap.add_argument('--aero-1d-tables', type=int, default=None, choices=[0, 1],
                    help='numerics option (DESIGN.md §4); default: scenario / NPF16_AERO_1D_TABLES / off')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    import torch
    from neuralplane_amd import sharding
    rank, local_rank, world = sharding.env_world()
    if world != args.gpus:
        raise SystemExit(f'WORLD_SIZE={world} does not match --gpus {args.gpus}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path')
    ndev = torch.cuda.device_count()
    if args.backend == 'nccl' and local_rank >= ndev:
        raise SystemExit(f'LOCAL_RANK={local_rank} but only {ndev} GPU(s) visible')
    dev = torch.device('cuda', local_rank % ndev)
    torch.cuda.set_device(dev)
    dist = sharding.init_distributed(args.backend, dev)  # RCCL: timing barrier / max (+ the combat exchange)

    if args.task == 'combat':
        out = run_combat(args, rank, local_rank, world, dev, dist)
    else:
        out = run_env(args, rank, local_rank, world, dev, dist)
    if rank == 0:
        if world == 1 and not args.headline_only:
            out['cpu_baseline'] = cpu_baseline('heading' if args.task == 'combat' else args.task)
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
