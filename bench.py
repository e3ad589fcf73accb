#!/usr/bin/env python3
"""bench.py — aircraft-steps/sec of the fused F-16 Heading env.step on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n AIRCRAFT_PER_GPU] [--task heading]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: F-16 Heading, N = 1e6 aircraft per GPU (weak scaling: the
batch shards embarrassingly, no data-path collective), envs/configs/heading.yaml constants
(Euler, dt = 0.02, noise_scale = 0.01), per-step uniform random actions from a fixed seed so that
terminations / auto-resets occur at a realistic rate.  A "step" is one `ControlEnv.step(action)`
= one fused HIP kernel launch; inputs (state, actions) are resident in HBM before the timed region.
Protocol follows the reference's own benchmark (envs/measure_env.py:65-78: back-to-back env.step,
500 steps) with a warm-up and device synchronisation added.

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for the roofline arithmetic).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic work per aircraft-step, F-16 Heading, Euler
ALGO_BYTES = 278.0      # HBM: read s48+u20+tgt12+step_count8+flags3+action16, write s48+u20+step_count8+obs88+reward4+flags3
ALGO_FLOP = 33.8e3      # aero eval 23 770 + force-side re-evaluation 9 160 + ~900 non-MLP (FMA = 2)
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 vector peak == fp32 (f32-input) MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def pmc_traffic(n, task):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json;
    bench.py cannot collect PMC counters itself).  None unless the profile matches this workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json'))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if int(d.get('n', -1)) == int(n) and task == 'heading':
            best = float(d['traffic_bytes_per_launch'])
    return best


def cpu_baseline(task, budget_s=15.0):
    """The CPU oracle (oracle/, the C restatement of the reference path) timed on this box's host
    cores on a bounded sample of the same workload.  A reported baseline, never the thing shipped."""
    import math
    import numpy as np
    # threads = CPUs this process may really use (affinity mask, capped by the cgroup CPU quota): libgomp's
    # default is the machine's CPU count, which oversubscribes a container limited to a few cores
    cpus = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            cpus = max(1, min(cpus, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
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
    return {'value': n * steps / el, 'unit': 'aircraft-steps/s', 'cores': int(o.threads), 'kind': 'port',
            'sample': f'F-16 {task}, N={n} aircraft x {steps} steps, oracle/f16_oracle.c (OpenMP, fp32 scalar, '
                      f'same numerics spec), {el:.1f} s',
            # context only: the reference's own eager-PyTorch path cannot travel to this box (its source never leaves the build
            # container); measured there and published by its authors
            'reference_context': {'pytorch_cpu_8_vcpu_N1e5': 4.7e5, 'published_pytorch_cuda_N1e6': 4.75e6, 'unit': 'aircraft-steps/s'}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--n', '--aircraft', dest='n', type=int, default=1_000_000,
                    help='aircraft per GPU (use --aircraft under torch.distributed.run, whose parser treats --n as an abbreviation)')
    ap.add_argument('--task', default='heading', choices=['heading', 'control', 'tracking'])
    ap.add_argument('--actions', default='random', choices=['random', 'constant'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend of the timing barrier: 'nccl' = RCCL (default); 'gloo' lets the multi-rank path "
                         'be exercised on a box with fewer GPUs than ranks (ranks then share GPUs: a functional check, not a benchmark)')
    ap.add_argument('--aero-1d-tables', type=int, default=None, choices=[0, 1],
                    help='numerics option (DESIGN.md §4); default: scenario / NPF16_AERO_1D_TABLES / off')
    args = ap.parse_args()

    from neuralplane_amd import sharding
    rank, local_rank, world = sharding.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f'--gpus {args.gpus} needs one process per GPU: launch with '
                             f'python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...')
        raise SystemExit(f'WORLD_SIZE={world} does not match --gpus {args.gpus}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path')
    ndev = torch.cuda.device_count()
    if args.backend == 'nccl' and local_rank >= ndev:
        raise SystemExit(f'LOCAL_RANK={local_rank} but only {ndev} GPU(s) visible')
    dev = torch.device('cuda', local_rank % ndev)
    torch.cuda.set_device(dev)
    dist = sharding.init_distributed(args.backend, dev)  # RCCL; used for the timing barrier / max only

    from neuralplane_amd.envs.control_env import ControlEnv
    n = args.n  # weak scaling: every GPU simulates args.n aircraft, global rows [rank*n, (rank+1)*n)
    row0, n_local = sharding.shard_rows(world * n, world, rank)
    assert n_local == n
    env = ControlEnv(num_envs=n, config=args.task, model='F16', random_seed=0, device=str(dev), row0=row0,
                     aero_1d_tables=args.aero_1d_tables)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    if args.actions == 'random':
        pool = [torch.rand((n, 4), generator=g, device=dev) * 2 - 1 for _ in range(8)]
    else:  # the reference benchmark's clamped constant action (measure_env.py:12-16,68-72)
        pool = [torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).repeat(n, 1)]

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    env.reset()
    for i in range(args.warmup):
        env.step(pool[i % len(pool)])
    env._batch.set_timing(True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        env.step(pool[i % len(pool)])
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_cnt = env._batch.get_timing()
    env._batch.set_timing(False)

    mem_mb = torch.cuda.max_memory_allocated(dev) / 2 ** 20   # env state + cache + outputs of one step + the 8-entry action pool
    elapsed = sharding.max_over_ranks(elapsed, dist, dev if args.backend == 'nccl' else 'cpu')
    # sanity of the timed region: states finite for live rows, counters advanced
    fin = bool(torch.isfinite(env.model.s).all().item())

    if rank == 0:
        value = world * n * args.steps / elapsed
        kern_s = kern_ms * 1e-3
        ach_tflops = n * ALGO_FLOP / kern_s / 1e12 if kern_s > 0 else 0.0
        ach_gbs = n * ALGO_BYTES / kern_s / 1e9 if kern_s > 0 else 0.0
        out = {
            'metric': 'aircraft-steps/sec at N=1e6 F-16 Heading; 1/2/4/8 MI355X scaling',
            'value': value, 'unit': 'aircraft-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'F-16 {args.task}, N={n} aircraft per GPU, euler dt=0.02, noise_scale per YAML, '
                                   f'{args.actions} actions, one fused HIP kernel per env.step',
                       'aircraft_per_gpu': n, 'sharding': f'rows split over {world} GPU(s), no data-path collective',
                       'cross_step_coefficient_reuse': True, 'aero_1d_tables': bool(env._batch.aero_1d_tables)},
            'roofline': {'bound': 'mfma', 'achieved': ach_tflops, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': ach_tflops / PEAK_FP32_TFLOPS, 'traffic': pmc_traffic(n, args.task),
                         'traffic_unit': 'HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/)', 'algorithmic_bytes_per_launch': n * ALGO_BYTES,
                         'kernel': 'f16_env_kernel<task,solver,STEP>', 'kernel_avg_ms': kern_ms, 'launches_timed': kern_cnt,
                         'note': 'fp32 VECTOR (VALU) roof: 157.3 TFLOP/s is both the fp32 vector peak and the '
                                 'f32-input MFMA peak on gfx950; the kernel issues no MFMA. achieved = 33.8 KFLOP x N / '
                                 'avg launch duration (HIP events on the launch stream)'},
            'roofline_hbm': {'bound': 'hbm', 'achieved': ach_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                             'frac': ach_gbs / PEAK_HBM_GBS, 'note': '278 algorithmic B per aircraft-step; not the binding roof'},
            'state_finite': fin,
            # the reference publishes 245.5 MB allocated after its N = 1e6 run (envs/measure_env/gpu_memory_neuralplane.npy)
            'device_memory_mb': mem_mb,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.task)
        if world == 1 and not env._batch.aero_1d_tables and not args.no_cpu_baseline:
            # optional numerics mode, reported beside the headline (never as `value`): the 22 single-input aero nets
            # through their exact piecewise-linear tables (same functions, different rounding)
            del env
            torch.cuda.empty_cache()
            env2 = ControlEnv(num_envs=n, config=args.task, model='F16', random_seed=0, device=str(dev), row0=row0, aero_1d_tables=1)
            env2.reset()
            for i in range(10):
                env2.step(pool[i % len(pool)])
            env2._batch.set_timing(True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            k2 = min(args.steps, 100)
            for i in range(k2):
                env2.step(pool[i % len(pool)])
            torch.cuda.synchronize(dev)
            el2 = time.perf_counter() - t1
            ms2, _ = env2._batch.get_timing()
            out['optional_modes'] = {'aero_1d_tables': {'value': n * k2 / el2, 'unit': 'aircraft-steps/s', 'kernel_avg_ms': ms2,
                                                       'steps': k2, 'note': 'not the headline: changes the rounding of 22 of the 42 aero '
                                                       'coefficients by ~1e-5 rel (tests: masks identical to the reference, HIP == oracle bit-exact)'}}
        if world == 1 and not args.no_cpu_baseline:
            # the other integrator of the reference (`solver: rk4`, torchdiffeq's 3/8 rule: 4 aero evaluations per step)
            torch.cuda.empty_cache()
            env4 = ControlEnv(num_envs=n, config=args.task, model='F16', random_seed=0, device=str(dev), row0=row0, solver='rk4')
            env4.reset()
            for i in range(5):
                env4.step(pool[i % len(pool)])
            env4._batch.set_timing(True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            k4 = min(args.steps, 50)
            for i in range(k4):
                env4.step(pool[i % len(pool)])
            torch.cuda.synchronize(dev)
            el4 = time.perf_counter() - t1
            ms4, _ = env4._batch.get_timing()
            out.setdefault('optional_modes', {})['solver_rk4'] = {
                'value': n * k4 / el4, 'unit': 'aircraft-steps/s', 'kernel_avg_ms': ms4, 'steps': k4,
                'algorithmic_tflops': n * 105e3 / (ms4 * 1e-3) / 1e12 if ms4 > 0 else 0.0,
                'note': '4 x 23.8 KFLOP aero evaluations + the Overload re-evaluation = 105 KFLOP per aircraft-step (SURVEY 8d); '
                        'reference parity of rk4 is unpinned (no artefact exercises it), HIP == oracle bit-exact'}
            del env4
        if world == 1 and not args.no_cpu_baseline and n == 1_000_000:
            # the same kernel on a batch that amortises launch, de-phasing delay and the last partial generation of workgroups
            # (N = 1e6 is 5.09 generations of 1536 workgroups): the kernel's asymptotic rate
            torch.cuda.empty_cache()
            nb = 10_000_000
            env5 = ControlEnv(num_envs=nb, config=args.task, model='F16', random_seed=0, device=str(dev))
            env5.reset()
            ab = torch.rand((nb, 4), generator=g, device=dev) * 2 - 1
            for i in range(3):
                env5.step(ab)
            env5._batch.set_timing(True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            k5 = 20
            for i in range(k5):
                env5.step(ab)
            torch.cuda.synchronize(dev)
            el5 = time.perf_counter() - t1
            ms5, _ = env5._batch.get_timing()
            out.setdefault('optional_modes', {})['batch_1e7'] = {
                'value': nb * k5 / el5, 'unit': 'aircraft-steps/s', 'kernel_avg_ms': ms5, 'steps': k5,
                'fp32_roof_frac': nb * ALGO_FLOP / (ms5 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if ms5 > 0 else 0.0,
                'note': 'N = 1e7 aircraft on one GPU (3 GB of state + observations of the 288 GB): same kernel, same numerics'}
            del env5, ab
        if world == 1 and not args.no_cpu_baseline:
            # BASELINE.json configs[0] size (N = 256 = 4 waves): the latency regime — microseconds per env.step, not a roofline
            envs = ControlEnv(num_envs=256, config=args.task, model='F16', random_seed=0, device=str(dev))
            envs.reset()
            a_s = torch.rand((256, 4), generator=g, device=dev) * 2 - 1
            for i in range(50):
                envs.step(a_s)
            envs._batch.set_timing(True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            k6 = 1000
            for i in range(k6):
                envs.step(a_s)
            torch.cuda.synchronize(dev)
            el6 = time.perf_counter() - t1
            ms6, _ = envs._batch.get_timing()
            out.setdefault('optional_modes', {})['latency_n256'] = {
                'value': 1e6 * el6 / k6, 'unit': 'us per env.step (wall, back-to-back launches)', 'kernel_avg_us': 1e3 * ms6, 'steps': k6,
                'aircraft_steps_per_s': 256 * k6 / el6,
                'note': 'one wave executes the whole step serially (~12.6 K VALU instructions): latency-bound, the GPU is idle otherwise'}
            del envs
        if world == 1 and not args.no_cpu_baseline:
            # BASELINE.json configs[4] (a parity-test case, reported beside the headline, never as `value`): SingleCombat 1v1,
            # 1e5 engagements = 2e5 aircraft, one launch per env.step = 5 FDM steps behind the attitude PID stack
            from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
            torch.cuda.empty_cache()
            E = 100_000
            cenv = SingleCombatEnv(num_envs=E, config='selfplay', random_seed=0, device=str(dev))
            cenv.reset()
            g2 = torch.Generator(device='cpu').manual_seed(5)
            cpool = [(torch.rand((2 * E, 4), generator=g2) * 2 - 1).to(dev) for _ in range(4)]
            for i in range(10):
                cenv.step(cpool[i % 4])
            cenv._batch.set_timing(True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            k3 = min(args.steps, 100)
            for i in range(k3):
                cenv.step(cpool[i % 4])
            torch.cuda.synchronize(dev)
            el3 = time.perf_counter() - t1
            ms3, _ = cenv._batch.get_timing()
            out.setdefault('optional_modes', {})['singlecombat_1v1'] = {
                'value': E * k3 / el3, 'unit': 'engagement-steps/s', 'aircraft_fdm_steps_per_s': 2 * E * 5 * k3 / el3,
                'kernel_avg_ms': ms3, 'steps': k3, 'engagements': E,
                'note': 'one f16_combat_kernel launch per SingleCombatEnv.step (pairwise reset, 5 x {PID stack, FDM step, '
                        'terminations}, pairwise obs/reward/blood); HIP == oracle bit-exact (tests/test_gpu_combat_parity.py)'}
        if world == 1 and not args.no_cpu_baseline:
            # BASELINE.json configs[3] (hierarchical Tracking, a parity-test case reported beside the headline): PlanningEnv.step =
            # 50 x {low-level obs, frozen PPOActor-architecture controller as ONE fused MFMA kernel, fused env step}, at the batch
            # size of the reference's own training script (n = 1e4); random-init controller weights of that architecture
            import numpy as np
            from neuralplane_amd.actor import FusedActor, NUM_FLOATS
            from neuralplane_amd.envs.planning_env import PlanningEnv
            torch.cuda.empty_cache()
            npl = 10_000
            ctrl = FusedActor(np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32), str(dev))
            penv = PlanningEnv(num_envs=npl, config='tracking', model='F16', random_seed=0, device=str(dev), controller=ctrl)
            ap = torch.rand((npl, 3), generator=g, device=dev) * 2 - 1
            for i in range(3):
                penv.step(ap)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            k7 = 20
            for i in range(k7):
                penv.step(ap)
            torch.cuda.synchronize(dev)
            el7 = time.perf_counter() - t1
            out.setdefault('optional_modes', {})['planning_tracking_n1e4'] = {
                'value': 1e3 * el7 / k7, 'unit': 'ms per PlanningEnv.step (50 inner FDM steps + 50 controller calls)', 'steps': k7,
                'aircraft_fdm_steps_per_s': npl * 50 * k7 / el7,
                'note': 'controller = np_actor_forward (K=1 fp32 MFMA chains, bit-exact to its oracle: tests/test_gpu_actor.py); '
                        'the same step with the controller as eager torch modules: 22 ms (tools/microbench/planning_bench.py)'}
            del penv, ctrl
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
