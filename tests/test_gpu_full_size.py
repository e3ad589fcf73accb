"""GPU tests at BASELINE.json's full size (N = 1e6, F-16 Heading / Control / Tracking): parity through
size-independent properties — aircraft are independent, so (1) any sampled block of rows, re-run through the
CPU oracle with the same global row indices and the same actions, must match bit for bit after many fused
steps; (2) the batch split into shards with `row0` offsets must reproduce the unsharded batch bit for bit;
(3) the same seed reproduces the same trajectory; (4) flagged rows are re-initialised by the next step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.f16_oracle import Oracle  # noqa: E402  (the checker)

N = 1_000_000


def _env(task, n, seed, row0=0):
    from neuralplane_amd.envs.control_env import ControlEnv
    return ControlEnv(num_envs=n, config=task, model='F16', random_seed=seed, device='cuda:0', row0=row0)


def _actions(steps, n, seed):
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)
    return [(torch.rand((n, 4), generator=g, device='cuda') * 2.6 - 1.3) for _ in range(steps)]


@pytest.mark.parametrize('task', ['heading', 'control', 'tracking'])
def test_sampled_blocks_match_oracle_bit_for_bit_at_full_size(task):
    steps, seed = 25, 2024
    env = _env(task, N, seed)
    acts = _actions(steps, N, 77)
    env.reset()
    outs = [env.step(a) for a in acts]
    torch.cuda.synchronize()
    rng = np.random.RandomState(1)
    starts = [0, N - 200] + [int(x) for x in rng.randint(0, N - 200, 10)]   # first rows, ragged tail, random blocks
    o = Oracle(task, threads=8)
    s_gpu = env.model.s
    n_flag = 0
    for r0 in starts:
        m = 200
        st = Oracle.new_state(m)
        o.reset(st, seed=seed, call_idx=0, row0=r0)
        for t in range(steps):
            a = acts[t][r0:r0 + m].cpu().numpy()
            o_obs, o_rew, o_done, o_bad, _ = o.step(st, a, seed=seed, call_idx=t + 1, row0=r0)
            obs, rew, done, bad, tmo, _ = outs[t]
            assert np.array_equal(obs[r0:r0 + m].cpu().numpy(), o_obs), f'{task}: obs, rows {r0}.., step {t}'
            assert np.array_equal(rew[r0:r0 + m].cpu().numpy(), o_rew, equal_nan=True)
            assert np.array_equal(done[r0:r0 + m].cpu().numpy(), o_done.astype(bool))
            assert np.array_equal(bad[r0:r0 + m].cpu().numpy(), o_bad.astype(bool))
            n_flag += int(o_bad.sum() + o_done.sum())
        assert np.array_equal(s_gpu[r0:r0 + m].cpu().numpy(), st['s'], equal_nan=True), f'{task}: final state rows {r0}..'
        assert np.array_equal(env.step_count[r0:r0 + m].cpu().numpy(), st['step_count'])
    assert n_flag > 0, 'the sample never exercised a termination / auto-reset'


@pytest.mark.parametrize('n', [20_000, 57_344, 81_921, 120_000, 262_144])
def test_automatic_variant_at_the_mid_sizes_matches_oracle(n):
    """The variant np_f16_step picks on its own at each of the batch-size ranges the reference trains in and above — latency
    (four waves per tile), latency4w (the same at four waves per SIMD), latency2, the pair variant capped at two waves per SIMD and
    at three — against the oracle on sampled blocks, 20 fused steps with auto-resets (hazard-rich actions), ragged last tile."""
    steps, seed = 20, 11
    env = _env('heading', n, seed)
    acts = _actions(steps, n, 5)
    env.reset()
    outs = [env.step(a) for a in acts]
    torch.cuda.synchronize()
    o = Oracle('heading', threads=8)
    n_flag = 0
    for r0 in (0, n - 150, n // 3):
        m = 150
        st = Oracle.new_state(m)
        o.reset(st, seed=seed, call_idx=0, row0=r0)
        for t in range(steps):
            o_obs, o_rew, o_done, o_bad, _ = o.step(st, acts[t][r0:r0 + m].cpu().numpy(), seed=seed, call_idx=t + 1, row0=r0)
            obs, rew, done, bad, tmo, _ = outs[t]
            assert np.array_equal(obs[r0:r0 + m].cpu().numpy(), o_obs), f'n={n}: obs, rows {r0}.., step {t}'
            assert np.array_equal(rew[r0:r0 + m].cpu().numpy(), o_rew, equal_nan=True)
            assert np.array_equal(bad[r0:r0 + m].cpu().numpy(), o_bad.astype(bool)) and np.array_equal(done[r0:r0 + m].cpu().numpy(), o_done.astype(bool))
            n_flag += int(o_bad.sum() + o_done.sum())
        assert np.array_equal(env.model.s[r0:r0 + m].cpu().numpy(), st['s'], equal_nan=True), f'n={n}: final state rows {r0}..'
    assert n_flag > 0


def test_sharded_batches_reproduce_the_unsharded_batch():
    """Two shards with row0 offsets == the full batch (what every rank of a multi-GPU run relies on)."""
    n, steps, seed = 300_001, 12, 5          # odd size: ragged last workgroup in both shards
    acts = _actions(steps, n, 3)
    full = _env('heading', n, seed)
    full.reset()
    for a in acts:
        o_full = full.step(a)
    cut = 123_457
    parts = []
    for r0, r1 in ((0, cut), (cut, n)):
        e = _env('heading', r1 - r0, seed, row0=r0)
        e.reset()
        for a in acts:
            out = e.step(a[r0:r1])
        parts.append((e, out))
    s = torch.cat([p[0].model.s for p in parts])
    assert torch.equal(s, full.model.s)
    for k in range(5):
        assert torch.equal(torch.cat([p[1][k] for p in parts]), o_full[k])


@pytest.mark.parametrize('solver', ['euler', 'rk4'])
def test_pair_kernel_builds_agree_across_the_grid_size_rule(solver):
    """The pair variant exists in two builds (two / three waves per SIMD, the second with cold registers parked in scratch);
    launch_env picks one per grid size (two waves up to 1 024 workgroups of 128 aircraft, three — de-phased — above; until the end of
    round 3 the two-wave build also served 1 537-3 071 workgroups).  On either side of every old and new threshold the default must
    equal the single-set throughput variant bit for bit."""
    steps, seed = 4, 11
    for wgs in (1024, 1025, 1536, 1537, 3071, 3072):
        n = wgs * 128 - 37                    # ragged last workgroup
        acts = _actions(steps, n, wgs)
        res = []
        for variant in ('auto', 'throughput'):
            from neuralplane_amd.envs.control_env import ControlEnv
            e = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=seed, device='cuda:0', solver=solver)
            e._batch.set_kernel_variant(variant)
            e.reset()
            for a in acts:
                out = e.step(a)
            res.append((e.model.s.clone(), [x.clone() for x in out[:5]]))
            del e
        assert torch.equal(res[0][0], res[1][0]), wgs
        for k in range(5):
            assert torch.equal(res[0][1][k], res[1][1][k]), (wgs, k)


def test_inner_step_pair_builds_agree_across_the_grid_size_rule():
    """PlanningEnv's inner iteration (np_f16_io.inner_step + ll_obs) follows the same grid-size rule as the plain step: two waves per
    SIMD up to 1 024 workgroups, the three-wave build above.  On either side of the threshold the default must equal the single-set
    throughput variant bit for bit — states, task observation, reward, flags and the low-level observation the launch writes."""
    from neuralplane_amd.envs.control_env import ControlEnv
    steps, seed = 4, 5
    for wgs in (1024, 1025, 1537):
        n = wgs * 128 - 19
        acts = _actions(steps, n, wgs + 1)
        tgt3 = (torch.rand((3, n), generator=torch.Generator().manual_seed(wgs)) * 0.4 - 0.2).to('cuda').contiguous()
        res = []
        for variant in ('auto', 'throughput'):
            e = ControlEnv(num_envs=n, config='tracking', model='F16', random_seed=seed, device='cuda:0')
            b = e._batch
            b.set_kernel_variant(variant)
            b.reset()
            ll = torch.empty((n, 22), dtype=torch.float32, device='cuda')
            for a in acts:
                obs, rew, flags = b.step(a, inner=True, ll_tgt=tgt3, ll_obs=ll, want_obs=True)
            res.append((b.s.clone(), obs.clone(), rew.clone(), flags.clone(), ll.clone()))
            del e, b
        for k in range(5):
            assert torch.equal(res[0][k], res[1][k]), (wgs, k)


def test_same_seed_same_trajectory_and_flagged_rows_are_reinitialised():
    n, steps = N, 30
    acts = _actions(steps, n, 11)
    runs = []
    for _ in range(2):
        e = _env('heading', n, 99)
        e.reset()
        for a in acts:
            out = e.step(a)
        runs.append((e.model.s.clone(), out[0].clone(), out[3].clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    bad = runs[0][2]
    assert bool(bad.any()) and bool(torch.isfinite(runs[0][0][~bad]).all())
    # one more step: every flagged row restarts from the reset distribution and has step_count == 1
    e.step(acts[0])
    sc = e.step_count[bad]
    assert bool((sc == 1).all())
    alt0 = e.model.s[bad, 2]
    assert bool(((alt0 > 18000) & (alt0 < 21000)).all())


def _bench_json(args, timeout=300, raw=False):
    """Runs bench.py as a plain command.  stdout must END with the contract line (the only line that starts with '{', < 4 KB, the line the
    driver parses) and hold one BENCH_DETAILS line before it; returns the details dict (a superset) after checking that every scalar
    of the contract line equals the details' value."""
    import json
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    lines = [ln for ln in out_lines if ln.startswith('{')]
    assert len(lines) == 1 and out_lines[-1] == lines[0], r.stdout[-2000:]                     # rank 0 only, and LAST
    assert len(lines[0]) < 4000, len(lines[0])
    line = json.loads(lines[0])
    det = [ln for ln in out_lines if ln.startswith('BENCH_DETAILS ')]
    assert len(det) == 1
    d = json.loads(det[0][len('BENCH_DETAILS '):])
    for k, v in line.items():
        if not isinstance(v, dict) and k != 'details':
            assert d[k] == v, k
    assert line['roofline']['frac'] == d['roofline']['frac'] and line['roofline']['kernel_avg_ms'] == d['roofline']['kernel_avg_ms']
    assert line['ms_per_step'] * 1e-3 * line['steps'] < wall            # the timed region fits inside the run's own wall clock
    if line.get('details'):
        assert json.load(open(os.path.join(root, line['details'])))['value'] == line['value']
    return (d, line, wall) if raw else d


def test_bench_multi_rank_path_on_one_gpu():
    """`python bench.py --gpus 2 ...` as a PLAIN command (no launcher environment): the script re-executes itself under
    torch.distributed.run, one process per rank — here with the gloo backend so that two ranks can share this box's single GPU:
    sharding, barrier, max-over-ranks and the rank-0 JSON line of the weak-scaling path are exercised on real hardware (RCCL
    itself is not)."""
    d = _bench_json(['--gpus', '2', '--steps', '20', '--warmup', '3', '--n', '200000', '--backend', 'gloo', '--prelude-ms', '50'])
    assert d['n_gpus'] == 2 and d['world_size'] == 2 and d['steps'] == 20 and d['scaling'] == 'weak' and d['state_finite'] is True
    assert d['backend'] == 'gloo' and d['rccl_ranks'] == 0
    assert d['value'] == pytest.approx(2 * 200000 * 20 / (d['ms_per_step'] * 1e-3 * 20), rel=1e-6)
    assert 'cpu_baseline' not in d and d['roofline']['frac'] > 0 and d['roofline']['bound'] == 'valu'
    assert d['prelude']['steps'] > 0 and d['cold_start']['launches_timed'] == 20 and d['roofline']['launches_timed'] == 20


def test_bench_single_gpu_line_has_the_contract_fields():
    d = _bench_json(['--steps', '20', '--warmup', '5', '--n', '100000', '--headline-only', '--prelude-ms', '50'])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline'):
        assert k in d, k
    r = d['roofline']
    assert r['bound'] == 'valu' and r['unit'] == 'TFLOP/s' and r['frac'] == pytest.approx(r['achieved'] / r['peak'])
    assert r['executed_flop_per_aircraft_step'] < r['algorithmic_flop_per_aircraft_step'] and r['executed_frac'] < r['frac']
    assert r['traffic'] is None or r['traffic_source'].startswith('profiles/')
    assert 1500 < r['effective_shader_mhz'] < 2600
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 0 and d['steps'] == 20 and d['warmup'] == 5
    assert d['world_size'] == 1 and d['backend'] is None and len(d['per_rank']['host_enqueue_us_per_step']) == 1
    assert d['expected_scaling']['host_keeps_gpu_fed'] in (True, False) and d['expected_scaling']['measured_at_world_size'] == 1


def test_bench_driver_command_last_line_is_the_small_contract_line():
    """The driver's own command (`bench.py --gpus 1 --steps 20 --warmup 5`, full size, CPU baselines included): the LAST stdout line is
    the contract line and nothing else — < 4 KB (round 5's 28.8 KB line was not parsed), json.loads succeeds, it carries value,
    ms_per_step, roofline.frac and cpu_baseline.value, K x ms_per_step fits inside the run's wall time, and the run stays under 60 s
    (40 s is the target on a warm box; the first `import torch` of a fresh one takes longer)."""
    d, line, wall = _bench_json(['--gpus', '1', '--steps', '20', '--warmup', '5'], raw=True)
    assert set(line) >= {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                         'dtype', 'data', 'config', 'roofline', 'cpu_baseline'}
    assert line['metric'].startswith('aircraft-steps/sec at N=1e6 F-16 Heading') and line['config']['aircraft_per_gpu'] == 1_000_000
    assert line['value'] == pytest.approx(1_000_000 * 20 / (line['ms_per_step'] * 1e-3 * 20), rel=1e-6) and line['value'] > 1e9
    r, c = line['roofline'], line['cpu_baseline']
    assert 0.3 < r['frac'] < 1.0 and r['executed_frac'] < r['frac'] and r['kernel_avg_ms'] <= line['ms_per_step'] * 1.001
    assert r['algorithmic_bytes_per_launch'] == 278.0 * 1_000_000 and (r['traffic'] is None or r['traffic'] >= r['algorithmic_bytes_per_launch'])
    assert c['value'] > 0 and c['cores'] >= 1 and c['kind'] == 'port' and 'oracle/f16_oracle.c' in c['sample'] and c['torch_eager'] > 0
    assert 'optional_modes' not in d and wall < 60, wall


@pytest.mark.parametrize('task', ['tracking', 'combat'])
def test_bench_other_configs_multi_rank(task):
    """BASELINE.json configs[3] (Tracking, rows sharded) and configs[4] (SingleCombat, envs sharded, the opponent exchange inside
    the stepped loop) through the same plain command, two gloo ranks on this GPU."""
    extra = ['--n', '100000'] if task == 'tracking' else ['--engagements', '20001']     # odd: ragged shards
    d = _bench_json(['--gpus', '2', '--task', task, '--steps', '10', '--warmup', '2', '--backend', 'gloo', '--prelude-ms', '20'] + extra)
    assert d['n_gpus'] == 2 and d['state_finite'] is True and d['value'] > 0
    if task == 'combat':
        assert d['scaling'] == 'strong' and d['exchange']['collectives_per_step'] == 2 and d['unit'] == 'engagement-steps/s'


def test_bench_eight_ranks_on_one_gpu_heading():
    """The command the driver's SCALE run issues at N = 8 (`bench.py --gpus 8 ...`), with eight gloo ranks sharing this box's one GPU:
    rendezvous on 127.0.0.1, per-rank row shards, the barriers, the max-over-ranks, ONE JSON line from rank 0 — and the fields that
    let the launch skew of an 8-process run be read off the line: the live group's size / backend and every rank's own host time."""
    d = _bench_json(['--gpus', '8', '--steps', '10', '--warmup', '2', '--n', '20000', '--backend', 'gloo', '--prelude-ms', '20', '--headline-only'], timeout=600)
    assert d['n_gpus'] == 8 and d['world_size'] == 8 and d['backend'] == 'gloo' and d['rccl_ranks'] == 0 and d['scaling'] == 'weak'
    assert 'get_world_size' in d['group_source'] and d['state_finite'] is True
    assert d['value'] == pytest.approx(8 * 20000 * 10 / (d['ms_per_step'] * 1e-3 * 10), rel=1e-6)
    pr = d['per_rank']
    assert len(pr['host_enqueue_us_per_step']) == 8 and len(pr['elapsed_ms_per_step']) == 8
    assert all(v > 0 for v in pr['host_enqueue_us_per_step']) and max(pr['elapsed_ms_per_step']) == pytest.approx(d['ms_per_step'], rel=1e-6)
    es = d['expected_scaling']
    assert es['measured_at_world_size'] == 8 and len(es['predicted_speedup_8_gpus']) == 2


@pytest.mark.parametrize('lag', [0, 1])
def test_bench_eight_ranks_on_one_gpu_combat(lag):
    """BASELINE.json configs[4] at eight ranks (gloo, one GPU): envs sharded by engagement, two all-gathers per step on the side
    stream, opponent lag 0 (the reference runner's semantics) and 1 (the exchange overlaps the env kernel)."""
    d = _bench_json(['--gpus', '8', '--task', 'combat', '--engagements', '8000', '--steps', '6', '--warmup', '2', '--backend', 'gloo',
                     '--prelude-ms', '20', '--opponent-lag', str(lag), '--headline-only'], timeout=600)
    assert d['n_gpus'] == 8 and d['world_size'] == 8 and d['backend'] == 'gloo' and d['state_finite'] is True and d['value'] > 0
    assert d['scaling'] == 'strong' and d['exchange']['collectives_per_step'] == 2 and d['exchange']['opponent_lag'] == lag
    assert len(d['per_rank']['host_enqueue_us_per_step']) == 8


@pytest.mark.parametrize('task', ['heading', 'tracking'])
def test_soak_long_horizon_block_stays_bit_exact(task):
    """6000 consecutive env.steps (2.4 episodes of the 2500-step limit, hundreds of auto-resets, in-kernel noise and reset
    draws) on a 70 000-aircraft batch: a 192-row block in the middle is re-simulated by the oracle with the same global row
    keys and must agree bit for bit at the end — state, counters, masks, the last observation and reward."""
    n, steps, seed, lo, cnt = 70_000, 6000, 99, 33_333, 192
    b = _env(task, n, seed)._batch
    o = Oracle(task)
    st = Oracle.new_state(cnt)
    g = torch.Generator(device='cpu').manual_seed(7)
    pool = [(torch.rand((n, 4), generator=g) * 2 - 1) for _ in range(16)]
    pool_dev = [p.cuda() for p in pool]
    pool_np = [p[lo:lo + cnt].numpy().copy() for p in pool]
    resets = 0
    for t in range(steps):
        obs, rew, flags = b.step(pool_dev[t % 16])
        o_obs, o_rew, dn, bd, tm = o.step(st, pool_np[t % 16], seed=seed, call_idx=t, row0=lo)
        resets += int((dn | bd | tm).sum())
    rows = slice(lo, lo + cnt)
    assert np.array_equal(b.s.cpu().numpy().T[rows], st['s']) and np.array_equal(b.u.cpu().numpy().T[rows], st['u'])
    assert np.array_equal(b.step_count.cpu().numpy()[rows], st['step_count'])
    f = flags.cpu().numpy()
    assert np.array_equal(f[0][rows], st['done']) and np.array_equal(f[1][rows], st['bad'])
    assert np.array_equal(obs.cpu().numpy()[rows], o_obs) and np.array_equal(rew.cpu().numpy()[rows], o_rew)
    assert resets > cnt       # every aircraft of the block went through at least one episode boundary on average
    assert torch.isfinite(b.s).all()


def test_drop_in_example_runs():
    """examples/drop_in_rollout.py — host code written against the reference's import paths (numpy VecEnv loop, device-resident
    loop, device-resident collection into DeviceReplayBuffer + returns + mini-batches) runs end to end."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'examples', 'drop_in_rollout.py'), '512', '24'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'numpy VecEnv loop' in out.stdout and 'device-resident loop' in out.stdout and 'device-resident collection' in out.stdout


def test_drop_in_planning_example_runs():
    """examples/drop_in_planning.py — `from envs.planning_env import PlanningEnv` under GPUVecEnv with `controller='fused'` and a checkpoint
    file in the reference's format: the macro-steps run on the persistent kernel with the default numerics and never fall back."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'examples', 'drop_in_planning.py'), '1000', '6'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'controller numerics i8, fallbacks 0' in out.stdout and out.stdout.rstrip().endswith('OK')
