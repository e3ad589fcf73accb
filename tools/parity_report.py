#!/usr/bin/env python3
"""SURVEY.md §8(d) parity report: an engine (the HIP env on a GPU box, or the CPU oracle) against what the REFERENCE recorded.

    python tools/parity_report.py --engine hip    [--out profiles/r02_parity.json]      # GPU box
    python tools/parity_report.py --engine oracle [--out ...]                            # anywhere

Inputs are committed data only (tests/golden/traj_*.npz — trajectories of the imported reference, plain ATen arithmetic,
its reset draws recorded; tests/golden/recorded_episode0.npz — the authors' own CUDA recording, renders/result/*.npy).
Metric: per state k  err = |x - x_ref| / max(|x_ref|, floor_k), floors (npos, epos, alt: 100 ft; angles:
0.1 rad; vt: 10 ft/s; P, Q, R: 0.1 rad/s); per aircraft the max over the 12 states; reported as median / p90 / p99 / max over
the aircraft that still follow the reference's episode schedule at step t (an aircraft whose done/bad_done mask differed once
has reset at a different time and is counted in `rows_diverged` from then on), at t in {1, 10, 100, 426, 1000} (Heading) or
{1, 10, 100, 300}.  tests/test_gpu_step_parity.py asserts the bounds on the same numbers and writes this report.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
STATE_FLOORS = np.array([100, 100, 100, 0.1, 0.1, 0.1, 10, 0.1, 0.1, 0.1, 0.1, 0.1], np.float32)
TRAJ = (('heading', 256, 1000, (1, 10, 100, 426, 1000)), ('control', 64, 300, (1, 10, 100, 300)), ('tracking', 64, 300, (1, 10, 100, 300)))


def traj_actions(T, n, seed=123):
    """The action sequence of the fixtures (tools/gen_golden.py::traj_actions; numpy RandomState is version-stable)."""
    rng = np.random.RandomState(seed)
    t = np.arange(T, dtype=np.float64)[:, None, None]
    phase = rng.uniform(0, 2 * np.pi, (1, n, 4))
    freq = rng.uniform(0.002, 0.02, (1, n, 4))
    a = 0.3 * np.sin(2 * np.pi * freq * t + phase) + rng.uniform(-1, 1, (T, n, 4)) * np.array([1.0, 0.3, 0.3, 0.3])
    a[..., 0] = 0.5 + 0.5 * a[..., 0]
    return np.clip(a, -1, 1).astype(np.float32)


def closed_loop_setup(n, seed=321):
    """Per-aircraft attitude commands and the per-step dither of the closed-loop fixture (numpy RandomState: version-stable)."""
    rng = np.random.RandomState(seed)
    th_cmd = rng.uniform(-0.05, 0.10, n).astype(np.float32)
    phi_cmd = rng.uniform(-0.3, 0.3, n).astype(np.float32)
    return th_cmd, phi_cmd, rng


def closed_loop_action(s, th_cmd, phi_cmd, dither):
    """A deterministic attitude-hold policy, a pure float32 function of the CURRENT state s[n, 12]: pitch and roll PD loops on
    elevator / aileron, a yaw damper, fixed throttle, plus a small recorded-seed dither.  It keeps every aircraft flying for the
    whole 1000 steps, so the fixture measures how an engine's round-off grows over LONG uninterrupted episodes with the policy
    in the loop (the open-loop random-action fixture resets each aircraft every ~35 steps)."""
    f = np.float32
    a = np.empty((s.shape[0], 4), np.float32)
    a[:, 0] = f(0.3)
    a[:, 1] = f(2.0) * (s[:, 4] - th_cmd) + f(1.0) * s[:, 10]
    a[:, 2] = f(1.0) * (s[:, 3] - phi_cmd) + f(0.5) * s[:, 9]
    a[:, 3] = f(0.2) * s[:, 11]
    a[:, 1:] += dither
    return np.clip(a, f(-1), f(1)).astype(np.float32)


class OracleEngine:
    name = 'oracle (oracle/f16_oracle.c, CPU)'

    def __init__(self, task, n, overrides=None):
        from oracle.f16_oracle import Oracle
        self.o = Oracle(task, overrides=dict(overrides or {}, noise_scale=0))
        self.st = Oracle.new_state(n)

    def step(self, action, rand_u):
        _, _, d, b, tm = self.o.step(self.st, action, rand_u=rand_u)
        return self.st['s'], np.stack([d, b, tm], 1).astype(bool)

    def targets(self):
        return self.st['tgt']

    def reset(self, rand_u):
        self.o.reset(self.st, rand_u=rand_u, want_obs=False)

    def set_state(self, s):
        self.st['s'][:] = s

    @staticmethod
    def xdot(s, u):
        from oracle.f16_oracle import Oracle
        return Oracle('heading').nlplant(np.hstack([s, u]).astype(np.float32))


class HipEngine:
    name = 'hip (libneuralplane_hip.so on cuda:0)'

    def __init__(self, task, n, overrides=None):
        import torch
        from neuralplane_amd.core import F16Batch
        from neuralplane_amd.envs.utils.utils import parse_config
        cfg = parse_config(task)
        for k, v in (overrides or {}).items():
            setattr(cfg, k, v)
        cfg.noise_scale = 0
        self.torch = torch
        self.b = F16Batch(n, cfg, task, 'cuda:0', seed=0)

    def step(self, action, rand_u):
        _, _, flags = self.b.step(self.torch.from_numpy(action).cuda(), rand_u=rand_u)
        return self.b.s.cpu().numpy().T, flags.cpu().numpy().astype(bool).T

    def targets(self):
        return self.b.tgt.cpu().numpy().T

    def reset(self, rand_u):
        self.b.reset(rand_u=rand_u, want_obs=False)

    def set_state(self, s):
        self.b.s.copy_(self.torch.from_numpy(np.ascontiguousarray(s.T)))   # a caller writing model.s between two steps (the cache keys notice)

    @staticmethod
    def xdot(s, u):
        import torch
        from neuralplane_amd.core import F16Batch
        from neuralplane_amd.envs.utils.utils import parse_config
        b = HipEngine._xb = getattr(HipEngine, '_xb', None) or F16Batch(1, parse_config('heading'), 'heading', 'cuda:0', seed=0)
        b.s.copy_(torch.from_numpy(np.ascontiguousarray(s.T)).cuda())
        b.u.copy_(torch.from_numpy(np.ascontiguousarray(u.T)).cuda())
        return b.derived()[:12].cpu().numpy().T


PID_TRAJ = (('heading', 64, 2600, (1, 100, 1000, 1500, 2000, 2500, 2600)), ('control', 64, 400, (1, 20, 100, 200, 300, 400)))


def trajectory_report(engine_cls, task, n, T, at, fixture=None):
    """fixture = None: tests/golden/traj_<task>_N<n>_T<T>.npz, open-loop actions from traj_actions().  fixture = 'traj_pid_...': a trajectory
    flown by the reference's own PID stack (tools/gen_golden.py::gen_traj_pid_*): the actions the reference env stepped on are in the fixture
    (int16 multiples of 1 / action_quantum), as are the scenario keys it changed; in these `done` fires and rows are re-initialised with new
    targets mid-trajectory (`done_events`, `first_steps_after_done_compared` count them)."""
    g = np.load(os.path.join(GOLDEN, fixture or f'traj_{task}_N{n}_T{T}.npz'))
    if 'actions_q' in g.files:
        acts = (g['actions_q'].astype(np.float32) / np.float32(g['action_quantum'])).astype(np.float32)
        overrides = json.loads(str(g['overrides'])) if 'overrides' in g.files else None
        eng = engine_cls(task, n, overrides)
        eng.reset(g['rand_u_reset'])     # the PID stack reads the env's state before the first step: the reference run began with env.reset()
    else:
        acts = traj_actions(T, n)
        eng = engine_cls(task, n)
    rec = {int(t): i for i, t in enumerate(g['rec_steps'])}
    diverged = np.zeros(n, bool)
    first_mask_diff, rows = 0, []
    done_events = after_done_compared = 0
    after_done_worst = tgt_worst = 0.0
    edits = {int(t): i for i, t in enumerate(g['state_edit_steps'])} if 'state_edit_steps' in g.files else {}
    for t in range(T):
        if t in edits:     # the reference's TECS writes the altitude it read (a view of model.s) in place on its first call: replayed here
            eng.set_state(g['state_edits'][edits[t]])
        s, f = eng.step(acts[t], g['rand_u'][t])
        diff = (f != g['flags'][t].astype(bool)).any(axis=1)
        first_mask_diff += int((diff & ~diverged).sum())
        diverged |= diff
        if fixture and t > 0 and g['flags'][t - 1, :, 0].any() and t in rec:
            # the step that re-initialised the rows whose `done` fired: whole state redrawn (F16_model.py:37-45), then one step flown
            rows_d = g['flags'][t - 1, :, 0].astype(bool) & ~diverged
            ref = g['state'][rec[t]][:, :12]
            if rows_d.any():
                after_done_compared += int(rows_d.sum())
                after_done_worst = max(after_done_worst, float(np.nanmax(np.abs(s[rows_d] - ref[rows_d]) / np.maximum(np.abs(ref[rows_d]), STATE_FLOORS))))
        done_events += int(g['flags'][t, :, 0].sum())
        if fixture and t in rec and (~diverged).any():   # the task's targets (re-drawn by task.reset after every done / bad)
            ref_t, got_t = g['state'][rec[t]][:, 16:19][~diverged], eng.targets()[~diverged]
            tgt_worst = max(tgt_worst, float(np.max(np.abs(got_t - ref_t) / np.maximum(np.abs(ref_t), np.float32(1.0)))))
        if (t + 1) in at:
            ref = g['state'][rec[t]][:, :12]
            e = np.nanmax(np.abs(s - ref) / np.maximum(np.abs(ref), STATE_FLOORS), axis=1)[~diverged]
            rows.append({'t': t + 1, 'rows_compared': int(e.size), 'rows_diverged': int(diverged.sum()), 'median': float(np.median(e)),
                         'p90': float(np.percentile(e, 90)), 'p99': float(np.percentile(e, 99)), 'max': float(e.max())})
    out = {'task': task, 'n': n, 'T': T, 'first_mask_differences': first_mask_diff, 'rows_diverged_final': int(diverged.sum()),
           'resets_in_reference': int(g['flags'].any(axis=2).sum()), 'at': rows}
    if fixture:
        out.update({'fixture': fixture, 'done_events_in_reference': done_events, 'bad_events_in_reference': int(g['flags'][:, :, 1].sum()),
                    'first_steps_after_done_compared': after_done_compared, 'first_step_after_done_max_rel': after_done_worst,
                    'targets_max_rel_at_recorded_steps': tgt_worst})
    return out


def closed_loop_report(engine_cls, n=256, T=1000, at=(1, 10, 100, 426, 1000)):
    """Policy in the loop on both sides: the engine computes its actions from ITS OWN state with the same float32 policy the
    reference run used on its state (tools/gen_golden.py::gen_traj_closed)."""
    g = np.load(os.path.join(GOLDEN, f'traj_heading_closed_N{n}_T{T}.npz'))
    th_cmd, phi_cmd, rng = closed_loop_setup(n)
    eng = engine_cls('heading', n)
    rec = {int(t): i for i, t in enumerate(g['rec_steps'])}
    diverged = np.zeros(n, bool)
    first_mask_diff, rows = 0, []
    s = np.zeros((n, 12), np.float32)      # the policy sees the state BEFORE the step; the first step resets every row anyway
    for t in range(T):
        dither = rng.uniform(-0.05, 0.05, (n, 3)).astype(np.float32)
        a = closed_loop_action(s, th_cmd, phi_cmd, dither)
        s, f = eng.step(a, g['rand_u'][t])
        s = np.array(s, np.float32)
        diff = (f != g['flags'][t].astype(bool)).any(axis=1)
        first_mask_diff += int((diff & ~diverged).sum())
        diverged |= diff
        if (t + 1) in at:
            ref = g['state'][rec[t]][:, :12]
            e = np.nanmax(np.abs(s - ref) / np.maximum(np.abs(ref), STATE_FLOORS), axis=1)[~diverged]
            rows.append({'t': t + 1, 'rows_compared': int(e.size), 'rows_diverged': int(diverged.sum()), 'median': float(np.median(e)),
                         'p90': float(np.percentile(e, 90)), 'p99': float(np.percentile(e, 99)), 'max': float(e.max())})
    return {'task': 'heading (closed loop: attitude-hold policy on the engine\'s own state)', 'n': n, 'T': T, 'first_mask_differences': first_mask_diff,
            'rows_diverged_final': int(diverged.sum()), 'resets_in_reference': int(g['flags'][1:].any(axis=2).sum()),
            'longest_episode_in_reference': int(g['step_count_final'].max()), 'at': rows}


def recorded_episode_report(engine_cls):
    """The authors' CUDA recording replayed: x_{t+1} = x_t + dt * xdot(x_t, recorded controls) with the engine's nlplant."""
    g = np.load(os.path.join(GOLDEN, 'recorded_episode0.npz'))
    rows, cols = g['rows'], list(g['columns'])
    ix = {c: cols.index(c) for c in cols}
    s = np.zeros((1, 12), np.float32)
    s[0, 2], s[0, 6] = rows[0, ix['altitude']], rows[0, ix['vt']]
    dt = np.float32(0.02)
    floors = STATE_FLOORS[:9]
    out, worst = [], 0.0
    for t in range(426):
        u = np.array([[rows[t + 1, ix['T']], rows[t + 1, ix['el']], rows[t + 1, ix['ail']], rows[t + 1, ix['rud']], 0]], np.float32)
        s = (s + dt * engine_cls.xdot(s, u)[:, :12]).astype(np.float32)
        ref = rows[t + 1, :9]
        e = float(np.max(np.abs(s[0, :9] - ref) / np.maximum(np.abs(ref), floors)))
        worst = max(worst, e)
        if (t + 1) in (1, 10, 100, 200, 400, 426):
            out.append({'t': t + 1, 'err': e, 'max_so_far': worst})
    # attribution of the end-of-episode residual: the REFERENCE's own CPU dynamics replayed the same way in the build container
    # (tools/gen_golden.py::gen_recorded_episode -> tests/golden/recorded_episode0_ref_cpu.npz), as it runs and with its MLPs / sin / cos / pow
    # in fp64; the oracle's pin mode against the latter, step by step
    from oracle.f16_oracle import MODE_LIBM, MODE_MLP_F64, Oracle
    r = np.load(os.path.join(GOLDEN, 'recorded_episode0_ref_cpu.npz'))
    o = Oracle('heading', mode=MODE_MLP_F64 | MODE_LIBM)
    sp = r['states_pin'][:1].copy()
    same, vs_ref_cpu, sq = True, 0.0, r['states'][:1].copy()
    for t in range(426):
        u = np.array([[rows[t + 1, ix['T']], rows[t + 1, ix['el']], rows[t + 1, ix['ail']], rows[t + 1, ix['rud']], 0]], np.float32)
        sp = (sp + dt * o.nlplant(np.hstack([sp, u]).astype(np.float32))).astype(np.float32)
        same = same and bool(np.array_equal(sp[0], r['states_pin'][t + 1]))
        sq = (sq + dt * engine_cls.xdot(sq, u)[:, :12]).astype(np.float32)
        vs_ref_cpu = max(vs_ref_cpu, float(np.max(np.abs(sq[0] - r['states'][t + 1]) / np.maximum(np.abs(r['states'][t + 1]), STATE_FLOORS))))
    att = {'this_engine_vs_cuda_recording': worst, 'reference_cpu_vs_cuda_recording': float(r['worst_vs_cuda_recording'][0]),
           'reference_cpu_pin_mode_vs_cuda_recording': float(r['worst_vs_cuda_recording'][1]),
           'this_engine_vs_reference_cpu_replay': vs_ref_cpu, 'oracle_pin_mode_equals_reference_cpu_pin_mode_bit_for_bit': same,
           'reading': 'the reference evaluated exactly (pin mode: MLPs in fp64, rounded once) ends as far from the CUDA recording as this build '
                      'does, and the oracle reproduces that pin-mode trajectory bit for bit through the departure: the 1.8e-4 is the fp32 noise '
                      'floor of the reference\'s own arithmetic (ATen-CPU and ATen-CUDA share an sgemm summation order and agree with each other '
                      '7 x better than with the correctly rounded result), amplified by the departure that ends the episode - not a restatement error'}
    return {'source': 'renders/result/*.npy rows 0..426 (the authors\' CUDA run of the reference), 9 recorded states', 'at': out, 'attribution': att}


def _dist(e):
    return {'median': float(np.median(e)), 'p99': float(np.percentile(e, 99)), 'max': float(e.max())}


def planning_report(engine):
    """PlanningEnv: the reference's PlanningEnv.step with a seeded random-init low-level actor, its recorded low-level
    actions replayed through reset / low_level_obs / 50 inner steps (tests/golden/planning_kat.npz; the reference ran plain ATen
    arithmetic).  Per high-level step: per-aircraft max over the 12 states of |x - ref| / max(|ref|, floor), rows still flying and
    rows frozen mid-step reported separately; masks counted.  Three evaluations of the SAME replay so that the residual can be
    attributed (SURVEY F7 method): the shipped numerics (fp32 fma chains), the oracle's pin mode (MLPs / sin / cos / pow in fp64,
    rounded once — the mode in which the oracle reproduces the reference's pinned fixtures bit for bit), and the two against each
    other.  If plain-vs-reference, pin-vs-reference and plain-vs-pin are the same size, the residual is the fp32 evaluation noise
    of equally valid orderings of the reference's own arithmetic, amplified by 50-150 closed-loop steps — not a restatement error."""
    from oracle.f16_oracle import MODE_MLP_F64, Oracle
    g = np.load(os.path.join(GOLDEN, 'planning_kat.npz'))
    hi = g['hi_actions']
    n = hi.shape[1]

    def run(mode):
        if engine == 'hip' and mode == 0:
            import torch
            from neuralplane_amd.core import F16Batch
            from neuralplane_amd.envs.utils.utils import parse_config
            b = F16Batch(n, parse_config('tracking'), 'tracking', 'cuda:0', seed=0)
            outs = []
            for k in range(hi.shape[0]):
                b.reset(rand_u=g[f'rand_u_{k}'], want_obs=False)
                a = np.clip(hi[k], -1, 1).astype(np.float32)
                s = b.s.cpu().numpy().T
                tgt3 = np.stack([s[:, 4] + a[:, 0] * np.float32(0.3), s[:, 5] + a[:, 1] * np.float32(0.3), s[:, 6] + a[:, 2] * np.float32(30)]).astype(np.float32)
                ll_err = 0.0
                for i in range(50):
                    ll = b.lowlevel_obs(torch.from_numpy(tgt3).cuda()).cpu().numpy()
                    ll_err = max(ll_err, float(np.max(np.abs(ll - g[f'll_obs_{k}'][i]) / np.maximum(np.abs(g[f'll_obs_{k}'][i]), 0.1))))
                    obs, rew, flags = b.step(torch.from_numpy(g[f'll_act_{k}'][i]).cuda(), inner=True)
                outs.append((b.s.cpu().numpy().T.copy(), obs.cpu().numpy(), rew.cpu().numpy(), flags.cpu().numpy(), ll_err))
            return outs
        o = Oracle('tracking', mode=mode)
        st = Oracle.new_state(n)
        outs = []
        for k in range(hi.shape[0]):
            o.reset(st, rand_u=g[f'rand_u_{k}'], want_obs=False)
            a = np.clip(hi[k], -1, 1).astype(np.float32)
            s = st['s']
            tgt3 = np.stack([s[:, 4] + a[:, 0] * np.float32(0.3), s[:, 5] + a[:, 1] * np.float32(0.3), s[:, 6] + a[:, 2] * np.float32(30)], 1).astype(np.float32)
            ll_err = 0.0
            for i in range(50):
                ll = o.lowlevel_obs(st, tgt3)
                ll_err = max(ll_err, float(np.max(np.abs(ll - g[f'll_obs_{k}'][i]) / np.maximum(np.abs(g[f'll_obs_{k}'][i]), 0.1))))
                obs, rew, d, bd, tm = o.step_inner(st, g[f'll_act_{k}'][i])
            outs.append((st['s'].copy(), obs.copy(), rew.copy(), np.stack([d, bd, tm]), ll_err))
        return outs

    plain, pin = run(0), run(MODE_MLP_F64)
    rows = []
    for k in range(hi.shape[0]):
        ref_s, fl = g[f's_{k}'], g[f'flags_{k}']
        live = ~fl[1].astype(bool)
        e = lambda a, b: np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), STATE_FLOORS), axis=1)  # noqa: E731
        ep, eq, epq = e(plain[k][0], ref_s), e(pin[k][0], ref_s), e(plain[k][0], pin[k][0])
        rows.append({'outer_step': k + 1, 'inner_steps_so_far': 50 * (k + 1), 'rows': int(n), 'rows_frozen_mid_step': int((~live).sum()),
                     'masks_equal_to_reference': bool(np.array_equal(plain[k][3].astype(bool), fl.astype(bool))),
                     'shipped_vs_reference_live': _dist(ep[live]), 'shipped_vs_reference_frozen': _dist(ep[~live]) if (~live).any() else None,
                     'pin_mode_vs_reference_live': _dist(eq[live]), 'shipped_vs_pin_mode_live': _dist(epq[live]),
                     'lowlevel_obs_max_rel': plain[k][4],
                     'obs_max_rel': float(np.max(np.abs(plain[k][1] - g[f'obs_{k}']) / np.maximum(np.abs(g[f'obs_{k}']), 0.1))),
                     'reward_max_rel': float(np.max(np.abs(plain[k][2] - g[f'reward_{k}']) / np.maximum(np.abs(g[f'reward_{k}']), 1.0)))})
    return {'fixture': 'tests/golden/planning_kat.npz (reference PlanningEnv.step, 3 high-level steps x 50 inner steps, seeded random-init actor)',
            'engine_for_shipped_numerics': 'hip' if engine == 'hip' else 'oracle', 'at': rows,
            'reading': 'shipped-vs-reference, pin-vs-reference and shipped-vs-pin of the same size => the residual is fp32 evaluation-order noise of '
                       'the reference\'s own arithmetic (the reference\'s fp32-vs-fp64 spread over 100-300 closed-loop steps, SURVEY App. D.5: '
                       'median 6e-6, worst 2.3e-4 .. 2.7e-3), amplified by the closed loop — not a restatement error'}


def planning_closed_report(engine):
    """PlanningEnv CLOSED LOOP (round 5): the reference's own PlanningEnv.step x 3 with a stored actor state_dict
    (tests/golden/planning_closed_kat.npz, tools/gen_golden.py::gen_planning_closed) against the engine running its own controller in
    the loop: HIP = PlanningEnv(controller=FusedActor) with the automatic schedule (the persistent kernel), oracle = Oracle + ActorOracle."""
    from neuralplane_amd.actor import pack_ppo_actor
    from tests.planning_closed import OracleClosedLoop, actor_state_dict, compare_with_reference
    g = np.load(os.path.join(GOLDEN, 'planning_closed_kat.npz'))
    w = pack_ppo_actor(actor_state_dict(g))
    rows = []
    if engine == 'hip':
        import torch
        from neuralplane_amd.actor import FusedActor
        from neuralplane_amd.envs.planning_env import PlanningEnv
        n = g['hi_actions'].shape[1]
        env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cuda:0', controller=FusedActor(w, 'cuda:0'))
    else:
        cl = OracleClosedLoop(g, w)
    for k in range(g['hi_actions'].shape[0]):
        if engine == 'hip':
            env._batch.reset(rand_u=g[f'rand_u_{k}'], want_obs=False)
            obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(g['hi_actions'][k]).cuda())
            res = {'s': env.model.s.cpu().numpy(), 'u': env.model.u.cpu().numpy(), 'tgt': env._batch.tgt.cpu().numpy().T.copy(),
                   'step_count': env.step_count.cpu().numpy(), 'rnn': env.ego_rnn_states.cpu().numpy()[:, 0], 'obs': obs.cpu().numpy(),
                   'reward': rew.cpu().numpy(), 'flags': np.stack([done.cpu().numpy(), bad.cpu().numpy(), tmo.cpu().numpy()]).astype(np.uint8)}
        else:
            res = cl.macro_step(k)
        e = compare_with_reference(res, g, k)
        rows.append(dict({'macro_step': k + 1, 'closed_loop_inner_steps_so_far': 50 * (k + 1), 'masks_and_counters_equal_to_reference': True,
                          'rows_bad_done_this_macro_step': int(g[f'flags_{k}'][1].sum()),
                          'rows_flying_since_the_start': int((g[f'step_count_{k}'] == 50 * (k + 1)).sum())}, **{'max_' + q: v for q, v in e.items()}))
    return {'fixture': 'tests/golden/planning_closed_kat.npz (reference PlanningEnv.step x 3 = 150 closed-loop inner steps, n = 80, actor state_dict stored)',
            'engine': 'hip: PlanningEnv(controller=FusedActor), automatic schedule' if engine == 'hip' else 'oracle: Oracle(tracking) + ActorOracle',
            'bounds': 'masks / counters equal; states 1e-4 (SURVEY floors), recurrent state 5e-5 (absolute), low-level actions 2e-5', 'at': rows}


def policy_report(engine):
    """The rollout policy's inference step: the reference's PPOPolicy.get_actions recorded over five chained calls
    (tests/golden/policy_kat.npz, tools/gen_golden.py::gen_policy; 4- and 3-action policies, the normal draws stored) against the engine
    fed the same inputs and draws, recurrent states chained on the engine's side; both numerics."""
    from neuralplane_amd.policy import pack_policy_actor, pack_policy_critic
    from tests.policy_kat import load
    rows = []
    for act_dim in (4, 3):
        g, sa, sc = load(GOLDEN, act_dim)
        for numerics in ('fp32', 'i8'):
            n = g['obs'].shape[1]
            if engine == 'hip':
                import torch
                from neuralplane_amd.policy import FusedPolicy
                fp = FusedPolicy((sa, sc), 'cuda:0', numerics=numerics)
                ha = hc = torch.zeros((n, 1, 128), device='cuda:0')
            else:
                from oracle.f16_oracle import PolicyOracle
                o = PolicyOracle(pack_policy_actor(sa)[0], pack_policy_critic(sc), g['std'], g['log_std'], numerics)
                ha = hc = np.zeros((n, 128), np.float32)
            worst = {'actions': 0.0, 'values': 0.0, 'log_probs': 0.0, 'rnn_states': 0.0}
            for t in range(g['obs'].shape[0]):
                if engine == 'hip':
                    v, a, lp, ha, hc = fp.get_actions(torch.from_numpy(g['obs'][t]).cuda(), ha, hc, torch.from_numpy(g['masks'][t]).cuda(),
                                                      noise=torch.from_numpy(g['eps'][t]).cuda())
                    vn, an, lpn, han, hcn = (x.cpu().numpy() for x in (v, a, lp, ha, hc))
                else:
                    vn, an, lpn, ha, hc = o.run(g['obs'][t], ha, hc, g['masks'][t], g['eps'][t])
                    han, hcn = ha, hc
                e = {'actions': np.abs(an - g['actions'][t]).max(), 'values': np.abs(vn.reshape(-1) - g['values'][t].reshape(-1)).max(),
                     'log_probs': np.abs(lpn.reshape(-1) - g['logp'][t].reshape(-1)).max(),
                     'rnn_states': max(np.abs(han.reshape(n, 128) - g['ha'][t][:, 0]).max(), np.abs(hcn.reshape(n, 128) - g['hc'][t][:, 0]).max())}
                worst = {k: max(worst[k], float(x)) for k, x in e.items()}
            rows.append(dict({'act_dim': act_dim, 'numerics': numerics, 'chained_calls': int(g['obs'].shape[0]), 'rows': int(n)}, **{'max_abs_err_' + k: x for k, x in worst.items()}))
    return {'fixture': 'tests/golden/policy_kat.npz (reference PPOPolicy.get_actions, five chained calls, sampled actions from stored normal draws)',
            'bounds': {'actions': 2e-5, 'values': 1e-4, 'log_probs': 5e-5, 'rnn_states': 5e-5}, 'cases': rows}


def combat_report(engine):
    """SingleCombat 1v1: the recorded env.steps of tests/golden/combat_kat.npz (the reference's components driven in
    the order of the stale singlecombat_env.step, plain ATen arithmetic), teacher-forced per env.step (5 FDM steps behind the
    high-gain PID stack: a last-bit difference grows ~x10 per FDM step, so free-running agreement is only meaningful bit-exactly —
    the pin-mode fixture, tests/test_combat_oracle_golden.py).  Per env.step: positions / attitude / speed / flow angles (states
    0..8) and the body rates P, Q, R (states 9..11) separately; shipped numerics and pin mode against the recording and against each
    other (same reading as the PlanningEnv rows)."""
    from oracle.f16_oracle import MODE_MLP_F64, CombatOracle
    d = np.load(os.path.join(GOLDEN, 'combat_kat.npz'))
    K, n = d['actions'].shape[:2]

    def run(mode):
        outs = []
        if engine == 'hip' and mode == 0:
            import torch
            from neuralplane_amd.core import F16CombatBatch
            from neuralplane_amd.envs.utils.utils import parse_config
            b = F16CombatBatch(n // 2, parse_config('selfplay'), 'cuda:0', seed=0)

            def load(s, u, pid, blood, sc, fl):
                b.s.copy_(torch.from_numpy(np.ascontiguousarray(s.T)))
                b.u.copy_(torch.from_numpy(np.ascontiguousarray(u.T)))
                if pid is not None:
                    b.pid.copy_(torch.from_numpy(np.ascontiguousarray(pid.T)))
                b.blood.copy_(torch.from_numpy(blood))
                b.step_count.copy_(torch.from_numpy(sc))
                b.flags.copy_(torch.from_numpy(np.ascontiguousarray(fl)))
            load(d['s_init'], d['u_init'], None, d['blood_init'], d['step_count_init'], np.zeros((3, n), np.uint8))
            for k in range(K):
                b.pid_first = (k == 0)
                obs, rew, flags = b.step(torch.from_numpy(d['actions'][k]).cuda(), rand_u=d['rand_u'][k])
                outs.append((b.s.cpu().numpy().T.copy(), obs.cpu().numpy(), rew.cpu().numpy(), flags.cpu().numpy()))
                load(d[f's_{k}'], d[f'u_{k}'], d[f'pid_{k}'], d[f'blood_{k}'], d[f'step_count_{k}'], d[f'flags_{k}'])
            return outs
        o = CombatOracle(mode=mode)
        st = o.new_state(n // 2)
        st['s'][:], st['u'][:], st['blood'][:], st['step_count'][:] = d['s_init'], d['u_init'], d['blood_init'], d['step_count_init']
        st['done'][:] = 0
        st['bad'][:] = 0
        st['timeout'][:] = 0
        for k in range(K):
            obs, rew, done, bad, tmo = o.combat_step(st, d['actions'][k], rand_u=d['rand_u'][k], pid_first=(k == 0))
            outs.append((st['s'].copy(), obs.copy(), rew.copy(), np.stack([done, bad, tmo])))
            st['s'][:], st['u'][:], st['pid'][:], st['blood'][:] = d[f's_{k}'], d[f'u_{k}'], d[f'pid_{k}'], d[f'blood_{k}']
            st['step_count'][:] = d[f'step_count_{k}']
            st['done'][:], st['bad'][:], st['timeout'][:] = d[f'flags_{k}']
        return outs

    plain, pin = run(0), run(MODE_MLP_F64)
    acc = {k: [] for k in ('slow_plain', 'rates_plain', 'slow_pin', 'rates_pin', 'slow_pp', 'rates_pp')}
    masks_equal, obs_max, rew_max = 0, 0.0, 0.0
    for k in range(K):
        ref = d[f's_{k}']
        e = lambda a, b, sl: np.nanmax(np.abs(a[:, sl] - b[:, sl]) / np.maximum(np.abs(b[:, sl]), STATE_FLOORS[sl]), axis=1)  # noqa: E731
        for tag, a, b in (('plain', plain[k][0], ref), ('pin', pin[k][0], ref), ('pp', plain[k][0], pin[k][0])):
            acc['slow_' + tag].append(e(a, b, slice(0, 9)))
            acc['rates_' + tag].append(e(a, b, slice(9, 12)))
        masks_equal += int(np.array_equal(plain[k][3].astype(bool), d[f'flags_{k}'].astype(bool)))
        obs_max = max(obs_max, float(np.max(np.abs(plain[k][1] - d[f'obs_{k}']))))
        rew_max = max(rew_max, float(np.max(np.abs(plain[k][2] - d[f'reward_{k}']))))
    cat = {k: np.concatenate(v) for k, v in acc.items()}
    return {'fixture': f'tests/golden/combat_kat.npz: {K} env.steps x {n} aircraft, teacher-forced per env.step (5 FDM steps)',
            'engine_for_shipped_numerics': 'hip' if engine == 'hip' else 'oracle', 'env_steps_with_all_masks_equal': masks_equal, 'env_steps': int(K),
            'states_0_8': {'shipped_vs_reference': _dist(cat['slow_plain']), 'pin_mode_vs_reference': _dist(cat['slow_pin']), 'shipped_vs_pin_mode': _dist(cat['slow_pp'])},
            'body_rates_PQR': {'shipped_vs_reference': _dist(cat['rates_plain']), 'pin_mode_vs_reference': _dist(cat['rates_pin']), 'shipped_vs_pin_mode': _dist(cat['rates_pp'])},
            'obs_max_abs': obs_max, 'reward_max_abs': rew_max,
            'reading': 'P, Q, R sit behind a rate PID of gain 573 deg per rad/s written straight to the surfaces: a 1e-7 difference in an MLP output is '
                       '~1e-3 in the rates after the 5 FDM steps of one env.step.  Shipped-vs-pin (two evaluations of the SAME restatement that differ '
                       'only in rounding) is as large as shipped-vs-reference: the 5e-3 acceptance on P, Q, R is the reference controller\'s own '
                       'amplification of fp32 noise; the env-level composition itself is ours (the reference env file cannot be constructed)'}


def build(engine):
    cls = HipEngine if engine == 'hip' else OracleEngine
    return {'engine': cls.name, 'reference': 'tests/golden/traj_*.npz: free-running trajectories of the imported reference (tools/gen_golden.py), '
                                             'reset draws injected, observation noise off',
            'metric': 'per aircraft max_k |x_k - ref_k| / max(|ref_k|, floor_k); aircraft that left the reference episode schedule excluded',
            'trajectories': [trajectory_report(cls, *t) for t in TRAJ],
            'trajectories_with_done_events': [trajectory_report(cls, *t, fixture=f'traj_pid_{t[0]}_N{t[1]}_T{t[2]}.npz') for t in PID_TRAJ],
            'closed_loop': closed_loop_report(cls), 'recorded_episode': recorded_episode_report(cls),
            'planning_env': planning_report(engine), 'planning_env_closed_loop': planning_closed_report(engine), 'single_combat': combat_report(engine),
            'rollout_policy': policy_report(engine)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--engine', default='hip', choices=['hip', 'oracle'])
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    rep = build(args.engine)
    txt = json.dumps(rep, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as f:
            f.write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
