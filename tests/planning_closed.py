"""Shared by the CPU and GPU suites (test infrastructure): PlanningEnv closed loop through the ORACLE — Oracle('tracking') reset /
low_level_obs / inner step + ActorOracle, the recurrent state feeding back — on the inputs of tests/golden/planning_closed_kat.npz
(the reference's own PlanningEnv.step with a stored actor state_dict, tools/gen_golden.py::gen_planning_closed; reference
envs/planning_env.py:144-177, algorithms/ppo/ppo_actor.py:38-64)."""
import numpy as np

from oracle.f16_oracle import ActorOracle, Oracle

STATE_FLOORS = np.array([100, 100, 100, .1, .1, .1, 10, .1, .1, .1, .1, .1], np.float32)
# controls: thrust in lbf (0 ... 5.6e4), surfaces in degrees (+-45): u' = 0.9 u + 4.5 a per step (F16_model.py:52-62), so the action bound
# 2e-5 is 9e-4 degrees — the scale floors are a tenth of the surfaces' working range, like the 0.1 rad of the angles
U_FLOORS = np.array([1000, 10, 10, 10, 10], np.float32)
INNER = 50


def actor_state_dict(g):
    return {k[4:]: g[k] for k in g.files if k.startswith('sd::')}


def planning_targets(s, hi_action):
    """planning_env.py:146-152 in fp32: clamp, then pitch + a0*0.3, yaw + a1*0.3, vt + a2*30."""
    a = np.clip(hi_action, -1, 1).astype(np.float32)
    return np.stack([s[:, 4] + a[:, 0] * np.float32(0.3), s[:, 5] + a[:, 1] * np.float32(0.3),
                     s[:, 6] + a[:, 2] * np.float32(30)], 1).astype(np.float32)


def relerr(a, ref, floor):
    return float(np.nanmax(np.abs(a - ref) / np.maximum(np.abs(ref), floor)))


class OracleClosedLoop:
    """macro_step(k) -> dict of everything PlanningEnv.step leaves behind; per-inner-iteration actions / observations are kept
    for the comparison with the fixture's recording."""

    def __init__(self, g, packed_weights, numerics='fp32'):
        self.g = g
        self.n = g['hi_actions'].shape[1]
        self.o = Oracle('tracking')
        self.actor = ActorOracle(packed_weights, numerics)
        self.st = Oracle.new_state(self.n)
        self.h = np.zeros((self.n, 128), np.float32)
        self.ones = np.ones(self.n, np.float32)

    def macro_step(self, k):
        g, o, st = self.g, self.o, self.st
        o.reset(st, rand_u=g[f'rand_u_{k}'], want_obs=False)
        tgt3 = planning_targets(st['s'], g['hi_actions'][k])
        ll_obs, ll_act, ll_rnn = [], [], []
        for i in range(INNER):
            ll = o.lowlevel_obs(st, tgt3)
            a, self.h = self.actor.forward(ll, self.h, self.ones)
            obs, rew, d, b, t = o.step_inner(st, a)
            ll_obs.append(ll)
            ll_act.append(a)
            if i % 10 == 9:
                ll_rnn.append(self.h.copy())
        return {'s': st['s'].copy(), 'u': st['u'].copy(), 'tgt': st['tgt'].copy(), 'step_count': st['step_count'].copy(), 'rnn': self.h.copy(),
                'obs': obs, 'reward': rew, 'flags': np.stack([d, b, t]).astype(np.uint8), 'll_obs': np.stack(ll_obs), 'll_act': np.stack(ll_act),
                'll_rnn': np.stack(ll_rnn)}


def compare_with_reference(res, g, k):
    """Closed-loop result of macro-step k (oracle or HIP) against the reference's recording: masks and counters equal, states within
    BASELINE's 1e-4 with the SURVEY §8(d) floors, recurrent state within 5e-5 (absolute; |h| < 1), low-level actions within 2e-5.
    Returns the measured errors."""
    fl = g[f'flags_{k}']
    assert np.array_equal(res['flags'], fl), f'macro-step {k}: masks differ from the reference at rows {np.nonzero((res["flags"] != fl).any(0))[0]}'
    assert np.array_equal(res['step_count'], g[f'step_count_{k}']), f'macro-step {k}: step counters'
    e = {'state': relerr(res['s'], g[f's_{k}'], STATE_FLOORS), 'u': relerr(res['u'], g[f'u_{k}'], U_FLOORS), 'tgt': relerr(res['tgt'], g[f'tgt_{k}'], 1.0),
         'rnn': float(np.max(np.abs(res['rnn'] - g[f'rnn_{k}']))), 'obs': relerr(res['obs'], g[f'obs_{k}'], 0.1),
         'reward': relerr(res['reward'], g[f'reward_{k}'], 1.0)}
    if f'll_act_last_{k}' in g.files and 'll_act' in res:      # the long fixture keeps the 50th controller call of every macro-step only
        e['ll_act_last'] = float(np.max(np.abs(res['ll_act'][-1] - g[f'll_act_last_{k}'])))
        assert e['ll_act_last'] < 2e-5, (k, e)
    elif 'll_act' in res:
        e['ll_act'] = float(np.max(np.abs(res['ll_act'] - g[f'll_act_{k}'])))
        e['ll_rnn'] = float(np.max(np.abs(res['ll_rnn'] - g[f'll_rnn_{k}'])))
        e['ll_obs'] = relerr(res['ll_obs'][g['ll_obs_at']], g[f'll_obs_{k}'], 0.1)
    assert e['state'] < 1e-4, (k, e)
    assert e['rnn'] < 5e-5, (k, e)
    assert e['obs'] < 1e-4 and e['reward'] < 1e-5 and e['u'] < 1e-4 and e['tgt'] < 1e-6, (k, e)
    if 'll_act' in e:
        assert e['ll_act'] < 2e-5 and e['ll_rnn'] < 5e-5 and e['ll_obs'] < 1e-4, (k, e)
    return e
