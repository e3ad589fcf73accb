"""Shared by the CPU and GPU policy tests: tests/golden/policy_kat.npz (tools/gen_golden.py gen_policy — the reference's PPOPolicy.get_actions
recorded over chained steps for the heading (4 actions) and tracking (3 actions) policies, with the normal draws behind every sample)."""
import numpy as np

# bounds against the reference's recording (fp32 ATen GEMMs / vectorised kernels there, ordered fmaf chains here)
TOL = {'actions': 2e-5, 'means': 2e-5, 'values': 1e-4, 'logp': 5e-5, 'rnn': 5e-5}


def load(golden_dir, act_dim, obs_dim=22):
    d = np.load(f'{golden_dir}/policy_kat.npz')
    pre = f'a{act_dim}::' if obs_dim == 22 else f'a{act_dim}o{obs_dim}::'
    g = {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}
    sa = {k[len('actor::'):]: v for k, v in g.items() if k.startswith('actor::')}
    sc = {k[len('critic::'):]: v for k, v in g.items() if k.startswith('critic::')}
    return g, sa, sc


def check_step(g, t, values, actions, logp, ha, hc):
    """one recorded get_actions call against results computed from the recorded inputs of the same call"""
    err = {'values': np.max(np.abs(values.reshape(-1) - g['values'][t].reshape(-1))), 'actions': np.max(np.abs(actions - g['actions'][t])),
           'logp': np.max(np.abs(logp.reshape(-1) - g['logp'][t].reshape(-1))),
           'rnn': max(np.max(np.abs(ha.reshape(-1, 128) - g['ha'][t][:, 0])), np.max(np.abs(hc.reshape(-1, 128) - g['hc'][t][:, 0])))}
    for k, v in err.items():
        assert v < TOL[k], (t, k, float(v))
    return err


def random_state_dicts(act_dim, seed, scale=1.0, obs_dim=22):
    """(actor, critic) numpy state_dicts with PPOActor's / PPOCritic's keys and shapes, every parameter random (LayerNorm terms included)."""
    rng = np.random.RandomState(seed)

    def trunk(mlp):
        sd = {'base.feature_norm.weight': 1 + 0.3 * rng.normal(size=obs_dim), 'base.feature_norm.bias': 0.2 * rng.normal(size=obs_dim)}
        for name, (o, i) in (('base.mlp.fc.0', (128, obs_dim)), ('base.mlp.fc.3', (128, 128)), (mlp + '.fc.0', (128, 128)), (mlp + '.fc.3', (128, 128))):
            sd[name + '.weight'], sd[name + '.bias'] = scale * rng.normal(size=(o, i)) / np.sqrt(i), 0.1 * rng.normal(size=o)
        for name in ('base.mlp.fc.2', 'base.mlp.fc.5', 'rnn.norm', mlp + '.fc.2', mlp + '.fc.5'):
            sd[name + '.weight'], sd[name + '.bias'] = 1 + 0.3 * rng.normal(size=128), 0.2 * rng.normal(size=128)
        for k in ('ih', 'hh'):
            sd[f'rnn.gru.weight_{k}_l0'], sd[f'rnn.gru.bias_{k}_l0'] = scale * rng.normal(size=(384, 128)) / np.sqrt(128), 0.1 * rng.normal(size=384)
        return sd
    actor, critic = trunk('act.mlp'), trunk('mlp')
    actor['act.action_out.mu_net.fc.0.weight'], actor['act.action_out.mu_net.fc.0.bias'] = 0.2 * rng.normal(size=(act_dim, 128)), 0.1 * rng.normal(size=act_dim)
    actor['act.action_out.log_std'] = rng.uniform(-2.0, 0.5, act_dim)
    critic['value_out.weight'], critic['value_out.bias'] = 0.5 * rng.normal(size=(1, 128)), 0.3 * rng.normal(size=1)
    return ({k: np.asarray(v, np.float32) for k, v in actor.items()}, {k: np.asarray(v, np.float32) for k, v in critic.items()})


LONG_N, LONG_STEPS, LONG_EVERY = 48, 200, 10


def long_inputs(n=LONG_N, steps=LONG_STEPS, obs_dim=22, seed=4242):
    """Inputs of tests/golden/policy_long_kat.npz (tools/gen_golden.py::gen_policy_long), regenerated instead of stored (numpy RandomState is
    version-stable): observations that move like observations do — an AR(1) walk per row and column around per-column scales —
    and masks with episode ends (2 % per row and step; `masks[t] = 0` makes get_actions restart that row's recurrent state,
    algorithms/utils/gru.py; runner/F16sim_runner.py:131-154 builds them from done | bad_done | exceed_time_limit)."""
    rng = np.random.RandomState(seed)
    scale = rng.uniform(0.1, 3.0, (1, obs_dim))
    x = rng.normal(0, 1, (n, obs_dim))
    obs = np.empty((steps, n, obs_dim), np.float32)
    for t in range(steps):
        x = 0.97 * x + np.sqrt(1 - 0.97 ** 2) * rng.normal(0, 1, (n, obs_dim))
        obs[t] = (x * scale).astype(np.float32)
    masks = (rng.uniform(0, 1, (steps, n, 1)) > 0.02).astype(np.float32)
    masks[0] = 1.0
    return obs, masks


def check_long_chain(g, run, what):
    """`run(obs, ha, hc, masks, eps) -> values, actions, logp, ha, hc` chained for 200 steps on its OWN recurrent states against the
    reference's recording (every LONG_EVERY-th step): the bounds of the 5-step fixture, at 200 steps.  Returns the worst errors."""
    obs, masks = long_inputs()
    n = obs.shape[1]
    ha = hc = np.zeros((n, 128), np.float32)
    worst = {k: 0.0 for k in TOL if k != 'means'}
    rec = 0
    for t in range(obs.shape[0]):
        v, a, lp, ha, hc = run(obs[t], ha, hc, masks[t], g['eps'][t])
        ha, hc = np.asarray(ha, np.float32).reshape(n, 128), np.asarray(hc, np.float32).reshape(n, 128)
        if (t + 1) % LONG_EVERY == 0:
            e = {'values': np.max(np.abs(np.asarray(v).reshape(-1) - g['values'][rec].reshape(-1))), 'actions': np.max(np.abs(np.asarray(a) - g['actions'][rec])),
                 'logp': np.max(np.abs(np.asarray(lp).reshape(-1) - g['logp'][rec].reshape(-1))),
                 'rnn': max(np.max(np.abs(ha - g['ha'][rec])), np.max(np.abs(hc - g['hc'][rec])))}
            for k, x in e.items():
                assert x < TOL[k], (what, t + 1, k, float(x))
                worst[k] = max(worst[k], float(x))
            rec += 1
    assert rec == g['values'].shape[0] == LONG_STEPS // LONG_EVERY and int((masks == 0).sum()) > 50
    return worst


def load_long(golden_dir):
    d = np.load(f'{golden_dir}/policy_long_kat.npz')
    g = {k: d[k] for k in d.files if '::' not in k}
    sa = {k[len('actor::'):]: d[k] for k in d.files if k.startswith('actor::')}
    sc = {k[len('critic::'):]: d[k] for k in d.files if k.startswith('critic::')}
    return g, sa, sc
