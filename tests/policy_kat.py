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
