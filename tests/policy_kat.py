"""Shared by the CPU and GPU policy tests: tests/golden/policy_kat.npz (tools/gen_golden.py gen_policy — the reference's PPOPolicy.get_actions
recorded over chained steps for the heading (4 actions) and tracking (3 actions) policies, with the normal draws behind every sample)."""
import numpy as np

# bounds against the reference's recording (fp32 ATen GEMMs / vectorised kernels there, ordered fmaf chains here)
TOL = {'actions': 2e-5, 'means': 2e-5, 'values': 1e-4, 'logp': 5e-5, 'rnn': 5e-5}


def load(golden_dir, act_dim):
    d = np.load(f'{golden_dir}/policy_kat.npz')
    pre = f'a{act_dim}::'
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
