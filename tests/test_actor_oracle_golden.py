"""CPU tests: the oracle's restatement of PlanningEnv's frozen low-level controller (oracle/f16_actor.inc) against the
reference's own PPOActor (tests/golden/actor_kat.npz, tools/gen_golden.py gen_actor: seeded random init, deterministic mode,
4 consecutive calls with the recurrent state carried, some masks zeroed).  The reference evaluates the layers with ATen GEMMs
and vectorised LayerNorm / GRU kernels, the spec with ordered fmaf chains: agreement is a tolerance, stated below."""
import numpy as np

from neuralplane_amd.actor import NUM_FLOATS, pack_ppo_actor
from oracle.f16_oracle import ActorOracle


def _sd(d):
    return {k[4:]: d[k] for k in d.files if k.startswith('sd::')}


def test_packed_layout_and_architecture_check(golden_dir):
    d = np.load(f'{golden_dir}/actor_kat.npz')
    sd = _sd(d)
    w = pack_ppo_actor(sd)
    assert w.dtype == np.float32 and w.size == NUM_FLOATS == ActorOracle(w).w.size
    # Linear weights are stored k-major: element [k][j] of the packed block is W[j][k]
    assert w[44 + 128 + 5 * 128 + 7] == sd['base.mlp.fc.0.weight'][7, 5]
    bad = dict(sd)
    bad['base.mlp.fc.3.weight'] = np.zeros((64, 128), np.float32)
    try:
        pack_ppo_actor(bad)
        raise AssertionError('a different architecture must be rejected')
    except ValueError:
        pass


def test_actor_forward_matches_reference_ppo_actor(golden_dir):
    d = np.load(f'{golden_dir}/actor_kat.npz')
    o = ActorOracle(pack_ppo_actor(_sd(d)))
    steps, n = d['obs'].shape[:2]
    # teacher-forced: every call starts from the reference's recurrent state
    h = np.zeros((n, 128), np.float32)
    for t in range(steps):
        act, h_out = o.forward(d['obs'][t], h, d['masks'][t])
        assert np.max(np.abs(act - d['actions'][t])) < 2e-5, t          # actions in (-1, 1)
        assert np.max(np.abs(h_out - d['rnn'][t][:, 0])) < 2e-5, t       # GRU state in (-1, 1)
        h = d['rnn'][t][:, 0]
    # free-running over the 4 calls (the recurrent loop is contractive: no amplification)
    h = np.zeros((n, 128), np.float32)
    for t in range(steps):
        act, h = o.forward(d['obs'][t], h, d['masks'][t])
    assert np.max(np.abs(act - d['actions'][-1])) < 5e-5 and np.max(np.abs(h - d['rnn'][-1][:, 0])) < 5e-5
    assert np.abs(d['actions']).max() > 0.5     # the fixture exercises the head's tanh away from 0
