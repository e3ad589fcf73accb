"""The aero coefficient surrogates on their own, through the C ABI (np_f16_aero_coefficients), on the data the reference checks
its own surrogates with: envs/models/F16/model/test_model.py evaluates the MLPs on the 630-point (alpha, beta, el) grid of
model/coefs.csv and scores them against the table-interpolated values (r2, mean absolute error).  tests/golden/model_grid_kat.npz
holds that grid, those table values and the imported reference's MLP outputs (tools/gen_golden.py::gen_model_grid).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.f16_oracle import MODE_PWL, Oracle  # noqa: E402  (the checker; test infrastructure)

DEAD = 24   # delta_Czq_lef: F16Dynamics.nlplant never reads it (F16_dynamics.py:199); not part of the device weights
LIVE = [k for k in range(43) if k != DEAD]


def _batch(tables=False):
    from neuralplane_amd.core import F16Batch
    from neuralplane_amd.envs.utils.utils import parse_config
    return F16Batch(64, parse_config('heading'), 'heading', 'cuda:0', seed=0, aero_1d_tables=tables)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _scores(coef, g):
    r2, mae = np.zeros(43), np.zeros(43)
    for k in range(43):
        m = int(g['npts'][k])
        y, f = g['table'][:m, k], coef[:m, k].astype(np.float64)
        r2[k] = 1.0 - np.sum((y - f) ** 2) / np.sum((y - y.mean()) ** 2)
        mae[k] = np.mean(np.abs(y - f))
    return r2, mae


@pytest.mark.parametrize('tables', [False, True], ids=['mlp', 'aero_1d_tables'])
def test_surrogates_on_the_reference_validation_grid(golden_dir, tables):
    g = np.load(f'{golden_dir}/model_grid_kat.npz')
    b = _batch(tables)
    c = b.aero_coefficients(g['alpha_deg'], g['beta_deg'], g['el']).cpu().numpy().T          # [630, 43]
    o = Oracle('heading', mode=MODE_PWL if tables else 0).aero(g['alpha_deg'], g['beta_deg'], g['el'])
    assert _same(c[:, LIVE], o[:, LIVE]), 'HIP coefficients differ from the oracle'
    assert np.all(c[:, DEAD] == 0.0)
    # against the reference's MLP outputs on its own grid
    import json
    man = json.load(open(os.path.join(os.path.dirname(golden_dir), '..', 'neuralplane_amd', 'assets', 'f16_aero_mlp.json')))
    std = np.array([n['out_std'] for n in man['nets']], np.float32)
    err = np.abs(c - g['coef']) / np.maximum(np.abs(g['coef']), std[None, :])
    assert err[:, LIVE].max() < (3e-5 if tables else 2e-5)
    # and the figure of merit of the reference's script: r2 / MAE against the table values, net by net
    r2, mae = _scores(c, g)
    assert np.abs(r2 - g['ref_r2'])[LIVE].max() < (1e-5 if tables else 1e-6)
    assert np.abs(mae - g['ref_mae'])[LIVE].max() < 1e-5 * max(1.0, float(g['ref_mae'].max()))
    assert r2[LIVE].min() > 0.96


def test_ragged_sizes_nonfinite_inputs_and_the_reference_shaped_object(golden_dir):
    from neuralplane_amd.envs.control_env import ControlEnv
    env = ControlEnv(num_envs=8, config='heading', model='F16', random_seed=0, device='cuda:0')
    hifi = env.model.hifi_F16
    rng = np.random.RandomState(5)
    o = Oracle('heading')
    for m in (1, 63, 129, 1000):
        a = rng.uniform(-25, 95, m).astype(np.float32)
        bb = rng.uniform(-35, 35, m).astype(np.float32)
        e = rng.uniform(-30, 30, m).astype(np.float32)
        if m > 2:
            a[1], bb[2] = np.nan, np.inf
        c = hifi.coefficients(a, bb, e).cpu().numpy().T
        assert _same(c[:, LIVE], o.aero(a, bb, e)[:, LIVE])
        if m > 2:
            assert np.all(np.isnan(c[1, LIVE])) and np.all(np.isnan(c[2, LIVE])) and np.all(np.isfinite(c[0]))
    # the seven group methods of the reference's object: same numbers, the reference's tuple sizes and shapes
    a = torch.tensor([[10.0, 20.0], [30.0, -5.0]], device='cuda:0')
    bt, e = torch.full_like(a, 3.0), torch.full_like(a, -4.0)
    full = hifi.coefficients(a, bt, e)
    groups = (hifi.hifi_C(a, bt, e), hifi.hifi_damping(a), hifi.hifi_C_lef(a, bt), hifi.hifi_damping_lef(a), hifi.hifi_rudder(a, bt),
              hifi.hifi_ailerons(a, bt), hifi.hifi_other_coeffs(a, e))
    assert [len(t) for t in groups] == [6, 9, 6, 9, 3, 6, 5]
    assert all(t.shape == a.shape for grp in groups for t in grp)
    # nets that do not take beta / el are unaffected by what the group call passes for them
    flat = [t for grp in groups for t in grp][:-1]          # the last entry is delta_Cm_ds == 0
    one_in = {6, 7, 8, 9, 10, 11, 12, 13, 14, 21, 22, 23, 25, 26, 27, 28, 29, 39, 40, 41}
    for k in sorted(one_in):
        assert torch.equal(flat[k].reshape(-1), full[k])
    assert torch.equal(groups[0][0].reshape(-1), full[0]) and torch.equal(groups[6][3].reshape(-1), full[42])
    assert torch.all(groups[6][4] == 0)
    with pytest.raises(ValueError):
        env._batch.aero_coefficients(a, bt[:1], e)


def test_dynamics_object_nlplant_and_atmos_for_arbitrary_states(golden_dir):
    """env.model.dynamics (F16_dynamics.py:10-228) outside the step: nlplant(x[m,17]) and atmos(alt, vt) for states that are not
    the batch's own — HIP == oracle bit for bit, and within 1e-4 of the reference's recorded values (nlplant_kat, getters_kat)."""
    from neuralplane_amd.envs.control_env import ControlEnv
    env = ControlEnv(num_envs=4, config='heading', model='F16', random_seed=0, device='cuda:0')
    dyn = env.model.dynamics
    g = np.load(f'{golden_dir}/nlplant_kat.npz')
    xd = dyn.nlplant(torch.from_numpy(g['x17'])).cpu().numpy()
    assert xd.shape == g['x17'].shape and np.all(xd[:, 12:] == 0)
    o = Oracle('heading')
    assert _same(xd[:, :12], o.nlplant(g['x17']))
    floors = np.array([10, 10, 10, 0.1, 0.1, 0.1, 1, 0.1, 0.1, 0.1, 0.1, 0.1], np.float32)
    err = np.abs(xd[:, :12] - g['xdot']) / np.maximum(np.abs(g['xdot']), floors)
    assert np.nanmax(err) < 1e-4
    assert torch.equal(dyn(0.0, torch.from_numpy(g['x17'])).cpu(), torch.from_numpy(xd))      # forward(t, x), as odeint calls it
    k = np.load(f'{golden_dir}/getters_kat.npz')
    alt, vt = torch.from_numpy(k['s'][:, 2].copy()), torch.from_numpy(k['s'][:, 6].copy())
    mach, qbar, ps = dyn.atmos(alt.reshape(16, 16), vt.reshape(16, 16))
    got = torch.stack([mach, qbar, ps], -1).reshape(-1, 3).cpu().numpy()
    assert mach.shape == (16, 16) and _same(got, o.get_atmos(k['s']))
    assert np.abs(got - k['atmos']).max() / 2000 < 1e-6
    with pytest.raises(ValueError):
        dyn.nlplant(torch.zeros(3, 12))
    assert dyn.hifi_F16 is env.model.hifi_F16
    from neuralplane_amd.envs.tasks.heading_task import HeadingTask
    assert isinstance(env.task, HeadingTask)
