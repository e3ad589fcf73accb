"""The reference's REAL consumers, run unchanged against the Python mirror (build container only).

north_star's contract is "the repo's PPO/MAPPO runners drop in unchanged".  Here the reference's own files
  * algorithms/pid/controller.py (+ TECS / L1 / roll / pitch / yaw controllers: ~20 model getters per step),
  * renders/render_control.py (the script itself),
  * runner/F16sim_runner.py + base_runner.py (PPO: collect -> GPUVecEnv.step -> insert -> compute -> train -> save;
    algorithms/utils/{utils,act,buffer,flatten,mlp,gru}.py, algorithms/ppo/*)
are imported from /root/reference and run twice: on the reference's own `envs` (its PyTorch-CPU path) and on `envs` aliased to
`neuralplane_amd.envs` exactly as INTEGRATION.md §1 shows.  The build container has no GPU and the product has no CPU path, so on the
mirror side `core.F16Batch` is swapped for tests/oracle_batch.py (the parity oracle behind F16Batch's members — test
infrastructure only; the GPU suite holds the HIP kernels bit-exact to that oracle).  Each run is its own subprocess
(tests/ref_consumers_driver.py): module aliasing and sys.path edits stay out of the pytest process.

Nothing of the reference travels: the test skips where /root/reference is absent (the GPU box).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, 'ref_consumers_driver.py')
REF = os.environ.get('NP_REFERENCE_ROOT', '/root/reference')

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'runner')), reason='the reference checkout is only present in the build container')

# per-quantity floors of the relative error (positions 100 ft, angles / rates 0.1 rad, speeds 10 ft/s; controls in their units)
FLOORS = {'npos': 100, 'epos': 100, 'altitude': 100, 'roll': .1, 'pitch': .1, 'yaw': .1, 'vt': 10, 'alpha': .1, 'beta': .1, 'yaw_rate': .1, 'G': 1,
          'T': 100, 'throttle': .01, 'el': 1, 'ail': 1, 'rud': 1, 'roll_dem': .1, 'pitch_dem': .1, 'yaw_rate_dem': .1, 'target_altitude': 100,
          'target_heading': .1, 'target_vt': 10}
STATE_FLOORS = np.array([100, 100, 100, .1, .1, .1, 10, .1, .1, .1, .1, .1], np.float32)


def _run(side, what, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    r = subprocess.run([sys.executable, DRIVER, side, what, str(out_dir)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, f'{side} {what}:\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}'
    return out_dir


def test_pid_controller_stack_drives_the_mirror_like_the_reference(tmp_path):
    """Controller.cal_pitch_throttle / update_level_flight / update_heading_hold / stabilize(env) / get_action() for 200 steps from one
    pinned state on both env families: the actions the reference's controller computes FROM THE MIRROR'S GETTERS (get_position,
    get_posture, get_TAS, get_EAS2TAS, get_euler_angular_velocity, get_climb_rate, get_acceleration, get_ground_speed) and the states
    they produce follow the reference's.  Level flight (100 steps): 5e-5; after the heading-hold switch the high-gain roll loop amplifies
    fp32 rounding (the two sides round sin / cos / the MLP sums differently) to a few 1e-3 at the peak of the transient, decaying again."""
    a = np.load(os.path.join(_run('ref', 'pid', tmp_path / 'ref'), 'pid.npz'))
    b = np.load(os.path.join(_run('mirror', 'pid', tmp_path / 'mirror'), 'pid.npz'))
    assert a['actions'].shape == b['actions'].shape == (200, 6, 4)
    assert np.array_equal(a['flags'], b['flags'])            # nobody terminated on either side
    err_s = np.abs(a['states'] - b['states']) / np.maximum(np.abs(a['states']), STATE_FLOORS)
    err_a = np.abs(a['actions'] - b['actions']) / np.maximum(np.abs(a['actions']), 0.1)
    assert err_s[:100].max() < 5e-5 and err_a[:100].max() < 2e-4, (err_s[:100].max(), err_a[:100].max())
    assert err_s.max() < 2e-2 and err_a[-1].max() < 1e-2 and err_s[-1].max() < 2e-3, (err_s.max(), err_a[-1].max(), err_s[-1].max())
    assert np.abs(a['rewards'] - b['rewards']).max() < 1e-4


def test_render_control_script_runs_on_the_mirror(tmp_path):
    """renders/render_control.py executed as it is (its `device = "cuda:0"` line set to "cpu", the loop ended after 300 steps through the
    `dones` it polls): every attribute it reads exists on the mirror — env.model.dt, env.n, get_position / get_posture /
    get_extended_state()[:, 5] / get_vt / get_AOA / get_AOS / get_G / get_thrust / get_control_surface / get_state /
    get_acceleration / get_EAS2TAS, env.task.target_altitude / target_heading / target_vt, env.step(render=True, count=k) writing the
    TacView track — and the 22 series it saves agree with the run on the reference's env (<= 1e-4 with the SURVEY floors).  That includes
    the reference's own quirk: TECS keeps `altitude.reshape(-1, 1)` of its first call and rate-limits it IN PLACE (TECS.py:194), which
    writes through the getter's view into the live state (+83.3 ft on the first call) — on both env families."""
    a = np.load(os.path.join(_run('ref', 'render', tmp_path / 'ref'), 'render.npz'))
    b = np.load(os.path.join(_run('mirror', 'render', tmp_path / 'mirror'), 'render.npz'))
    assert sorted(a.files) == sorted(b.files) == sorted(FLOORS)
    for k in a.files:
        assert a[k].shape == b[k].shape == (301,), k
        err = np.abs(a[k] - b[k]) / np.maximum(np.abs(a[k]), FLOORS[k])
        assert err.max() < 1e-4, (k, float(err.max()), int(err.argmax()))
    assert float(a['altitude'][1] - a['altitude'][0]) == pytest.approx(83.33, abs=0.5)      # the in-place quirk, reproduced
    ta, tb = (json.load(open(os.path.join(tmp_path, s, 'render_tracks.json'))) for s in ('ref', 'mirror'))
    assert ta == tb == ['F16SimRecording-0.txt.acmi']


def test_ppo_runner_trains_two_iterations_through_gpuvecenv(tmp_path):
    """runner/F16sim_runner.py::F16SimRunner built from config.get_config() and run for two PPO iterations (12 envs x 8 steps each) on
    GPUVecEnv(ControlEnv): observation / action spaces pass `isinstance(space, gym.spaces.Box)` (algorithms/utils/utils.py:16-21,
    act.py:24-27), the numpy [E, A, .] step / reset contract feeds ReplayBuffer.insert (buffer.py:37-75), compute_returns and the PPO
    update run, two checkpoints are saved.  The mirror side ends with the same buffer shapes / dtypes, logged scalars and files as the
    reference side (the numbers differ: the two env families draw their resets and noise from different generators)."""
    ra = json.load(open(os.path.join(_run('ref', 'runner', tmp_path / 'ref'), 'runner.json')))
    rb = json.load(open(os.path.join(_run('mirror', 'runner', tmp_path / 'mirror'), 'runner.json')))
    assert rb['finite'] and ra['finite']
    assert rb == ra, {k: (ra[k], rb[k]) for k in ra if ra[k] != rb.get(k)}
    assert rb['total_num_steps'] == 192 and rb['num_agents'] == 1 and rb['obs_space'] == [22] and rb['act_space'] == [4]
    assert rb['shapes']['obs'] == [[9, 12, 1, 22], 'float32'] and len(rb['saved']) == 4
