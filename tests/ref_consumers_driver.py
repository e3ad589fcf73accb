"""Driver of tests/test_reference_consumers_cpu.py (test infrastructure; build container only — it imports /root/reference).

    python tests/ref_consumers_driver.py <side> <what> <out_dir>
        side: ref     the reference's own `envs` package (its PyTorch-CPU path)
              mirror  `envs` aliased to neuralplane_amd.envs as INTEGRATION.md §1 shows, with core.F16Batch swapped for the
                      oracle-backed stand-in tests/oracle_batch.py (tests only: the product keeps no CPU fallback)
        what: pid     algorithms/pid/controller.py — Controller.cal_pitch_throttle / update_* / stabilize(env) / get_action()
                      in the loop of renders/render_control.py, 200 steps from a pinned initial state
              render  renders/render_control.py itself (the script, executed unmodified but for its `device = "cuda:0"` line)
              runner  runner/F16sim_runner.py — two PPO iterations through GPUVecEnv (collect, insert, compute, train, save)

The reference's consumers are imported from /root/reference and run UNCHANGED; nothing of them is copied.  Both sides get the
same pinned initial state (the two env families draw their reset values from different generators), written through the public
`env.model.s` / `env.task.target_*` tensors.
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get('NP_REFERENCE_ROOT', '/root/reference')


def setup(side):
    sys.path[:0] = [os.path.join(REPO, 'tools', 'oracle_shims'), REF]
    if side == 'mirror':
        sys.path.insert(0, REPO)
        import neuralplane_amd.envs as npe
        import neuralplane_amd.envs.control_env
        import neuralplane_amd.envs.env_wrappers
        import neuralplane_amd.envs.env_base as eb
        import neuralplane_amd.envs.utils.utils
        from tests.oracle_batch import OracleBatch
        eb.F16Batch = OracleBatch                     # tests only: a CPU stand-in for the HIP batch
        sys.modules['envs'] = npe
        sys.modules['envs.control_env'] = npe.control_env
        sys.modules['envs.env_wrappers'] = npe.env_wrappers
        sys.modules['envs.utils'] = npe.utils
        sys.modules['envs.utils.utils'] = npe.utils.utils
    import numpy as np
    if not hasattr(np, 'product'):       # the reference targets numpy < 2 (algorithms/utils/flatten.py:83); this image has numpy 2.2
        np.product = np.prod
    import torch
    torch.set_num_threads(4)
    torch.manual_seed(0)
    return torch


def pin_state(env, torch):
    """the same flying state and targets on both sides, through the tensors the reference exposes"""
    n = env.n
    lin = torch.linspace(0.0, 1.0, n) if n > 1 else torch.zeros(1)
    s = env.model.s
    s[:, :] = 0.0
    s[:, 2] = 19000.0 + 1000.0 * lin
    s[:, 6] = 1000.0 + 200.0 * lin
    u = env.model.u
    u[:, :] = 0.0
    u[:, 0] = 2000.0
    env.task.target_altitude[:] = s[:, 2] + 1000.0
    env.task.target_heading[:] = 2.0943951
    env.task.target_vt[:] = s[:, 6]
    env.step_count[:] = 0


def run_pid(torch, out_dir, steps=200, n=6):
    import numpy as np
    from envs.control_env import ControlEnv
    from algorithms.pid.controller import Controller
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cpu')
    env.reset()
    pin_state(env, torch)
    controller = Controller(dt=env.model.dt, n=env.n, device='cpu')
    acts, states, rews = [], [], []
    flags = np.zeros(3, np.int64)
    for t in range(steps):
        hgt_dem = env.task.target_altitude.reshape(-1, 1)
        TAS_dem = env.task.target_vt.reshape(-1, 1)
        nav = env.task.target_heading.reshape(-1, 1)
        if t % 5 == 0:
            controller.cal_pitch_throttle(hgt_dem, TAS_dem, env)
            if t < 100:
                controller.update_level_flight(env)
            else:
                controller.update_heading_hold(nav, env)
        controller.stabilize(env)
        a = controller.get_action()
        acts.append(a.detach().numpy().copy())
        states.append(env.model.s.detach().numpy().copy())
        obs, rew, done, bad, tmo, info = env.step(a)
        rews.append(rew.detach().numpy().copy())
        flags += np.array([int(done.sum()), int(bad.sum()), int(tmo.sum())])
        assert obs.shape == (n, 22) and rew.shape == (n,) and done.dtype == torch.bool
    np.savez(os.path.join(out_dir, 'pid.npz'), actions=np.stack(acts), states=np.stack(states), rewards=np.stack(rews), flags=flags)


def run_render(torch, out_dir, cap=300):
    import numpy as np
    import envs.control_env as ce
    orig_reset, orig_step = ce.ControlEnv.reset, ce.ControlEnv.step
    counter = {'k': 0}

    def reset(self, *a, **k):
        obs = orig_reset(self, *a, **k)
        if not counter.get('pinned'):                 # the reference's BaseEnv.step() calls self.reset() every step: pin the FIRST reset only
            pin_state(self, torch)
            counter['pinned'] = True
        return obs

    def step(self, action, *a, **k):
        out = list(orig_step(self, action, *a, **k))
        counter['k'] += 1
        if counter['k'] >= cap:                       # the script loops until torch.any(dones): end it after `cap` steps
            out[2] = torch.ones_like(out[2])
        return tuple(out)
    ce.ControlEnv.reset, ce.ControlEnv.step = reset, step
    src = open(os.path.join(REF, 'renders', 'render_control.py')).read()
    assert src.count('device = "cuda:0"') == 1
    src = src.replace('device = "cuda:0"', 'device = "cpu"')
    os.chdir(out_dir)
    os.makedirs('result', exist_ok=True)
    os.makedirs('tracks', exist_ok=True)
    g = {'__name__': '__main__', '__file__': os.path.join(REF, 'renders', 'render_control.py')}
    exec(compile(src, g['__file__'], 'exec'), g)     # the reference's script, run as it is
    assert counter['k'] == cap
    files = sorted(f for f in os.listdir('result') if f.endswith('.npy'))
    np.savez(os.path.join(out_dir, 'render.npz'), **{f[:-4]: np.load(os.path.join('result', f)) for f in files})
    with open(os.path.join(out_dir, 'render_tracks.json'), 'w') as f:
        json.dump(sorted(os.listdir('tracks')), f)


class _Writer:
    def __init__(self):
        self.scalars = []

    def add_scalar(self, k, v, step):
        self.scalars.append((k, float(v), int(step)))


def run_runner(torch, out_dir, threads=12, buffer_size=8):
    import numpy as np
    from config import get_config
    from envs.control_env import ControlEnv
    from envs.env_wrappers import GPUVecEnv
    from runner.F16sim_runner import F16SimRunner
    parser = get_config()
    group = parser.add_argument_group('F16Sim Env parameters')          # scripts/train/train_F16sim.py:parse_args
    group.add_argument('--env-name', type=str, default='Control')
    group.add_argument('--scenario-name', type=str, default='heading')
    group.add_argument('--model-name', type=str, default='F16')
    argv = ['--env-name', 'Control', '--algorithm-name', 'ppo', '--scenario-name', 'heading', '--model-name', 'F16', '--experiment-name', 'v1',
            '--seed', '5', '--device', 'cpu', '--n-training-threads', '1', '--n-rollout-threads', str(threads), '--log-interval', '1',
            '--save-interval', '1', '--num-mini-batch', '2', '--buffer-size', str(buffer_size), '--num-env-steps', str(2 * threads * buffer_size),
            '--lr', '3e-4', '--gamma', '0.99', '--ppo-epoch', '2', '--clip-params', '0.2', '--max-grad-norm', '2', '--entropy-coef', '1e-3',
            '--hidden-size', '128 128', '--act-hidden-size', '128 128', '--recurrent-hidden-size', '128', '--recurrent-hidden-layers', '1',
            '--data-chunk-length', '4']
    all_args = parser.parse_known_args(argv)[0]
    envs = GPUVecEnv([lambda: ControlEnv(num_envs=all_args.n_rollout_threads, config=all_args.scenario_name, model=all_args.model_name,
                                         random_seed=all_args.seed, device=all_args.device)])
    writer = _Writer()
    cfg = {'all_args': all_args, 'envs': envs, 'eval_envs': None, 'device': torch.device('cpu'), 'run_dir': out_dir}
    runner = F16SimRunner(cfg, writer)
    runner.run()
    envs.close()
    b = runner.buffer
    shapes = {k: [list(getattr(b, k).shape), str(getattr(b, k).dtype)] for k in ('obs', 'actions', 'rewards', 'masks', 'bad_masks', 'returns', 'value_preds',
                                                                                'rnn_states_actor', 'rnn_states_critic', 'action_log_probs')}
    finite = bool(np.isfinite(b.obs).all() and np.isfinite(b.rewards).all() and np.isfinite(b.returns).all())
    saved = sorted(os.path.join(d, f) for d in os.listdir(out_dir) if d.startswith('episode_') for f in os.listdir(os.path.join(out_dir, d)))
    with open(os.path.join(out_dir, 'runner.json'), 'w') as f:
        json.dump({'shapes': shapes, 'finite': finite, 'total_num_steps': int(runner.total_num_steps), 'scalars': sorted({k for k, _, _ in writer.scalars}),
                   'saved': saved, 'num_agents': int(runner.num_agents), 'obs_space': list(runner.obs_space.shape), 'act_space': list(runner.act_space.shape)}, f)


def main():
    side, what, out_dir = sys.argv[1:4]
    torch = setup(side)
    {'pid': run_pid, 'render': run_render, 'runner': run_runner}[what](torch, out_dir)
    print('OK', side, what)


if __name__ == '__main__':
    main()
