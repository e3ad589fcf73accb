#!/usr/bin/env python3
"""Drop-in demo for the hierarchical tracking env: alias the reference's `envs` package to neuralplane_amd.envs and
drive `PlanningEnv` through `GPUVecEnv` the way scripts/train/train_F16sim.py:28-37 builds it — the only change on the host side is
`controller='fused'`, which makes the env load the low-level controller's checkpoint (the `actor_latest.pt` its own runner wrote,
envs/planning_env.py:16,43) into the fused kernel instead of a torch PPOActor.  That checkpoint is not part of the reference snapshot, so
this demo writes a PPOActor-shaped state_dict with seeded weights to a temporary file and points the env at it.

    python examples/drop_in_planning.py [num_envs] [macro_steps]
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))   # run from a checkout without installing

import neuralplane_amd.envs as npe
import neuralplane_amd.envs.env_wrappers
import neuralplane_amd.envs.planning_env

sys.modules.setdefault('envs', npe)
sys.modules.setdefault('envs.planning_env', npe.planning_env)
sys.modules.setdefault('envs.env_wrappers', npe.env_wrappers)

# ---- from here on: code as one would write it against the reference ----
from envs.env_wrappers import GPUVecEnv  # noqa: E402
from envs.planning_env import PlanningEnv  # noqa: E402


def controller_checkpoint(path, seed=7):
    """A state_dict with PPOActor's keys and shapes (algorithms/ppo/ppo_actor.py:14-36 under planning_env.py:18-29's arguments): small seeded
    weights, unit LayerNorm gains — a controller that keeps its commands near trim, which is all a demo needs."""
    g = torch.Generator().manual_seed(seed)

    def lin(o, i, s):
        return torch.randn(o, i, generator=g) * (s / i ** 0.5), torch.zeros(o)

    sd = {'base.feature_norm.weight': torch.ones(22), 'base.feature_norm.bias': torch.zeros(22)}
    for name, shape, s in (('base.mlp.fc.0', (128, 22), 1.0), ('base.mlp.fc.3', (128, 128), 1.0), ('act.mlp.fc.0', (128, 128), 1.0),
                           ('act.mlp.fc.3', (128, 128), 1.0), ('act.action_out.mu_net.fc.0', (4, 128), 0.05)):
        sd[name + '.weight'], sd[name + '.bias'] = lin(*shape, s)
    for name in ('base.mlp.fc.2', 'base.mlp.fc.5', 'rnn.norm', 'act.mlp.fc.2', 'act.mlp.fc.5'):
        sd[name + '.weight'], sd[name + '.bias'] = torch.ones(128), torch.zeros(128)
    for k in ('ih', 'hh'):
        sd[f'rnn.gru.weight_{k}_l0'], sd[f'rnn.gru.bias_{k}_l0'] = lin(384, 128, 1.0)
    sd['act.action_out.logstd._bias'] = torch.zeros(4, 1)     # present in the reference's file, unused by a deterministic forward
    torch.save(sd, path)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000          # scripts/train_tracking.sh: --n-rollout-threads 10000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = 'cuda:0'
    with tempfile.TemporaryDirectory() as tmp:
        ckpt = os.path.join(tmp, 'actor_latest.pt')
        controller_checkpoint(ckpt)
        envs = GPUVecEnv([lambda: PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=1, device=dev, controller='fused',
                                              controller_checkpoint=ckpt)])
    rng = np.random.default_rng(0)
    obs = envs.reset()                                            # np [E, A, obs]
    ret, done_count = 0.0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        act = rng.uniform(-1, 1, (n, 1, envs.action_space.shape[0])).astype(np.float32)      # the high-level policy's (dpitch, dheading, dvt)
        obs, rew, done, bad, tmo, _ = envs.step(act)              # one macro-step = 50 x {controller forward, FDM step}
        ret += float(rew.sum())
        done_count += int(done.sum()) + int(bad.sum()) + int(tmo.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert np.isfinite(obs).all() and obs.shape[0] == n
    env = envs.env if hasattr(envs, 'env') else None
    print(f'PlanningEnv x {n}: {steps} macro-steps ({steps * 50} controller + FDM iterations) in {dt * 1e3:.1f} ms = {dt / steps * 1e3:.3f} ms per '
          f'macro-step through the numpy VecEnv contract, {n * steps * 50 / dt:.3e} aircraft-steps/s; episodes ended {done_count}, mean reward '
          f'{ret / (n * steps):.4f}' + (f', controller numerics {env.controller.numerics}, fallbacks {env.loop_fallbacks}' if env is not None else ''))
    print('OK')


if __name__ == '__main__':
    main()
