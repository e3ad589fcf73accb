#!/usr/bin/env python3
"""Drop-in demo for the hierarchical tracking env: alias the reference's `envs` package to neuralplane_amd.envs and
drive `PlanningEnv` through `GPUVecEnv` the way scripts/train/train_F16sim.py:28-37 builds it — the only change on the host side is
`controller='fused'`, which makes the env load the low-level controller's checkpoint (the `actor_latest.pt` its own runner wrote,
envs/planning_env.py:16,43) into the fused kernel instead of a torch PPOActor.  That checkpoint is not part of the reference snapshot, so
this demo writes a PPOActor-shaped state_dict with seeded weights to a temporary file and points the env at it.

Second part: the collect step of the reference's runner (runner/F16sim_runner.py:123-154) kept on the device — the high-level policy's
`get_actions` as ONE launch (neuralplane_amd.policy.FusedPolicy over PPOPolicy-shaped networks with three actions), the env's macro-step as
ONE launch, the rollout storage's insert as ONE launch.

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
from neuralplane_amd.buffer import DeviceReplayBuffer
from neuralplane_amd.envs.env_wrappers import DeviceVecEnv
from neuralplane_amd.policy import FusedPolicy

sys.modules.setdefault('envs', npe)
sys.modules.setdefault('envs.planning_env', npe.planning_env)
sys.modules.setdefault('envs.env_wrappers', npe.env_wrappers)

# ---- from here on: code as one would write it against the reference ----
from envs.env_wrappers import GPUVecEnv  # noqa: E402
from envs.planning_env import PlanningEnv  # noqa: E402


def ppo_state_dicts(act_dim, seed):
    """(actor, critic) state_dicts with PPOActor's / PPOCritic's keys and shapes (algorithms/ppo/ppo_actor.py:14-36, ppo_critic.py:10-36 under the
    training scripts' arguments): small seeded weights, unit LayerNorm gains — networks that keep their outputs moderate, which is all a demo needs."""
    g = torch.Generator().manual_seed(seed)

    def lin(o, i, s=1.0):
        return torch.randn(o, i, generator=g) * (s / i ** 0.5), torch.zeros(o)

    def trunk(mlp):
        sd = {'base.feature_norm.weight': torch.ones(22), 'base.feature_norm.bias': torch.zeros(22)}
        for name, shape in (('base.mlp.fc.0', (128, 22)), ('base.mlp.fc.3', (128, 128)), (mlp + '.fc.0', (128, 128)), (mlp + '.fc.3', (128, 128))):
            sd[name + '.weight'], sd[name + '.bias'] = lin(*shape)
        for name in ('base.mlp.fc.2', 'base.mlp.fc.5', 'rnn.norm', mlp + '.fc.2', mlp + '.fc.5'):
            sd[name + '.weight'], sd[name + '.bias'] = torch.ones(128), torch.zeros(128)
        for k in ('ih', 'hh'):
            sd[f'rnn.gru.weight_{k}_l0'], sd[f'rnn.gru.bias_{k}_l0'] = lin(384, 128)
        return sd
    actor, critic = trunk('act.mlp'), trunk('mlp')
    actor['act.action_out.mu_net.fc.0.weight'], actor['act.action_out.mu_net.fc.0.bias'] = lin(act_dim, 128, 0.05)
    actor['act.action_out.log_std'] = torch.full((act_dim,), -0.5)
    critic['value_out.weight'], critic['value_out.bias'] = lin(1, 128)
    return actor, critic


class RolloutArgs:   # the fields ReplayBuffer reads from the reference's argument bag (algorithms/utils/buffer.py:27-52)
    gamma, use_proper_time_limits, use_gae, gae_lambda = 0.99, True, True, 0.95
    recurrent_hidden_size, recurrent_hidden_layers = 128, 1


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000          # scripts/train_tracking.sh: --n-rollout-threads 10000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = 'cuda:0'
    with tempfile.TemporaryDirectory() as tmp:
        ckpt = os.path.join(tmp, 'actor_latest.pt')
        torch.save(ppo_state_dicts(4, seed=7)[0], ckpt)           # the low-level controller: a control-task actor (22 observations, 4 commands)
        make = lambda: PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=1, device=dev, controller='fused', controller_checkpoint=ckpt)
        envs, denvs = GPUVecEnv([make]), DeviceVecEnv([make])
    rng = np.random.default_rng(0)
    obs = envs.reset()                                            # np [E, A, obs]
    ret, done_count = 0.0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        act = rng.uniform(-1, 1, (n, 1, envs.action_space.shape[0])).astype(np.float32)      # a high-level policy's (dpitch, dheading, dvt)
        obs, rew, done, bad, tmo, _ = envs.step(act)              # one macro-step = 50 x {controller forward, FDM step}
        ret += float(rew.sum())
        done_count += int(done.sum()) + int(bad.sum()) + int(tmo.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert np.isfinite(obs).all() and obs.shape[0] == n
    env = envs.env
    print(f'PlanningEnv x {n}: {steps} macro-steps ({steps * 50} controller + FDM iterations) in {dt * 1e3:.1f} ms = {dt / steps * 1e3:.3f} ms per '
          f'macro-step through the numpy VecEnv contract, {n * steps * 50 / dt:.3e} aircraft-steps/s; episodes ended {done_count}, mean reward '
          f'{ret / (n * steps):.4f}, controller numerics {env.controller.numerics}, fallbacks {env.loop_fallbacks}')

    # ---- the collect step on the device: policy.get_actions -> envs.step -> buffer.insert, three launches (+ the normal draws) ----
    policy = FusedPolicy(ppo_state_dicts(denvs.action_space.shape[0], seed=11), device=dev)
    RolloutArgs.buffer_size, RolloutArgs.n_rollout_threads = steps, n
    buf = DeviceReplayBuffer(RolloutArgs, 1, denvs.observation_space, denvs.action_space, device=dev)
    buf.obs[0].copy_(denvs.reset())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        values, actions, logp, ha, hc = policy.get_actions(buf.obs[s].reshape(n, -1), buf.rnn_states_actor[s].reshape(n, 128),
                                                           buf.rnn_states_critic[s].reshape(n, 128), buf.masks[s].reshape(n, 1))
        obs, rew, done, bad, tmo, _ = denvs.step(actions)
        buf.insert_step(obs, actions, rew, done, bad, tmo, logp, values, ha, hc)
    buf.compute_returns(policy.get_values(buf.obs[-1].reshape(n, -1), buf.rnn_states_critic[-1].reshape(n, 128), buf.masks[-1].reshape(n, 1)).reshape(n, 1, 1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(buf.returns).all() and torch.isfinite(buf.action_log_probs).all()
    print(f'device-resident collect loop: {steps} x (FusedPolicy.get_actions + PlanningEnv macro-step + insert) + returns in {dt * 1e3:.1f} ms = '
          f'{dt / steps * 1e3:.3f} ms per collect step, mean |action| {float(buf.actions.abs().mean()):.3f}, mean value {float(buf.value_preds[:-1].mean()):.3f}, '
          f'fallbacks {denvs.env.loop_fallbacks}')
    # ---- the same step with the buffer's addresses pre-bound: the policy writes into the slot in place (neuralplane_amd.collect) ----
    from neuralplane_amd.collect import DeviceCollector
    col = DeviceCollector(policy, denvs, buf)
    for _ in range(3):
        col.step()                                                # first use: allocations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        col.step()
    col.compute_returns()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(buf.returns).all()
    print(f'DeviceCollector: {dt / steps * 1e3:.3f} ms per collect step')
    print('OK')


if __name__ == '__main__':
    main()
