#!/usr/bin/env python3
"""Drop-in demo: alias the reference's `envs` package to neuralplane_amd.envs, then run host code
written against the REFERENCE's import paths and env surface — a rollout loop with a small recurrent torch policy, first
through the numpy VecEnv contract the reference's runners use, then with tensors kept on the GPU.

    python examples/drop_in_rollout.py [num_envs] [steps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))   # run from a checkout without installing

import neuralplane_amd.envs as npe
import neuralplane_amd.envs.control_env
import neuralplane_amd.envs.env_wrappers
import neuralplane_amd.envs.utils.utils

sys.modules.setdefault('envs', npe)
sys.modules.setdefault('envs.control_env', npe.control_env)
sys.modules.setdefault('envs.env_wrappers', npe.env_wrappers)
sys.modules.setdefault('envs.utils', npe.utils)
sys.modules.setdefault('envs.utils.utils', npe.utils.utils)

# ---- from here on: code as one would write it against the reference (scripts/train/train_F16sim.py:20-37) ----
from envs.control_env import ControlEnv  # noqa: E402
from envs.env_wrappers import GPUVecEnv  # noqa: E402


class Policy(torch.nn.Module):
    def __init__(self, obs_dim, act_dim):
        super().__init__()
        self.enc = torch.nn.Sequential(torch.nn.LayerNorm(obs_dim), torch.nn.Linear(obs_dim, 128), torch.nn.ReLU())
        self.gru = torch.nn.GRUCell(128, 128)
        self.head = torch.nn.Linear(128, act_dim)

    def forward(self, obs, h):
        h = self.gru(self.enc(obs), h)
        return torch.tanh(self.head(h)), h


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000          # scripts/train_heading.sh: --n-rollout-threads 3000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = 'cuda:0'
    envs = GPUVecEnv([lambda: ControlEnv(num_envs=n, config='heading', model='F16', random_seed=1, device=dev)])
    policy = Policy(envs.observation_space.shape[0], envs.action_space.shape[0]).to(dev).eval()
    h = torch.zeros(n, 128, device=dev)
    obs = envs.reset()                                            # np [E, A, 22]
    ret = np.zeros((n, 1, 1), np.float32)
    done_count = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(steps):
            a, h = policy(torch.from_numpy(obs).to(dev).reshape(n, -1), h)
            obs, rew, done, bad, tmo, _ = envs.step(a.cpu().numpy().reshape(n, 1, -1))
            ret += rew
            ended = (done | bad | tmo).reshape(n)
            done_count += int(ended.sum())
            h[torch.from_numpy(ended).to(dev)] = 0
    dt = time.perf_counter() - t0
    print(f'numpy VecEnv loop: {n} envs x {steps} steps in {dt:.2f} s = {n * steps / dt:.3e} env-steps/s '
          f'(policy included), {done_count} episodes ended, mean return {float(ret.mean()):.3f}')
    print('termination statistics:', envs.env.termination_counts())

    # the same loop without leaving the GPU
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=1, device=dev)
    obs = env.reset()
    h = torch.zeros(n, 128, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(steps):
            a, h = policy(obs, h)
            obs, rew, done, bad, tmo, _ = env.step(a)
            h = h * (~(done | bad | tmo)).unsqueeze(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'device-resident loop: {n * steps / dt:.3e} env-steps/s (policy included)')

    # a PPO-style collection phase that never leaves the GPU: DeviceVecEnv -> DeviceReplayBuffer (the reference's ReplayBuffer
    # API on device tensors) -> compute_returns() as one kernel launch -> recurrent mini-batches gathered on the device
    from neuralplane_amd.buffer import DeviceReplayBuffer
    from neuralplane_amd.envs.env_wrappers import DeviceVecEnv

    class Args:
        buffer_size, n_rollout_threads, gamma, gae_lambda = min(steps, 100), n, 0.99, 0.95
        use_proper_time_limits, use_gae, recurrent_hidden_size, recurrent_hidden_layers = False, True, 128, 1

    venv = DeviceVecEnv([lambda: ControlEnv(num_envs=n, config='heading', model='F16', random_seed=1, device=dev)])
    buf = DeviceReplayBuffer(Args, 1, venv.observation_space, venv.action_space, device=dev)
    value = torch.nn.Linear(128, 1).to(dev)
    obs = venv.reset()                                            # torch [E, A, 22] on the device
    buf.obs[0].copy_(obs)
    h = torch.zeros(n, 128, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(Args.buffer_size):
            a, h = policy(obs.reshape(n, -1), h)
            v = value(h)
            obs, rew, done, bad, tmo, _ = venv.step(a.reshape(n, 1, -1))
            ended = (done | bad | tmo).reshape(n, 1)
            h = h * (~ended)
            masks = (~ended).float().reshape(n, 1, 1)
            buf.insert(obs, a.reshape(n, 1, -1), rew, masks, torch.zeros(n, 1, 1, device=dev), v.reshape(n, 1, 1), h.reshape(n, 1, 1, 128),
                       h.reshape(n, 1, 1, 128), bad_masks=(~bad).float().reshape(n, 1, 1))
        buf.compute_returns(value(h).reshape(n, 1, 1))
        nb = sum(batch[0].shape[0] for batch in DeviceReplayBuffer.recurrent_generator(buf, 5, 8))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    buf.after_update()
    print(f'device-resident collection: {Args.buffer_size} steps x {n} envs into DeviceReplayBuffer + returns + 5 mini-batches ({nb} rows) '
          f'in {dt * 1e3:.1f} ms = {n * Args.buffer_size / dt:.3e} env-steps/s; mean return-to-go {float(buf.returns[:-1].mean()):.3f}')


if __name__ == '__main__':
    main()
