#!/usr/bin/env python3
"""Drop-in demo: alias the reference's `envs` package to neuralplane_amd.envs, then run host code
written against the REFERENCE's import paths and env surface — a rollout loop with a small recurrent torch policy, first
through the numpy VecEnv contract the reference's runners use, then with tensors kept on the GPU.

    python examples/drop_in_rollout.py [num_envs] [steps]
"""
import sys
import time

import numpy as np
import torch

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


if __name__ == '__main__':
    main()
