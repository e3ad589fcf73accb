"""GPUVecEnv — numpy <-> torch adapter with the reference's contract (envs/env_wrappers.py:84-123):
`reset() -> np[E,A,obs]`, `step(np[E,A,act]) -> (np[E,A,obs], np[E,A,1] x4, info)`."""
import numpy as np
import torch


def _t2n(x):
    return x.detach().cpu().numpy()


class GPUVecEnv:
    def __init__(self, env_fns):
        assert len(env_fns) == 1, 'GPUVecEnv wraps exactly one batched env'
        self.env = env_fns[0]()
        self.num_envs = self.env.num_envs
        self.agents = self.num_agents = self.env.num_agents
        self.n = self.env.n
        self.device = self.env.device
        self.observation_space = self.env.observation_space
        self.action_space = self.env.action_space

    def _shape(self, x, k):
        return x.reshape(self.num_envs, self.num_agents, k)

    def reset(self):
        obs = self.env.reset()
        return _t2n(self._shape(obs, obs.shape[-1]))

    def step(self, actions):
        a = torch.as_tensor(np.asarray(actions), dtype=torch.float32, device=self.device).reshape(self.n, -1)
        obs, reward, done, bad_done, exceed_time_limit, info = self.env.step(a)
        return (_t2n(self._shape(obs, obs.shape[-1])), _t2n(self._shape(reward, 1)), _t2n(self._shape(done, 1)),
                _t2n(self._shape(bad_done, 1)), _t2n(self._shape(exceed_time_limit, 1)), info)

    def close(self):
        pass


class DeviceVecEnv(GPUVecEnv):
    """GPUVecEnv with the same `[E, A, ...]` shapes but torch tensors that never leave the GPU
    (SURVEY.md §8f N1): GPUVecEnv pays 1 H2D + 5 D2H copies per step (~126 B per aircraft, ~2 ms at
    N = 1e6 over PCIe Gen5 — five times the fused kernel); a device-resident policy / rollout buffer
    does not need them."""

    def reset(self):
        obs = self.env.reset()
        return self._shape(obs, obs.shape[-1])

    def step(self, actions):
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device).reshape(self.n, -1)
        obs, reward, done, bad_done, exceed_time_limit, info = self.env.step(a)
        return (self._shape(obs, obs.shape[-1]), self._shape(reward, 1), self._shape(done, 1),
                self._shape(bad_done, 1), self._shape(exceed_time_limit, 1), info)
