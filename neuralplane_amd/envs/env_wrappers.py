"""GPUVecEnv — numpy <-> torch adapter with the reference's contract (envs/env_wrappers.py:84-123):
`reset() -> np[E,A,obs]`, `step(np[E,A,act]) -> (np[E,A,obs], np[E,A,1] x4, info)`."""
from abc import ABC, abstractmethod

import numpy as np
import torch


def _t2n(x):
    return x.detach().cpu().numpy()


class VecEnv(ABC):
    """The reference's abstract vectorised-env protocol (envs/env_wrappers.py:9-82): reset / step_async / step_wait / close,
    with step() = step_async + step_wait."""
    closed = False

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        self.close_extras()
        self.closed = True

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()


class GPUVecEnv(VecEnv):
    def __init__(self, env_fns):
        assert len(env_fns) == 1, 'GPUVecEnv wraps exactly one batched env'
        self.env = self.gpu_vec_env = env_fns[0]()      # `gpu_vec_env`: the attribute name of the reference (env_wrappers.py:88)
        self.num_envs = self.env.num_envs
        self.agents = self.num_agents = self.env.num_agents
        self.n = self.env.n
        self.device = self.env.device
        self.observation_space = self.env.observation_space
        self.action_space = self.env.action_space
        self.closed = False
        self._pending = None

    def _shape(self, x, k):
        return x.reshape(self.num_envs, self.num_agents, k)

    def reset(self):
        obs = self.env.reset()
        return _t2n(self._shape(obs, obs.shape[-1]))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    # The reference's VecEnv protocol (envs/env_wrappers.py:40-82,113-123) splits a step into step_async / step_wait; its
    # GPUVecEnv leaves both empty.  Here they are the natural halves of a GPU step: step_async enqueues the H2D copy and the
    # fused kernel on the current stream and returns at once, step_wait brings the results to the host (the synchronising part).
    def step_async(self, actions):
        if self.closed:
            raise RuntimeError('step_async on a closed GPUVecEnv')
        a = torch.as_tensor(np.asarray(actions), dtype=torch.float32, device=self.device).reshape(self.n, -1)
        self._pending = self.env.step(a)

    def step_wait(self):
        if self._pending is None:
            raise RuntimeError('step_wait without a step_async in flight')
        obs, reward, done, bad_done, exceed_time_limit, info = self._pending
        self._pending = None
        return (_t2n(self._shape(obs, obs.shape[-1])), _t2n(self._shape(reward, 1)), _t2n(self._shape(done, 1)),
                _t2n(self._shape(bad_done, 1)), _t2n(self._shape(exceed_time_limit, 1)), info)

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        self.close_extras()
        self._pending = None
        self.closed = True


class PinnedVecEnv(GPUVecEnv):
    """GPUVecEnv with the same numpy-in / numpy-out contract, moved through page-locked staging buffers: the actions go
    host -> pinned -> device, and obs / reward / the three masks come back with ONE synchronisation into a small ring of
    pinned host buffers.  GPUVecEnv's `.cpu().numpy()` allocates fresh pageable arrays every step (5 D2H copies through
    the driver's bounce buffers + page faults on 95 MB): 31.9 ms per step at N = 1e6 against 0.4 ms of kernel time.

    The returned arrays are VIEWS of the ring: they stay valid until `ring` further `step`/`reset` calls (default 2 — the
    reference's runners copy what they keep: `buffer.obs[step + 1] = obs.copy()`, runner/F16sim_runner.py:123-154)."""

    def __init__(self, env_fns, ring=2):
        super().__init__(env_fns)
        self._ring = [dict() for _ in range(max(1, int(ring)))]
        self._slot = 0
        self._act_host = None
        self._act_dev = None

    def _to_host(self, bufs, name, src):
        dst = bufs.get(name)
        if dst is None or dst.shape != src.shape or dst.dtype != src.dtype:
            dst = bufs[name] = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        dst.copy_(src, non_blocking=True)
        return dst

    def _next(self):
        self._slot = (self._slot + 1) % len(self._ring)
        return self._ring[self._slot]

    def reset(self):
        obs = self.env.reset()
        h = self._to_host(self._next(), 'obs', obs)
        torch.cuda.current_stream(self.device).synchronize()
        return self._shape(h, h.shape[-1]).numpy()

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def step_async(self, actions):
        if self.closed:
            raise RuntimeError('step_async on a closed PinnedVecEnv')
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, -1)
        if self._act_host is None or self._act_host.shape != a.shape:
            self._act_host = torch.empty(a.shape, dtype=torch.float32, pin_memory=True)
            self._act_dev = torch.empty(a.shape, dtype=torch.float32, device=self.device)
        self._act_host.numpy()[...] = a                          # host memcpy into the page-locked staging buffer
        self._act_dev.copy_(self._act_host, non_blocking=True)
        obs, reward, done, bad_done, exceed_time_limit, info = self.env.step(self._act_dev)
        bufs = self._next()
        h = [self._to_host(bufs, k, v) for k, v in (('obs', obs), ('reward', reward), ('done', done), ('bad', bad_done),
                                                    ('tmo', exceed_time_limit))]
        self._pending = (h, info)                                   # kernel and the five D2H copies are in flight

    def step_wait(self):
        if self._pending is None:
            raise RuntimeError('step_wait without a step_async in flight')
        h, info = self._pending
        self._pending = None
        torch.cuda.current_stream(self.device).synchronize()       # the only host<->device synchronisation of the step
        return (self._shape(h[0], h[0].shape[-1]).numpy(), self._shape(h[1], 1).numpy(), self._shape(h[2], 1).numpy(),
                self._shape(h[3], 1).numpy(), self._shape(h[4], 1).numpy(), info)


class DeviceVecEnv(GPUVecEnv):
    """GPUVecEnv with the same `[E, A, ...]` shapes but torch tensors that never leave the GPU
    (SURVEY.md §8f N1): GPUVecEnv pays 1 H2D + 5 D2H copies per step (~126 B per aircraft, ~2 ms at
    N = 1e6 over PCIe Gen5 — five times the fused kernel); a device-resident policy / rollout buffer
    does not need them."""

    def reset(self):
        obs = self.env.reset()
        return self._shape(obs, obs.shape[-1])

    def step_async(self, actions):
        if self.closed:
            raise RuntimeError('step_async on a closed DeviceVecEnv')
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device).reshape(self.n, -1)
        self._pending = self.env.step(a)

    def step_wait(self):
        if self._pending is None:
            raise RuntimeError('step_wait without a step_async in flight')
        obs, reward, done, bad_done, exceed_time_limit, info = self._pending
        self._pending = None
        return (self._shape(obs, obs.shape[-1]), self._shape(reward, 1), self._shape(done, 1),
                self._shape(bad_done, 1), self._shape(exceed_time_limit, 1), info)
