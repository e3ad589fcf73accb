"""DeviceReplayBuffer — the reference's rollout storage (algorithms/utils/buffer.py:27-256, `ReplayBuffer`) resident on the GPU.

The reference keeps the rollout in numpy: with `GPUVecEnv` every env.step crosses PCIe six times and `insert()` copies
each field on the host.  With `DeviceVecEnv` the observations, rewards and masks are already torch
tensors on the device; this class keeps them there.  Same constructor (`args`, `num_agents`, `obs_space`, `act_space`),
same field names and shapes, same methods and semantics:

* `insert(obs, actions, rewards, masks, action_log_probs, value_preds, rnn_states_actor, rnn_states_critic, bad_masks=None)`
  — buffer.py:76-112 (device-to-device copies; numpy inputs are uploaded);
* `insert_step(obs, actions, rewards, dones, bad_dones, exceed_time_limits, action_log_probs, values, rnn_states_actor, rnn_states_critic)` — the
  runner's `insert(data)` (runner/F16sim_runner.py:131-154) and the buffer's `insert` fused into one launch (not in the reference: for device-resident loops);
* `after_update()`, `clear()` — buffer.py:114-135;
* `compute_returns(next_value)` — buffer.py:137-173: ONE kernel launch (`np_rollout_returns`, csrc/np_rollout.h) instead of a
  Python loop over `buffer_size` steps of numpy array operations; bit-exact to the reference's float32 arithmetic;
* `advantages` — buffer.py:68-74 (population std, as numpy's);
* `recurrent_generator(buffer, num_mini_batch, data_chunk_length)` — buffer.py:175-256: the same chunking and the same
  `torch.randperm` shuffle (CPU generator, so a seeded run draws the reference's permutation); the batches are gathered on the
  device and yielded as torch tensors.

There is no CPU fallback: `compute_returns` needs the HIP library and a device tensor.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _shape(space):
    """get_shape_from_space (algorithms/utils/utils.py:16-28) for the Box spaces of the envs on this path."""
    if hasattr(space, 'shape') and space.shape is not None and len(space.shape) > 0:
        return tuple(space.shape)
    if hasattr(space, 'n'):
        return (1,)
    raise NotImplementedError(f"Unsupported action space type: {type(space)}!")


class DeviceReplayBuffer:

    @staticmethod
    def _flatten(T, N, x):
        return x.reshape(T * N, *x.shape[2:])

    @staticmethod
    def _cast(x):
        # [T, threads, agents, ...] -> [threads, agents, T, ...] -> [threads * agents * T, ...]   (buffer.py:33-35)
        return x.permute(1, 2, 0, *range(3, x.dim())).reshape(-1, *x.shape[3:])

    def __init__(self, args, num_agents, obs_space, act_space, device='cuda:0'):
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:   # 'cuda' = the CURRENT device (a torchrun rank after set_device)
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.buffer_size = args.buffer_size
        self.n_rollout_threads = args.n_rollout_threads
        self.num_agents = num_agents
        self.gamma = args.gamma
        self.use_proper_time_limits = args.use_proper_time_limits
        self.use_gae = args.use_gae
        self.gae_lambda = args.gae_lambda
        self.recurrent_hidden_size = args.recurrent_hidden_size
        self.recurrent_hidden_layers = args.recurrent_hidden_layers
        self._obs_shape, self._act_shape = _shape(obs_space), _shape(act_space)
        self._alloc()
        self.step = 0

    def _alloc(self):
        self._q_insert = None   # insert_step's argument struct holds the storage addresses
        T, N, A = self.buffer_size, self.n_rollout_threads, self.num_agents
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.device)
        o = lambda *s: torch.ones(s, dtype=torch.float32, device=self.device)
        self.obs = z(T + 1, N, A, *self._obs_shape)
        self.actions = z(T, N, A, *self._act_shape)
        self.rewards = z(T, N, A, 1)
        self.masks = o(T + 1, N, A, 1)
        self.bad_masks = o(T + 1, N, A, 1)
        self.action_log_probs = z(T, N, A, 1)
        self.value_preds = z(T + 1, N, A, 1)
        self.returns = z(T + 1, N, A, 1)
        self.rnn_states_actor = z(T + 1, N, A, self.recurrent_hidden_layers, self.recurrent_hidden_size)
        self.rnn_states_critic = torch.zeros_like(self.rnn_states_actor)

    def _dev(self, x, like):
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return x.to(device=self.device, dtype=torch.float32).reshape(like.shape)

    @property
    def advantages(self):
        adv = self.returns[:-1] - self.value_preds[:-1]
        return (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-5)

    def insert(self, obs, actions, rewards, masks, action_log_probs, value_preds, rnn_states_actor, rnn_states_critic, bad_masks=None,
               **kwargs):
        s = self.step
        self.obs[s + 1].copy_(self._dev(obs, self.obs[0]))
        self.actions[s].copy_(self._dev(actions, self.actions[0]))
        self.rewards[s].copy_(self._dev(rewards, self.rewards[0]))
        self.masks[s + 1].copy_(self._dev(masks, self.masks[0]))
        self.action_log_probs[s].copy_(self._dev(action_log_probs, self.action_log_probs[0]))
        self.value_preds[s].copy_(self._dev(value_preds, self.value_preds[0]))
        self.rnn_states_actor[s + 1].copy_(self._dev(rnn_states_actor, self.rnn_states_actor[0]))
        self.rnn_states_critic[s + 1].copy_(self._dev(rnn_states_critic, self.rnn_states_critic[0]))
        if bad_masks is not None:
            self.bad_masks[s + 1].copy_(self._dev(bad_masks, self.bad_masks[0]))
        self.step = (self.step + 1) % self.buffer_size

    _STORAGE = ('obs', 'actions', 'rewards', 'masks', 'bad_masks', 'action_log_probs', 'value_preds', 'rnn_states_actor', 'rnn_states_critic')

    def insert_step(self, obs, actions, rewards, dones, bad_dones, exceed_time_limits, action_log_probs, values, rnn_states_actor, rnn_states_critic):
        """One collect step, as the reference's runner hands it over — `F16SimRunner.insert(data)` (runner/F16sim_runner.py:131-154: the recurrent
        states of envs that ended are zeroed, masks / bad_masks from dones / bad_dones, `any` over the agents of an env) followed by
        `ReplayBuffer.insert` (buffer.py:76-112) — as ONE launch (np_rollout_insert) instead of ~20 small torch kernels.  Device tensors in the
        shapes DeviceVecEnv.step and the policy return them ([E, A, ...] or flat [E * A, ...]); dones etc. bool or uint8."""
        if self.device.type != 'cuda':
            raise RuntimeError('DeviceReplayBuffer.insert_step runs on the GPU (np_rollout_insert); there is no CPU fallback')
        E, A = self.n_rollout_threads, self.num_agents
        N = E * A
        dev = self.device

        def f(x, d):   # device-resident float32 rows are used where they are (only the address is needed)
            if type(x) is torch.Tensor and x.dtype is torch.float32 and x.device == dev and x.is_contiguous() and x.numel() == N * d:
                return x
            return x.to(device=dev, dtype=torch.float32).reshape(N, d).contiguous()

        def u8(x):
            if type(x) is torch.Tensor and x.device == dev and x.is_contiguous() and x.numel() == N and x.dtype in (torch.bool, torch.uint8):
                return x   # bool and uint8 are both one byte, 0 / 1
            return (x.view(torch.uint8) if x.dtype == torch.bool else x.to(torch.uint8)).to(dev).reshape(N).contiguous()
        od, ad = int(np.prod(self._obs_shape)), int(np.prod(self._act_shape))
        rd = self.recurrent_hidden_layers * self.recurrent_hidden_size
        keep = (f(obs, od), f(actions, ad), f(rewards, 1), f(action_log_probs, 1), f(values, 1), f(rnn_states_actor, rd), f(rnn_states_critic, rd),
                u8(dones), u8(bad_dones), u8(exceed_time_limits))
        q = self._q_insert
        if q is None or any(getattr(q, name) != getattr(self, name).data_ptr() for name in self._STORAGE):   # somebody replaced a storage tensor
            q = self._q_insert = _lib.NpRolloutStep()
            q.num_envs, q.num_agents, q.obs_dim, q.act_dim, q.rnn_dim = E, A, od, ad, rd
            for name in self._STORAGE:
                t = getattr(self, name)
                assert t.is_contiguous()
                setattr(q, name, t.data_ptr())
        q.step = self.step
        (q.obs_in, q.actions_in, q.rewards_in, q.action_log_probs_in, q.values_in, q.rnn_states_actor_in, q.rnn_states_critic_in, q.done_in, q.bad_done_in,
         q.exceed_time_limit_in) = [t.data_ptr() for t in keep]
        stream = _lib.stream_ptr(self.device)
        _lib.check(_lib.load().np_rollout_insert(C.byref(q), self.device.index, stream))
        self.step = (self.step + 1) % self.buffer_size

    def after_update(self):
        self.obs[0].copy_(self.obs[-1])
        self.masks[0].copy_(self.masks[-1])
        self.bad_masks[0].copy_(self.bad_masks[-1])
        self.rnn_states_actor[0].copy_(self.rnn_states_actor[-1])
        self.rnn_states_critic[0].copy_(self.rnn_states_critic[-1])

    def clear(self):
        self.step = 0
        self._alloc()

    def compute_returns(self, next_value):
        if self.device.type != 'cuda':
            raise RuntimeError('DeviceReplayBuffer.compute_returns runs on the GPU (np_rollout_returns); there is no CPU fallback')
        lib = _lib.load()
        nv = self._dev(next_value, self.value_preds[0]).contiguous()
        T, N = self.buffer_size, self.n_rollout_threads * self.num_agents
        for x in (self.rewards, self.value_preds, self.masks, self.bad_masks, self.returns):
            assert x.is_contiguous()
        stream = _lib.stream_ptr(self.device)
        _lib.check(lib.np_rollout_returns(T, N, float(self.gamma), float(self.gae_lambda), int(bool(self.use_gae)),
                                          int(bool(self.use_proper_time_limits)), self.rewards.data_ptr(), self.value_preds.data_ptr(),
                                          self.masks.data_ptr(), self.bad_masks.data_ptr(), nv.data_ptr(), self.returns.data_ptr(),
                                          self.device.index, stream))

    @staticmethod
    def recurrent_generator(buffer, num_mini_batch, data_chunk_length):
        buffer = [buffer] if isinstance(buffer, DeviceReplayBuffer) else buffer
        n_rollout_threads, buffer_size, num_agents = buffer[0].n_rollout_threads, buffer[0].buffer_size, buffer[0].num_agents
        for b in buffer:
            if not isinstance(b, DeviceReplayBuffer) or (b.n_rollout_threads, b.buffer_size, b.num_agents) != (n_rollout_threads, buffer_size, num_agents):
                raise AssertionError('recurrent_generator: every buffer must be a DeviceReplayBuffer of the same '
                                     '(n_rollout_threads, buffer_size, num_agents)')
        buffer_size = buffer_size * len(buffer)
        if n_rollout_threads * buffer_size < data_chunk_length:
            raise AssertionError(f'recurrent_generator: {n_rollout_threads} rollout threads x {buffer_size} steps hold fewer than one '
                                 f'chunk of data_chunk_length={data_chunk_length} (num_agents={num_agents})')
        cast, cat = DeviceReplayBuffer._cast, (lambda xs: torch.cat(xs, dim=0))
        obs = cat([cast(b.obs[:-1]) for b in buffer])
        actions = cat([cast(b.actions) for b in buffer])
        masks = cat([cast(b.masks[:-1]) for b in buffer])
        old_action_log_probs = cat([cast(b.action_log_probs) for b in buffer])
        advantages = cat([cast(b.advantages) for b in buffer])
        returns = cat([cast(b.returns[:-1]) for b in buffer])
        value_preds = cat([cast(b.value_preds[:-1]) for b in buffer])
        rnn_states_actor = cat([cast(b.rnn_states_actor[:-1]) for b in buffer])
        rnn_states_critic = cat([cast(b.rnn_states_critic[:-1]) for b in buffer])

        data_chunks = n_rollout_threads * buffer_size // data_chunk_length   # as the reference: agents are not counted here
        mini_batch_size = data_chunks // num_mini_batch
        rand = torch.randperm(data_chunks)                                    # CPU generator: the reference's permutation
        L, N = data_chunk_length, mini_batch_size
        ar = torch.arange(L, device=obs.device)
        for i in range(num_mini_batch):
            indices = rand[i * mini_batch_size:(i + 1) * mini_batch_size].to(obs.device)
            first = indices * L                                               # [N] first row of every chunk
            rows = (first[None, :] + ar[:, None]).reshape(-1)                 # [L, N] -> L * N, time-major like np.stack(axis=1)
            yield (obs[rows], actions[rows], masks[rows], old_action_log_probs[rows], advantages[rows], returns[rows], value_preds[rows],
                   rnn_states_actor[first].reshape(N, *buffer[0].rnn_states_actor.shape[3:]),
                   rnn_states_critic[first].reshape(N, *buffer[0].rnn_states_critic.shape[3:]))
