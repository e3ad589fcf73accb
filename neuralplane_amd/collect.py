"""DeviceCollector — the runner's collect step with everything pre-bound.

`F16SimRunner.run` (reference runner/F16sim_runner.py:52-66) does, per step: `collect` (:123-129: policy.get_actions on the buffer's slot
`step`), `envs.step(actions)`, `insert` (:131-154: masks, zeroed recurrent states, ReplayBuffer.insert).  With `FusedPolicy`, `DeviceVecEnv`
and `DeviceReplayBuffer` those are three launches (+ the normal draws) — 47 us of kernels at 3 000 envs — behind ~55 us of Python (tensor
views of the buffer's slots, five output allocations, argument checks): at the sizes the reference trains at the host sets the rate.
`DeviceCollector(policy, envs, buffer).step()` is the same step with the addresses computed instead of sliced: the policy reads the buffer's
slot `step` in place and writes actions / log-probabilities / values straight into it, the env steps on that slot's actions, the insert
launch finishes the slot (rewards, next observation, masks, recurrent states zeroed where an env ended).  For ControlEnv the insert launch
disappears as well (`in_place`): the env writes observation and reward into the slots themselves, the policy its recurrent states, and the
NEXT policy launch applies the runner's insert rule from the env's flags (np_policy_step.prev_flags) — two launches + the normal draws per
step, 41 us at 3 000 envs.  Same results as the three calls, bit for bit (tests/test_gpu_policy.py); nothing here computes.
"""
import ctypes as C

import torch

from . import _lib
from .buffer import DeviceReplayBuffer
from .envs.env_base import BaseEnv
from .policy import ACTOR, CRITIC, FusedPolicy


def fuse_or_torch(policy, device='cuda:0', **kw):
    """FusedPolicy(policy) when its networks have the shape the fused kernels are built for (hidden "128 128", act-hidden "128 128", one GRU layer
    of 128 — the reference's training scripts; its command line also takes --hidden-size / --act-hidden-size / --recurrent-hidden-size,
    /root/reference/config.py:48-285); any other shape: a warning naming the supported one, and the policy itself — DeviceCollector then runs
    its torch get_actions on the device loop."""
    import warnings
    try:
        return FusedPolicy(policy, device, **kw)
    except ValueError as e:
        warnings.warn(f'FusedPolicy: {e}; collecting with the policy\'s own torch modules on the device loop', RuntimeWarning, stacklevel=2)
        return policy

HID = 128


class DeviceCollector:
    """collector = DeviceCollector(policy, envs, buffer); `collector.step()` = one collect step at `buffer.step` (which it advances);
    `collector.compute_returns()` = the runner's `compute` (:112-121).  Single-agent envs (ControlEnv, PlanningEnv under DeviceVecEnv)."""

    def __init__(self, policy, envs, buffer, in_place=True, noise_block=1):
        """in_place: for ControlEnv, skip the insert launch (see below); False keeps the three launches per step.
        noise_block = K > 1: the normal draws of K consecutive steps come from ONE torch.randn((K, n, A)) (the same generator, one launch per K
        steps instead of one per step; the draws are then not the ones K separate calls would have produced)."""
        if not isinstance(buffer, DeviceReplayBuffer):
            raise TypeError('DeviceCollector(policy, envs: DeviceVecEnv, buffer: DeviceReplayBuffer)')
        env = getattr(envs, 'env', envs)
        self.policy, self.env, self.buffer = policy, env, buffer
        # a policy of another shape than the fused kernels' (FusedPolicy raises ValueError for it, see fuse_or_torch): the same loop with the
        # policy's own torch get_actions / get_values on the device — env.step and the insert launch stay this library's
        self.fused = isinstance(policy, FusedPolicy)
        if not self.fused and not (hasattr(policy, 'get_actions') and hasattr(policy, 'get_values')):
            raise TypeError('policy: a FusedPolicy, or an object with the reference PPOPolicy\'s get_actions / get_values on device tensors')
        self.device = policy.device if self.fused else buffer.device
        n, A = int(env.n), int(getattr(env, 'num_agents', 1))
        if A != 1 or buffer.num_agents != 1 or buffer.n_rollout_threads != n:
            raise ValueError(f'single-agent envs only: env rows {n} x {A} agents, buffer {buffer.n_rollout_threads} x {buffer.num_agents}')
        if buffer.device != self.device or torch.device(env.device) != self.device:
            raise ValueError('policy, envs and buffer must live on the same device')
        self._plain = type(env).step is BaseEnv.step       # ControlEnv: the batch's own step (obs, reward, flags[3, n]) without the per-flag views
        self.n = n
        if not self.fused:
            self.in_place, self._pending, self.noise_block = False, None, 1
            return
        if (buffer.obs.shape[-1], buffer.actions.shape[-1]) != (policy.obs_dim, policy.act_dim):
            raise ValueError(f'buffer holds {buffer.obs.shape[-1]} observations / {buffer.actions.shape[-1]} actions, the policy {policy.obs_dim} / {policy.act_dim}')
        if buffer.recurrent_hidden_layers != 1 or buffer.recurrent_hidden_size != HID:
            raise ValueError('recurrent state: one layer of 128')
        self.n = n
        d = self.device
        self.noise_block = max(1, int(noise_block))
        self._block = torch.empty((self.noise_block, n, policy.act_dim), dtype=torch.float32, device=d)
        self.noise = self._block[0]
        self._drawn = 0            # steps of the current block already used
        self.ha, self.hc = torch.empty((n, HID), dtype=torch.float32, device=d), torch.empty((n, HID), dtype=torch.float32, device=d)
        self._lib = _lib.load()
        self._bound = None
        # ControlEnv: no insert launch at all.  The env writes observation / reward straight into the storage's slots, the policy writes its
        # recurrent states straight into slot step + 1, and the NEXT policy launch applies the runner's insert rule from the env's flags
        # (np_policy_step.prev_flags: masks / bad_masks of its slot, recurrent states of ended envs zeroed in place).  The last slot of a
        # rollout is finished by one in-place insert launch (`finish`, called when the buffer wraps and by compute_returns).
        self.in_place = bool(in_place) and self._plain and policy.obs_dim == 22
        self._flag_bufs = [torch.zeros((3, n), dtype=torch.uint8, device=d) for _ in range(2)]
        self._pending = None       # (flags tensor, slot) of the env step whose insert rule has not been applied yet

    def _bind(self):
        """Base addresses of the storage (re-read when the buffer re-allocated or somebody replaced a tensor)."""
        b = self.buffer
        key = tuple(getattr(b, k).data_ptr() for k in b._STORAGE)
        if self._bound is None or self._bound[0] != key:
            assert all(getattr(b, k).is_contiguous() for k in b._STORAGE)
            q = _lib.NpRolloutStep()
            q.num_envs, q.num_agents, q.obs_dim, q.act_dim, q.rnn_dim = self.n, 1, self.policy.obs_dim, self.policy.act_dim, HID
            for k in b._STORAGE:
                setattr(q, k, getattr(b, k).data_ptr())
            q.rnn_states_actor_in, q.rnn_states_critic_in = self.ha.data_ptr(), self.hc.data_ptr()
            self._bound = (key, dict(zip(b._STORAGE, key)), q)
        return self._bound[1], self._bound[2]

    def _noise_ptr(self):
        """The address of this step's normal draws (drawn now, or K steps at a time)."""
        if self.noise_block == 1:
            self.noise.normal_()                            # torch's generator on this device, as the reference's sample()
            return self.noise.data_ptr()
        k = self._drawn % self.noise_block
        if k == 0:
            self._block.normal_()
        self._drawn += 1
        return self._block.data_ptr() + 4 * k * self.n * self.policy.act_dim

    def finish(self):
        """Apply the insert rule that is still pending for the newest slot (in_place mode): masks, bad_masks, zeroed recurrent states of the envs
        that ended in the last env step — one np_rollout_insert launch working in place.

        Between two in_place steps slot step + 1 is UNFINISHED (stale masks / bad_masks, recurrent states of ended envs not yet zeroed): call
        finish() before reading the storage directly mid-rollout (a checkpoint, buffer.compute_returns, policy.get_values on the newest slot).
        compute_returns() here and the wrap of the buffer call it themselves."""
        if self._pending is None:
            return
        flags, slot = self._pending              # slot = step index of that env step: its results live in slot (actions …) and slot + 1 (obs …)
        base, qi = self._bind()
        n, f4, od, ad = self.n, 4 * self.n, self.policy.obs_dim, self.policy.act_dim
        qi.step = slot
        qi.obs_in, qi.rewards_in = base['obs'] + (slot + 1) * f4 * od, base['rewards'] + slot * f4
        qi.actions_in, qi.action_log_probs_in, qi.values_in = base['actions'] + slot * f4 * ad, base['action_log_probs'] + slot * f4, base['value_preds'] + slot * f4
        ra, rc = base['rnn_states_actor'] + (slot + 1) * f4 * HID, base['rnn_states_critic'] + (slot + 1) * f4 * HID
        qi.rnn_states_actor_in, qi.rnn_states_critic_in = ra, rc
        fp = flags.data_ptr()
        qi.done_in, qi.bad_done_in, qi.exceed_time_limit_in = fp, fp + n, fp + 2 * n
        _lib.check(self._lib.np_rollout_insert(C.byref(qi), self.device.index, _lib.stream_ptr(self.device)))
        qi.rnn_states_actor_in, qi.rnn_states_critic_in = self.ha.data_ptr(), self.hc.data_ptr()
        self._pending = None

    def _step_in_place(self):
        p, b, n = self.policy, self.buffer, self.n
        p._maybe_refresh()
        base, _ = self._bind()
        s, od, ad = b.step, p.obs_dim, p.act_dim
        f4 = 4 * n
        if self._pending is not None and self._pending[1] + 1 != s:      # somebody moved buffer.step: settle the old slot first
            self.finish()
        noise_ptr = self._noise_ptr()
        q = p._q
        q.n, q.flags = n, ACTOR | CRITIC
        q.obs, q.noise = base['obs'] + s * f4 * od, noise_ptr
        m_ptr, bm_ptr = base['masks'] + s * f4, base['bad_masks'] + s * f4
        if self._pending is not None:
            q.prev_flags, q.masks, q.masks_out, q.bad_masks_out = self._pending[0].data_ptr(), None, m_ptr, bm_ptr
        else:
            q.prev_flags, q.masks = None, m_ptr
        q.rnn_states_actor_in, q.rnn_states_critic_in = base['rnn_states_actor'] + s * f4 * HID, base['rnn_states_critic'] + s * f4 * HID
        q.rnn_states_actor_out, q.rnn_states_critic_out = base['rnn_states_actor'] + (s + 1) * f4 * HID, base['rnn_states_critic'] + (s + 1) * f4 * HID
        q.values, q.actions, q.action_log_probs = base['value_preds'] + s * f4, base['actions'] + s * f4 * ad, base['action_log_probs'] + s * f4
        try:
            _lib.check(self._lib.np_policy_act(C.byref(q), self.device.index, _lib.stream_ptr(self.device)))
        finally:
            q.prev_flags = None    # the struct is the policy's own: a failed launch must not leave the collector's mode set in it
        flags = self._flag_bufs[0] if self.env._batch.flags.data_ptr() != self._flag_bufs[0].data_ptr() else self._flag_bufs[1]
        out = (b.obs[s + 1].view(n, od), b.rewards[s].view(n), flags)
        self.env._batch.step(b.actions[s].view(n, ad), out=out)
        self._pending = (flags, s)
        b.step = (s + 1) % b.buffer_size
        if b.step == 0:
            self.finish()                        # the rollout is complete: its last slot settled before anybody reads or copies it
        return out

    def step(self):
        """One collect step at buffer.step: returns the env's (obs, reward, flags[3, n] uint8 = done / bad_done / exceed_time_limit)."""
        if self.in_place:
            return self._step_in_place()
        if not self.fused:
            return self._step_torch_policy()
        p, b, n = self.policy, self.buffer, self.n
        p._maybe_refresh()
        base, qi = self._bind()
        s, od, ad = b.step, p.obs_dim, p.act_dim
        f4 = 4 * n
        a_ptr, lp_ptr, v_ptr = base['actions'] + s * f4 * ad, base['action_log_probs'] + s * f4, base['value_preds'] + s * f4
        noise_ptr = self._noise_ptr()
        q = p._q
        q.n, q.flags = n, ACTOR | CRITIC
        q.obs, q.masks, q.noise, q.prev_flags = base['obs'] + s * f4 * od, base['masks'] + s * f4, noise_ptr, None
        q.rnn_states_actor_in, q.rnn_states_critic_in = base['rnn_states_actor'] + s * f4 * HID, base['rnn_states_critic'] + s * f4 * HID
        q.values, q.actions, q.action_log_probs = v_ptr, a_ptr, lp_ptr
        q.rnn_states_actor_out, q.rnn_states_critic_out = self.ha.data_ptr(), self.hc.data_ptr()
        stream = _lib.stream_ptr(self.device)
        _lib.check(self._lib.np_policy_act(C.byref(q), self.device.index, stream))
        actions = b.actions[s].view(n, ad)
        if self._plain:
            obs, reward, flags = self.env._batch.step(actions)
        else:
            obs, reward, done, bad, tmo, _ = self.env.step(actions)
            flags = torch.stack((done, bad, tmo)).view(torch.uint8) if done.dtype == torch.bool else torch.stack((done, bad, tmo)).to(torch.uint8)
        if not (obs.is_contiguous() and reward.is_contiguous() and flags.is_contiguous() and flags.dtype == torch.uint8):
            obs, reward, flags = obs.contiguous(), reward.contiguous(), flags.contiguous().view(torch.uint8)
        qi.step = s
        qi.obs_in, qi.rewards_in = obs.data_ptr(), reward.data_ptr()
        qi.actions_in, qi.action_log_probs_in, qi.values_in = a_ptr, lp_ptr, v_ptr          # already in their slot: the launch rewrites them in place
        fp = flags.data_ptr()
        qi.done_in, qi.bad_done_in, qi.exceed_time_limit_in = fp, fp + n, fp + 2 * n
        _lib.check(self._lib.np_rollout_insert(C.byref(qi), self.device.index, stream))
        b.step = (s + 1) % b.buffer_size
        return obs, reward, flags

    def _step_torch_policy(self):
        """The collect step with a policy this library has no fused kernel for: `policy.get_actions` as the reference's runner calls it
        (runner/F16sim_runner.py:123-129: rows = envs x agents, recurrent states [rows, layers, hidden], masks [rows, 1]) on views of the
        buffer's slot, env.step on its actions, the runner's insert as one np_rollout_insert launch."""
        p, b, n = self.policy, self.buffer, self.n
        s = b.step
        L, H = b.recurrent_hidden_layers, b.recurrent_hidden_size
        with torch.no_grad():
            values, actions, logp, ha, hc = p.get_actions(b.obs[s].reshape(n, -1), b.rnn_states_actor[s].reshape(n, L, H),
                                                          b.rnn_states_critic[s].reshape(n, L, H), b.masks[s].reshape(n, 1))
        actions = torch.as_tensor(actions, device=self.device).to(torch.float32).reshape(n, -1).contiguous()
        if self._plain:
            obs, reward, flags = self.env._batch.step(actions)
            done, bad, tmo = flags[0], flags[1], flags[2]
        else:
            obs, reward, done, bad, tmo, _ = self.env.step(actions)
            flags = torch.stack((done, bad, tmo)).view(torch.uint8) if done.dtype == torch.bool else torch.stack((done, bad, tmo)).to(torch.uint8)
        as_t = lambda x: torch.as_tensor(x, device=self.device)   # noqa: E731
        b.insert_step(obs, actions, reward, done, bad, tmo, as_t(logp), as_t(values), as_t(ha), as_t(hc))
        return obs, reward, flags

    def compute_returns(self):
        """F16SimRunner.compute (:112-121): next values from the critic on the last slot, then ReplayBuffer.compute_returns."""
        self.finish()
        b, n = self.buffer, self.n
        if self.fused:
            nv = self.policy.get_values(b.obs[-1].reshape(n, -1), b.rnn_states_critic[-1].reshape(n, HID), b.masks[-1].reshape(n, 1))
        else:
            with torch.no_grad():
                nv = torch.as_tensor(self.policy.get_values(b.obs[-1].reshape(n, -1), b.rnn_states_critic[-1].reshape(n, b.recurrent_hidden_layers, b.recurrent_hidden_size),
                                                            b.masks[-1].reshape(n, 1)), device=self.device).to(torch.float32)
        b.compute_returns(nv.reshape(n, 1, 1))
