#!/usr/bin/env python3
"""SURVEY §8(f) N1, measured end to end: the reference's COLLECT LOOP (runner/F16sim_runner.py:52-66 `run`, :123-129 `collect`,
:131-154 `insert`, algorithms/utils/buffer.py:76-112 `ReplayBuffer.insert`, :137-173 `compute_returns`) at the batch sizes the
reference trains at (scripts/train_heading.sh: 3 000 rollout threads; scripts/train_tracking.sh: 10 000), three ways:

  device   policy (torch, on the GPU) -> DeviceVecEnv.step -> DeviceReplayBuffer.insert, everything stays on the device;
  graph    the same, with {policy forward, env.step} replayed from ONE captured HIP graph per step (static buffers; the insert stays eager:
           its destination moves with the step index);
  numpy    the reference's contract as its runner uses it: numpy observations -> policy on the GPU -> numpy actions ->
           PinnedVecEnv.step (1 H2D + 5 D2H through page-locked buffers, one synchronisation) -> a numpy rollout buffer.

Per variant: wall microseconds per collect step (back-to-back, one synchronisation at the end) and — from a second pass with HIP events
between the phases — the GPU time of policy / env.step / insert; `host_gap` = wall - the three (the GPU waiting for the host).  Then
`compute_returns` over the collected buffer.

The policy is a from-scratch torch module of the PPO actor-critic SHAPE the reference trains (feature LayerNorm, MLP 22-128-128 + ReLU +
LayerNorm, GRU 128 + LayerNorm, MLP 128-128, Gaussian head / value head; ppo_actor.py:10-64, ppo_critic.py) with random weights: it is
the load the env is measured beside, not part of this library and not an RL implementation (no training step is run or timed).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _Tower(nn.Module):
    def __init__(self, obs_dim=22, hid=128, out=4):
        super().__init__()
        self.ln0 = nn.LayerNorm(obs_dim)
        self.l1, self.n1 = nn.Linear(obs_dim, hid), nn.LayerNorm(hid)
        self.l2, self.n2 = nn.Linear(hid, hid), nn.LayerNorm(hid)
        self.gru, self.n3 = nn.GRUCell(hid, hid), nn.LayerNorm(hid)
        self.a1, self.n4 = nn.Linear(hid, hid), nn.LayerNorm(hid)
        self.a2, self.n5 = nn.Linear(hid, hid), nn.LayerNorm(hid)
        self.head = nn.Linear(hid, out)

    def forward(self, obs, h, masks):
        x = self.ln0(obs)
        x = self.n1(torch.relu(self.l1(x)))
        x = self.n2(torch.relu(self.l2(x)))
        h = self.gru(x, h * masks)
        x = self.n3(h)
        x = self.n4(torch.relu(self.a1(x)))
        x = self.n5(torch.relu(self.a2(x)))
        return self.head(x), h

    def reference_state_dict(self, critic):
        """The same parameters under the key names of the reference's PPOActor / PPOCritic state_dict (what FusedPolicy packs)."""
        mlp = 'mlp' if critic else 'act.mlp'
        names = {'ln0': 'base.feature_norm', 'l1': 'base.mlp.fc.0', 'n1': 'base.mlp.fc.2', 'l2': 'base.mlp.fc.3', 'n2': 'base.mlp.fc.5', 'n3': 'rnn.norm',
                 'a1': mlp + '.fc.0', 'n4': mlp + '.fc.2', 'a2': mlp + '.fc.3', 'n5': mlp + '.fc.5', 'head': 'value_out' if critic else 'act.action_out.mu_net.fc.0'}
        sd = {}
        for k, v in self.state_dict().items():
            mod, par = k.split('.')
            sd[f'rnn.gru.{par}_l0' if mod == 'gru' else f'{names[mod]}.{par}'] = v
        return sd


class TorchPolicy(nn.Module):
    """get_actions(obs[n,22], h_actor[n,128], h_critic[n,128], masks[n,1]) -> values, actions, action_log_probs, h_actor, h_critic
    (PPOPolicy.get_actions, algorithms/ppo/ppo_policy.py:26-32, stochastic actions)."""

    def __init__(self, act_dim=4, obs_dim=22):
        super().__init__()
        self.actor, self.critic = _Tower(obs_dim=obs_dim, out=act_dim), _Tower(obs_dim=obs_dim, out=1)
        self.logstd = nn.Parameter(torch.zeros(act_dim))

    @torch.no_grad()
    def get_actions(self, obs, ha, hc, masks):
        mu, ha = self.actor(obs, ha, masks)
        mu = torch.tanh(mu)                                      # MuNet: Linear + Tanh (distributions.py:77-87)
        std = self.logstd.exp()
        actions = torch.randn_like(mu) * std + mu
        logp = (-0.5 * ((actions - mu) / std) ** 2 - self.logstd - 0.9189385332046727).sum(-1, keepdim=True)
        values, hc = self.critic(obs, hc, masks)
        return values, actions, logp, ha, hc

    def state_dicts(self):
        sa = self.actor.reference_state_dict(False)
        sa['act.action_out.log_std'] = self.logstd.detach()
        return sa, self.critic.reference_state_dict(True)

    @torch.no_grad()
    def get_values(self, obs, hc, masks):
        return self.critic(obs, hc, masks)[0]

    @torch.no_grad()
    def act(self, obs, ha, masks):                               # PPOPolicy.act, sampled (ppo_policy.py:51-57)
        mu, ha = self.actor(obs, ha, masks)
        mu = torch.tanh(mu)
        return torch.randn_like(mu) * self.logstd.exp() + mu, ha


class _Args:
    def __init__(self, n, T):
        self.buffer_size, self.n_rollout_threads = T, n
        self.gamma, self.use_proper_time_limits, self.use_gae, self.gae_lambda = 0.99, True, True, 0.95
        self.recurrent_hidden_size, self.recurrent_hidden_layers = 128, 1


FUSED_INSERT = True   # DeviceReplayBuffer.insert_step (np_rollout_insert: one launch); False: the same with torch operations + buffer.insert


def _device_insert(buf, obs, actions, rewards, dones, bad_dones, tmo, logp, values, ha, hc, n):
    """F16sim_runner.insert (:131-154) on device tensors: zero the recurrent state of envs that ended, masks / bad_masks, buffer.insert."""
    if FUSED_INSERT:
        buf.insert_step(obs, actions, rewards, dones, bad_dones, tmo, logp, values, ha, hc)
        return
    reset_env = (dones | bad_dones | tmo).reshape(n, 1)
    keep = (~reset_env).to(torch.float32)
    masks = (~dones).reshape(n, 1, 1).to(torch.float32)
    bad_masks = (~bad_dones).reshape(n, 1, 1).to(torch.float32)
    buf.insert(obs, actions.reshape(n, 1, -1), rewards, masks, logp.reshape(n, 1, 1), values.reshape(n, 1, 1), (ha * keep).reshape(n, 1, 1, 128),
               (hc * keep).reshape(n, 1, 1, 128), bad_masks)


class _Phases:
    """HIP events between the phases of a step (second pass only)."""

    def __init__(self, on):
        self.on, self.ev, self.t = on, [], {'policy': 0.0, 'env': 0.0, 'insert': 0.0}

    def mark(self):
        if self.on:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.ev.append(e)

    def close(self, names=('policy', 'env', 'insert')):
        if not self.on:
            return
        torch.cuda.synchronize()
        k = len(names) + 1
        for i in range(0, len(self.ev) - k + 1, k):
            for j, nm in enumerate(names):
                self.t[nm] += self.ev[i + j].elapsed_time(self.ev[i + j + 1]) * 1e3
        self.ev = []


def run_device(n, T, dev, graph=False, task='heading', fused_policy=False, policy_numerics='fp32'):
    from neuralplane_amd.buffer import DeviceReplayBuffer
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import DeviceVecEnv
    torch.manual_seed(0)
    policy = TorchPolicy(act_dim=3 if task == 'tracking' else 4).to(dev).eval()
    if fused_policy:   # the same networks through neuralplane_amd.policy.FusedPolicy: one launch per get_actions (np_policy_act)
        from neuralplane_amd.policy import FusedPolicy
        policy = FusedPolicy(policy.state_dicts(), device=dev, numerics=policy_numerics)
    if task == 'tracking':
        # scripts/train_tracking.sh: PlanningEnv (one high-level action = 50 x {frozen controller, FDM step}); the controller's checkpoint is
        # not part of the reference snapshot: random weights of its architecture, run by the persistent kernel
        from neuralplane_amd.actor import NUM_FLOATS, FusedActor
        from neuralplane_amd.envs.planning_env import PlanningEnv
        assert not graph
        ctrl = FusedActor(np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32), str(dev), numerics='i8')
        envs = DeviceVecEnv([lambda: PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device=str(dev), controller=ctrl)])
    else:
        envs = DeviceVecEnv([lambda: ControlEnv(num_envs=n, config=task, model='F16', random_seed=0, device=str(dev))])
    env = envs.env
    buf = DeviceReplayBuffer(_Args(n, T), 1, env.observation_space, env.action_space, device=dev)
    buf.obs[0].copy_(envs.reset())
    g = None
    if graph:
        # static buffers: the graph reads obs / recurrent states / masks from them and leaves the step's outputs in them
        b = env._batch
        st = {'obs': torch.zeros((n, 22), device=dev), 'ha': torch.zeros((n, 128), device=dev), 'hc': torch.zeros((n, 128), device=dev),
              'masks': torch.ones((n, 1), device=dev), 'fa': torch.zeros((3, n), dtype=torch.uint8, device=dev),
              'fb': torch.zeros((3, n), dtype=torch.uint8, device=dev), 'obs2': torch.zeros((n, 22), device=dev), 'rew': torch.zeros(n, device=dev)}
        st['fa'].copy_(b.flags)
        b.call_base.fill_(b.call_idx)

        def body():
            v, a, lp, ha, hc = policy.get_actions(st['obs'], st['ha'], st['hc'], st['masks'])
            b.launch_static(st['fa'], st['fb'], 0, action=a.contiguous(), obs=st['obs2'], reward=st['rew'], cache_valid=True)
            st['fa'].copy_(st['fb'])
            b.call_base.add_(1)
            return v, a, lp, ha, hc

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        saved = (b.state_dict(), b.coef_cache.clone(), b.term_counters.clone())
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        b.load_state_dict(saved[0]); b.coef_cache.copy_(saved[1]); b.term_counters.copy_(saved[2])
        st['fa'].copy_(b.flags)
        b.call_base.fill_(b.call_idx)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st['out'] = body()

    def loop(ph):
        for s in range(T):
            ph.mark()
            if g is None:
                v, a, lp, ha, hc = policy.get_actions(buf.obs[s].reshape(n, 22), buf.rnn_states_actor[s].reshape(n, 128),
                                                      buf.rnn_states_critic[s].reshape(n, 128), buf.masks[s].reshape(n, 1))
                ph.mark()
                obs, rew, d, bd, tm, _ = envs.step(a)
                ph.mark()
            else:
                st['obs'].copy_(buf.obs[s].reshape(n, 22)); st['ha'].copy_(buf.rnn_states_actor[s].reshape(n, 128))
                st['hc'].copy_(buf.rnn_states_critic[s].reshape(n, 128)); st['masks'].copy_(buf.masks[s].reshape(n, 1))
                g.replay()
                ph.mark()
                ph.mark()                     # policy and env.step are one graph: reported together under `policy`
                v, a, lp, ha, hc = st['out']
                fl = st['fa'].view(torch.bool)
                obs, rew, d, bd, tm = st['obs2'].reshape(n, 1, 22), st['rew'].reshape(n, 1, 1), fl[0].reshape(n, 1, 1), fl[1].reshape(n, 1, 1), fl[2].reshape(n, 1, 1)
            _device_insert(buf, obs, a, rew, d, bd, tm, lp, v, ha, hc, n)
            ph.mark()
        ph.close()

    loop(_Phases(False))                      # warm-up pass (allocator, kernels, clocks)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    loop(_Phases(False))
    torch.cuda.synchronize(dev)
    wall_us = 1e6 * (time.perf_counter() - t0) / T
    ph = _Phases(True)
    loop(ph)
    t0 = time.perf_counter()
    nv = policy.get_values(buf.obs[-1].reshape(n, 22), buf.rnn_states_critic[-1].reshape(n, 128), buf.masks[-1].reshape(n, 1))
    buf.compute_returns(nv.reshape(n, 1, 1))
    torch.cuda.synchronize(dev)
    ret_ms = 1e3 * (time.perf_counter() - t0)
    gpu = {k: v / T for k, v in ph.t.items()}
    out = {'us_per_step_wall': wall_us, 'gpu_us_policy' + ('_and_env_graph' if graph else ''): gpu['policy'], 'gpu_us_env_step': None if graph else gpu['env'],
           'gpu_us_insert': gpu['insert'], 'host_gap_us': wall_us - sum(gpu.values()), 'compute_returns_ms': ret_ms, 'steps': T,
           'env_steps_per_s': n * 1e6 / wall_us}
    del envs, buf
    return out


def run_collector(n, T, dev, task='heading', policy_numerics='i8', noise_block=1):
    """The device loop through neuralplane_amd.collect.DeviceCollector: FusedPolicy writing into the buffer's slot in place, env.step, insert."""
    from neuralplane_amd.buffer import DeviceReplayBuffer
    from neuralplane_amd.collect import DeviceCollector
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import DeviceVecEnv
    from neuralplane_amd.policy import FusedPolicy
    torch.manual_seed(0)
    policy = FusedPolicy(TorchPolicy().to(dev).eval().state_dicts(), device=dev, numerics=policy_numerics)
    envs = DeviceVecEnv([lambda: ControlEnv(num_envs=n, config=task, model='F16', random_seed=0, device=str(dev))])
    buf = DeviceReplayBuffer(_Args(n, T), 1, envs.observation_space, envs.action_space, device=dev)
    buf.obs[0].copy_(envs.reset())
    col = DeviceCollector(policy, envs, buf, noise_block=noise_block)
    for _ in range(T):
        col.step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(T):
        col.step()
    torch.cuda.synchronize(dev)
    wall_us = 1e6 * (time.perf_counter() - t0) / T
    t0 = time.perf_counter()
    for _ in range(T):
        col.step()
    host_us = 1e6 * (time.perf_counter() - t0) / T        # enqueue only (the GPU may lag behind)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    col.compute_returns()
    torch.cuda.synchronize(dev)
    return {'us_per_step_wall': wall_us, 'host_enqueue_us_per_step': host_us, 'compute_returns_ms': 1e3 * (time.perf_counter() - t0), 'steps': T,
            'env_steps_per_s': n * 1e6 / wall_us}


def run_selfplay(E, T, dev, fused_policy=False, policy_numerics='i8'):
    """The self-play runner's collect step (runner/selfplay_F16sim_runner.py:76-100 `collect`, :102-128 `insert`) on the device for the 1v1 combat
    env: ego policy `get_actions` on obs[:, :A//2], opponent policy `act` (sampled) on obs[:, A//2:], env.step on both actions, the ego half into
    the rollout storage.  15 observations, 4 actions (envs/configs/selfplay.yaml), the network shapes of config.py's defaults."""
    from neuralplane_amd.buffer import DeviceReplayBuffer
    from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
    from neuralplane_amd.envs.spaces import Box
    torch.manual_seed(0)
    ego, opp = TorchPolicy(act_dim=4, obs_dim=15).to(dev).eval(), TorchPolicy(act_dim=4, obs_dim=15).to(dev).eval()
    if fused_policy:
        from neuralplane_amd.policy import FusedPolicy
        ego, opp = FusedPolicy(ego.state_dicts(), device=dev, numerics=policy_numerics), FusedPolicy(opp.state_dicts(), device=dev, numerics=policy_numerics)
    env = SingleCombatEnv(num_envs=E, config='selfplay', random_seed=0, device=str(dev))
    buf = DeviceReplayBuffer(_Args(E, T), 1, Box(low=-np.inf, high=np.inf, shape=(15,)), Box(low=-np.inf, high=np.inf, shape=(4,)), device=dev)
    oe, oo = env.reset_split()
    buf.obs[0].copy_(oe.reshape(E, 1, 15))
    st = {'opp_obs': oo, 'opp_h': torch.zeros((E, 128), device=dev), 'opp_m': torch.ones((E, 1), device=dev)}

    def loop(ph):
        for s in range(T):
            ph.mark()
            v, a, lp, ha, hc = ego.get_actions(buf.obs[s].reshape(E, 15), buf.rnn_states_actor[s].reshape(E, 128), buf.rnn_states_critic[s].reshape(E, 128),
                                               buf.masks[s].reshape(E, 1))
            oa, oh = opp.act(st['opp_obs'], st['opp_h'], st['opp_m'])
            ph.mark()
            oe, oo, rew, d, bd, tm, _ = env.step_split(a, oa)
            ph.mark()
            # rows 2k / 2k + 1 = ego / enemy: the ego half of rewards and flags
            rew_e, d_e, bd_e, tm_e = rew.reshape(E, 2)[:, 0], d.reshape(E, 2).any(1), bd.reshape(E, 2).any(1), tm.reshape(E, 2).any(1)
            _device_insert(buf, oe.reshape(E, 1, 15), a, rew_e.reshape(E, 1, 1), d_e.reshape(E, 1, 1), bd_e.reshape(E, 1, 1), tm_e.reshape(E, 1, 1), lp, v, ha, hc, E)
            ended = (d_e | bd_e | tm_e).reshape(E, 1)
            st['opp_obs'], st['opp_h'], st['opp_m'] = oo, oh.reshape(E, 128) * (~ended), (~d_e).reshape(E, 1).float()
            ph.mark()
        ph.close()

    loop(_Phases(False))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    loop(_Phases(False))
    torch.cuda.synchronize(dev)
    wall_us = 1e6 * (time.perf_counter() - t0) / T
    ph = _Phases(True)
    loop(ph)
    gpu = {k: v / T for k, v in ph.t.items()}
    return {'us_per_step_wall': wall_us, 'gpu_us_both_policies': gpu['policy'], 'gpu_us_env_step': gpu['env'], 'gpu_us_insert_and_opponent_bookkeeping': gpu['insert'],
            'engagements': E, 'steps': T, 'engagement_steps_per_s': E * 1e6 / wall_us}


class _NumpyBuffer:
    """The fields and the insert of the reference's ReplayBuffer (algorithms/utils/buffer.py:37-112), numpy, for the timing of the numpy contract."""

    def __init__(self, n, T):
        f = np.float32
        self.obs, self.actions, self.rewards = np.zeros((T + 1, n, 1, 22), f), np.zeros((T, n, 1, 4), f), np.zeros((T, n, 1, 1), f)
        self.masks, self.bad_masks = np.ones((T + 1, n, 1, 1), f), np.ones((T + 1, n, 1, 1), f)
        self.action_log_probs, self.value_preds, self.returns = np.zeros((T, n, 1, 1), f), np.zeros((T + 1, n, 1, 1), f), np.zeros((T + 1, n, 1, 1), f)
        self.rnn_states_actor, self.rnn_states_critic = np.zeros((T + 1, n, 1, 1, 128), f), np.zeros((T + 1, n, 1, 1, 128), f)
        self.step, self.T = 0, T

    def insert(self, obs, actions, rewards, masks, logp, values, ha, hc, bad_masks):
        s = self.step
        self.obs[s + 1] = obs.copy(); self.actions[s] = actions.copy(); self.rewards[s] = rewards.copy(); self.masks[s + 1] = masks.copy()
        self.bad_masks[s + 1] = bad_masks.copy(); self.action_log_probs[s] = logp.copy(); self.value_preds[s] = values.copy()
        self.rnn_states_actor[s + 1] = ha.copy(); self.rnn_states_critic[s + 1] = hc.copy()
        self.step = (s + 1) % self.T

    def compute_returns(self, next_value, gamma=0.99, lam=0.95):   # buffer.py:139-155 (use_proper_time_limits, use_gae)
        self.value_preds[-1] = next_value
        gae = 0
        for step in reversed(range(self.T)):
            td = self.rewards[step] + gamma * self.value_preds[step + 1] * self.masks[step + 1] - self.value_preds[step]
            gae = td + gamma * lam * self.masks[step + 1] * gae
            gae = gae * self.bad_masks[step + 1]
            self.returns[step] = gae + self.value_preds[step]


def run_numpy(n, T, dev, task='heading'):
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import PinnedVecEnv
    torch.manual_seed(0)
    policy = TorchPolicy().to(dev).eval()
    envs = PinnedVecEnv([lambda: ControlEnv(num_envs=n, config=task, model='F16', random_seed=0, device=str(dev))])
    buf = _NumpyBuffer(n, T)
    buf.obs[0] = envs.reset().copy()
    t2n = lambda x: x.detach().cpu().numpy()   # noqa: E731  (runner/F16sim_runner.py:10)
    split = {'policy': 0.0, 'split': 0.0, 'env': 0.0, 'insert': 0.0}

    def loop(timed):
        for s in range(T):
            t0 = time.perf_counter()
            to = lambda x: torch.from_numpy(np.concatenate(x)).to(dev)   # noqa: E731
            v, a, lp, ha, hc = policy.get_actions(to(buf.obs[s]), to(buf.rnn_states_actor[s]).reshape(n, 128), to(buf.rnn_states_critic[s]).reshape(n, 128), to(buf.masks[s]))
            host = [t2n(x) for x in (v, a, lp, ha.reshape(n, 1, 128), hc.reshape(n, 1, 128))]
            ts = time.perf_counter()
            v, a, lp, ha, hc = (np.array(np.split(x, n)) for x in host)   # "split parallel data [N * M, shape] => [N, M, shape]" (F16sim_runner.py:124-129)
            t1 = time.perf_counter()
            obs, rew, d, bd, tm, _ = envs.step(a)
            t2 = time.perf_counter()
            reset_env = np.any((d + bd + tm).squeeze(axis=-1), axis=-1)
            ha[reset_env] = 0
            hc[reset_env] = 0
            masks = np.ones((n, 1, 1), np.float32)
            masks[np.any(d.squeeze(axis=-1), axis=-1)] = 0
            bad_masks = np.ones((n, 1, 1), np.float32)
            bad_masks[np.any(bd.squeeze(axis=-1), axis=-1)] = 0
            buf.insert(obs, a, rew, masks, lp, v, ha, hc, bad_masks)
            t3 = time.perf_counter()
            if timed:
                split['policy'] += ts - t0; split['split'] += t1 - ts; split['env'] += t2 - t1; split['insert'] += t3 - t2

    loop(False)
    t0 = time.perf_counter()
    loop(True)
    wall_us = 1e6 * (time.perf_counter() - t0) / T
    t0 = time.perf_counter()
    buf.compute_returns(np.zeros((n, 1, 1), np.float32))
    ret_ms = 1e3 * (time.perf_counter() - t0)
    out = {'us_per_step_wall': wall_us, 'host_us_policy_incl_h2d_d2h': 1e6 * split['policy'] / T,
           'host_us_np_split_of_the_policy_outputs': 1e6 * split['split'] / T, 'host_us_env_step_incl_pcie_and_sync': 1e6 * split['env'] / T,
           'host_us_insert': 1e6 * split['insert'] / T, 'compute_returns_ms': ret_ms, 'steps': T, 'env_steps_per_s': n * 1e6 / wall_us,
           'note': 'every phase ends in a synchronisation (numpy out), so the host clock splits the step'}
    del envs, buf
    return out


def collect_loop_report(n, T, dev):
    dev = torch.device(dev)
    global FUSED_INSERT
    FUSED_INSERT = False
    unfused = run_device(n, T, dev)
    FUSED_INSERT = True
    rep = {'aircraft': n, 'steps_per_rollout_timed': T, 'device': run_device(n, T, dev), 'device_graph': run_device(n, T, dev, graph=True),
           'device_torch_insert': {'us_per_step_wall': unfused['us_per_step_wall'], 'gpu_us_insert': unfused['gpu_us_insert'],
                                   'note': 'the same device loop with the insert as ~20 torch kernels (masks, 9 copies) instead of DeviceReplayBuffer.insert_step'},
           'numpy_contract': run_numpy(n, T, dev)}
    # the same loop with the policy's inference step as ONE launch (neuralplane_amd.policy.FusedPolicy, np_policy_act)
    rep['device_fused_policy'] = run_device(n, T, dev, fused_policy=True)
    # (round 5 also replayed {fused policy, env.step} from a HIP graph: 80.9 vs 68.4 us eager at 3 000 envs, 145 vs 124 at 10^4 — with the policy
    # ONE launch a graph has nothing left to amortise, and its four staging copies + the replay cost more than the two launches they replace.
    # Removed in round 6; `device_graph` above stays: with the ~110-kernel torch policy the graph halves the step.)
    rep['device_fused_policy']['speedup_vs_torch_policy'] = rep['device']['us_per_step_wall'] / rep['device_fused_policy']['us_per_step_wall']
    rep['device_collector_i8_noise_block16'] = run_collector(n, T, dev, noise_block=16)   # the normal draws of 16 steps from one randn
    rep['device_collector_i8'] = run_collector(n, T, dev)   # the same three launches behind neuralplane_amd.collect.DeviceCollector (addresses pre-bound)
    rep['device_fused_policy_i8'] = run_device(n, T, dev, fused_policy=True, policy_numerics='i8')   # both networks in the block-fixed-point numerics
    rep['device_fused_policy_i8']['speedup_vs_torch_policy'] = rep['device']['us_per_step_wall'] / rep['device_fused_policy_i8']['us_per_step_wall']
    if n == 3000:
        # the 1v1 combat env with the self-play runner's two policies (15 observations): BASELINE config 5's share of one GPU of eight
        E, Ts = 12500, max(20, T // 4)
        rep['selfplay_e12500_torch_policies'] = run_selfplay(E, Ts, dev)
        rep['selfplay_e12500_fused_policies'] = run_selfplay(E, Ts, dev, fused_policy=True)
        rep['selfplay_e12500_fused_policies']['speedup_vs_torch_policies'] = (rep['selfplay_e12500_torch_policies']['us_per_step_wall'] /
                                                                               rep['selfplay_e12500_fused_policies']['us_per_step_wall'])
    if n == 10000:
        # the configuration the reference runs 10 000 rollout threads on (scripts/train_tracking.sh): PlanningEnv macro-steps, a 3-action policy
        Tt = max(10, T // 10)
        rep['tracking_torch_policy'] = run_device(n, Tt, dev, task='tracking')
        rep['tracking_fused_policy'] = run_device(n, Tt, dev, task='tracking', fused_policy=True)
        rep['tracking_fused_policy']['speedup_vs_torch_policy'] = rep['tracking_torch_policy']['us_per_step_wall'] / rep['tracking_fused_policy']['us_per_step_wall']
    d = rep['device']
    parts = {'policy (eager torch, ~110 small kernels)': d['gpu_us_policy'], 'env.step kernel': d['gpu_us_env_step'], 'insert (one launch)': d['gpu_us_insert'],
             'host gaps (GPU idle)': max(0.0, d['host_gap_us'])}
    rep['dominant_part_device_loop'] = max(parts, key=parts.get)
    rep['note'] = ('reference loop shape: runner/F16sim_runner.py:52-66,123-154; buffer_size there is 3 000 (heading) / 100 (tracking): the per-step figures '
                   'do not depend on it, compute_returns_ms is for the T timed here')
    return rep


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, nargs='*', default=[3000, 10000])
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--out', default=None)
    ap.add_argument('--only', default=None, choices=['fused', 'torch', 'selfplay', 'collector'], help='profiling: run just the device loop with the fused / the eager torch policy')
    args = ap.parse_args()
    if args.only == 'selfplay':
        rep = {'selfplay_e12500': {'torch_policies': run_selfplay(12500, args.steps, 'cuda:0'), 'fused_policies': run_selfplay(12500, args.steps, 'cuda:0', fused_policy=True)}}
    elif args.only == 'collector':
        rep = {f'collect_loop_n{n}': {'device_collector_i8': run_collector(n, args.steps, 'cuda:0'), 'device_collector_i8_noise_block16': run_collector(n, args.steps, 'cuda:0', noise_block=16), 'device_fused_policy_i8': run_device(n, args.steps, 'cuda:0', fused_policy=True, policy_numerics='i8')} for n in args.n}
    elif args.only:
        rep = {f'collect_loop_n{n}': {'device_fused_policy' if args.only == 'fused' else 'device': run_device(n, args.steps, 'cuda:0', fused_policy=args.only == 'fused')}
               for n in args.n}
    else:
        rep = {f'collect_loop_n{n}': collect_loop_report(n, args.steps, 'cuda:0') for n in args.n}
    txt = json.dumps(rep, indent=1)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(txt + '\n')
    print(txt)
