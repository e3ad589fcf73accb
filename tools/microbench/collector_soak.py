#!/usr/bin/env python3
"""Soak of neuralplane_amd.collect.DeviceCollector (in_place: no insert launch) against the three-call collect step over whole rollouts:
the reference's loop shape (runner/F16sim_runner.py:52-66) — T steps, compute_returns, after_update — repeated, storage and flight state
compared bit for bit after every rollout.   python tools/microbench/collector_soak.py [n] [T] [rollouts]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuralplane_amd.buffer import DeviceReplayBuffer  # noqa: E402
from neuralplane_amd.collect import DeviceCollector  # noqa: E402
from neuralplane_amd.envs.control_env import ControlEnv  # noqa: E402
from neuralplane_amd.envs.env_wrappers import DeviceVecEnv  # noqa: E402
from neuralplane_amd.policy import FusedPolicy  # noqa: E402
from tests.policy_kat import random_state_dicts  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 6

    class Args:
        buffer_size, n_rollout_threads = T, n
        gamma, use_proper_time_limits, use_gae, gae_lambda = 0.99, True, True, 0.95
        recurrent_hidden_size, recurrent_hidden_layers = 128, 1
    sds = random_state_dicts(4, 31)
    sds[0]['act.action_out.mu_net.fc.0.weight'] *= 3.0

    def make():
        envs = DeviceVecEnv([lambda: ControlEnv(num_envs=n, config='heading', model='F16', random_seed=11, device='cuda:0')])
        buf = DeviceReplayBuffer(Args, 1, envs.observation_space, envs.action_space, device='cuda:0')
        buf.obs[0].copy_(envs.reset())
        return FusedPolicy(sds, 'cuda:0'), envs, buf
    pa, ea, ba = make()
    pb, eb, bb = make()
    col = DeviceCollector(pb, eb, bb)
    assert col.in_place
    bad, t0 = 0, time.time()
    for r in range(R):
        torch.manual_seed(100 + r)
        for _ in range(T):
            s = ba.step
            v, a, lp, ha, hc = pa.get_actions(ba.obs[s].reshape(n, -1), ba.rnn_states_actor[s].reshape(n, 128), ba.rnn_states_critic[s].reshape(n, 128),
                                              ba.masks[s].reshape(n, 1))
            obs, rew, d, bd, tm, _ = ea.step(a)
            ba.insert_step(obs, a, rew, d, bd, tm, lp, v, ha, hc)
        ba.compute_returns(pa.get_values(ba.obs[-1].reshape(n, -1), ba.rnn_states_critic[-1].reshape(n, 128), ba.masks[-1].reshape(n, 1)).reshape(n, 1, 1))
        torch.manual_seed(100 + r)
        for _ in range(T):
            col.step()
        col.compute_returns()
        same = all(torch.equal(getattr(ba, k), getattr(bb, k)) for k in ba._STORAGE) and torch.equal(ba.returns, bb.returns) and torch.equal(ea.env.model.s, eb.env.model.s)
        ended = int((ba.masks[1:] == 0).sum()), int((ba.bad_masks[1:] == 0).sum())
        print(f'rollout {r}: {T} steps x {n} envs, episodes ended done / bad {ended}, storage + returns + flight state {"equal" if same else "MISMATCH"}', flush=True)
        bad += 0 if same else 1
        ba.after_update()
        bb.after_update()
    print(f'{R} rollouts, {bad} mismatches, {time.time() - t0:.0f} s')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
