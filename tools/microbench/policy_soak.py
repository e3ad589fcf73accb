#!/usr/bin/env python3
"""Soak of the policy step: random networks (1..4 actions, 22 / 15 observations, weight scales 0.05 .. 8, both numerics), random ragged batch sizes,
chained steps with masks — every output of np_policy_act against the CPU restatement, bit for bit.   python tools/microbench/policy_soak.py [cases]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuralplane_amd.policy import FusedPolicy, pack_policy_actor, pack_policy_critic  # noqa: E402
from oracle.f16_oracle import PolicyOracle  # noqa: E402
from tests.policy_kat import random_state_dicts  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.RandomState(2025)
    t0, rows, bad = time.time(), 0, 0
    for c in range(cases):
        act_dim, obs_dim = int(rng.randint(1, 5)), int(rng.choice([22, 15]))
        scale, numerics = float(rng.choice([0.05, 0.3, 1.0, 3.0, 8.0])), ('fp32', 'i8')[c % 2]
        n = int(rng.choice([1, 31, 32, 33, 64, 1000, 3000, 4097, 10000, 20001, 40000]))
        sa, sc = random_state_dicts(act_dim, 1000 + c, scale, obs_dim)
        fp = FusedPolicy((sa, sc), 'cuda:0', numerics=numerics)
        o = PolicyOracle(pack_policy_actor(sa)[0], pack_policy_critic(sc), np.float32(fp.std), np.float32(fp.log_std), numerics, obs_dim)
        ha = hc = np.zeros((n, 128), np.float32)
        dha = dhc = torch.zeros((n, 1, 128), device='cuda:0')
        ok = True
        for t in range(3):
            obs = (rng.normal(0, 1, (n, obs_dim)) * rng.uniform(0.05, 20, (1, obs_dim))).astype(np.float32)
            mk = (rng.uniform(0, 1, (n, 1)) > 0.15).astype(np.float32)
            eps = rng.normal(0, 1, (n, act_dim)).astype(np.float32)
            flags = [3, 3, 1, 2, 5][int(rng.randint(0, 5))]          # get_actions (twice as likely), act, get_values, act deterministic
            if flags == 3:
                v, a, lp, dha, dhc = fp.get_actions(torch.from_numpy(obs).cuda(), dha, dhc, torch.from_numpy(mk).cuda(), noise=torch.from_numpy(eps).cuda())
                v_o, a_o, lp_o, ha, hc = o.run(obs, ha, hc, mk, eps)
                got, ref = (v, a, lp, dha, dhc), (v_o, a_o, lp_o, ha, hc)
            elif flags == 2:
                v = fp.get_values(torch.from_numpy(obs).cuda(), dhc, torch.from_numpy(mk).cuda())
                got, ref = (v,), (o.run(obs, ha, hc, mk, flags=2)[0],)
            else:
                a, h2 = fp.act(torch.from_numpy(obs).cuda(), dha, torch.from_numpy(mk).cuda(), deterministic=flags == 5, noise=torch.from_numpy(eps).cuda())
                r = o.run(obs, ha, hc, mk, eps, flags=flags)
                got, ref = (a, h2), (r[1], r[3])
            for x, y in zip(got, ref):
                ok = ok and np.array_equal(x.cpu().numpy().reshape(y.shape), y) and bool(np.all(np.isfinite(y)))
            rows += n
        bad += 0 if ok else 1
        print(f'case {c:3d}: {numerics:4s} obs {obs_dim} act {act_dim} scale {scale:<4} n {n:6d}  {"equal" if ok else "MISMATCH"}', flush=True)
    print(f'{cases} cases, {rows} row-steps, {bad} mismatches, {time.time() - t0:.0f} s')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
