"""Experiment: PlanningEnv's 50 inner iterations for TWO half-batches on two streams (eager launches), so that one half's
controller call overlaps the other's FDM step and — at 8 192 < n <= 16 384 — every controller call finds one tile per CU.
    python tools/microbench/planning_two_streams.py 10000"""
import sys, time, torch
import numpy as np
sys.path.insert(0, '.')
from neuralplane_amd.envs.planning_env import PlanningEnv, INNER_STEPS
from neuralplane_amd.actor import FusedActor, NUM_FLOATS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
dev = 'cuda:0'


def macro(envs, streams, acts):
    """the body of PlanningEnv.step for several envs, iteration by iteration, each env on its own stream"""
    st = []
    for e, s, a in zip(envs, streams, acts):
        with torch.cuda.stream(s):
            b = e._batch
            b.reset(want_obs=False)
            a = torch.clamp(a, -1, 1)
            tgt3 = torch.stack((b.s[4] + a[:, 0] * 0.3, b.s[5] + a[:, 1] * 0.3, b.s[6] + a[:, 2] * 30)).contiguous()
            st.append([tgt3, b.lowlevel_obs(tgt3), torch.empty((e.n, 22), device=dev)])
    for k in range(INNER_STEPS):
        last = k == INNER_STEPS - 1
        for e, s, x in zip(envs, streams, st):
            with torch.cuda.stream(s), torch.no_grad():
                act, _, e.ego_rnn_states = e.controller(x[1], e.ego_rnn_states, e._masks, deterministic=True)
                e._batch.step(act, inner=True, ll_tgt=None if last else x[0], ll_obs=None if last else x[2], want_obs=last)
                x[1], x[2] = x[2], x[1]


def bench(sizes, label):
    envs = [PlanningEnv(num_envs=m, config='tracking', model='F16', random_seed=i, device=dev, controller=FusedActor(w, dev)) for i, m in enumerate(sizes)]
    for e in envs:
        e._masks = torch.ones((e.n, 1), device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in envs]
    acts = [torch.rand(e.n, 3, device=dev) * 2 - 1 for e in envs]
    for _ in range(20):
        macro(envs, streams, acts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        macro(envs, streams, acts)
    torch.cuda.synchronize()
    print(f'{label}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per macro-step of {sum(sizes)} aircraft', flush=True)


bench([n], 'one stream ')
bench([n // 2, n - n // 2], 'two streams')
bench([n // 3, n // 3, n - 2 * (n // 3)], 'three streams')
