"""Does replaying PlanningEnv.step's three launches (reset, prelude, persistent kernel) from a HIP graph shorten the macro-step?
Timing probe only: the captured call index is frozen, so the replayed steps repeat the first step's random draws.
    python tools/microbench/planning_graph_probe.py [n]"""
import sys, time, torch
import numpy as np
sys.path.insert(0, '.')
from neuralplane_amd.envs.planning_env import PlanningEnv
from neuralplane_amd.actor import FusedActor, NUM_FLOATS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cuda:0', controller=FusedActor(w, 'cuda:0'))
env.loop_mode = 'persistent'
a = torch.rand(n, 3, device='cuda') * 2 - 1
for _ in range(5):
    env.step(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40):
    env.step(a)
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 40
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    env.step(a)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    env.step(a)
for _ in range(5):
    g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40):
    g.replay()
torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 40
print(f'n={n}: eager {eager * 1e3:.3f} ms per PlanningEnv.step, graph replay {graph * 1e3:.3f} ms')
