"""PlanningEnv.step with the fused controller, for rocprofv3 --kernel-trace --stats: which kernel takes what inside a macro-step.
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o p -- python tools/microbench/planning_profile.py 8192 [steps [groups]]"""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, '.')
from neuralplane_amd.envs.planning_env import PlanningEnv
from neuralplane_amd.actor import FusedActor, NUM_FLOATS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cuda:0', controller=FusedActor(w, 'cuda:0', numerics=os.environ.get('NUMERICS', 'i8')))
env.loop_groups = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # np_planning_loop.groups (0 = the library chooses)
env.loop_mode = sys.argv[4] if len(sys.argv) > 4 else 'auto'        # auto | launches | persistent | queue
env.loop_waves = int(sys.argv[5]) if len(sys.argv) > 5 else 0
env.loop_block = int(sys.argv[6]) if len(sys.argv) > 6 else 0
a = torch.rand(n, 3, device='cuda') * 2 - 1
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    env.step(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    env.step(a)
torch.cuda.synchronize()
print(f'numerics={os.environ.get("NUMERICS", "i8")} n={n} groups={env.loop_groups} mode={env.loop_mode} waves={env.loop_waves} block={env.loop_block}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per PlanningEnv.step (np_planning_inner_loop)')
