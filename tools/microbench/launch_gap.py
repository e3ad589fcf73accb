import sys, time, torch
sys.path.insert(0,'.')
from neuralplane_amd.envs.control_env import ControlEnv
n=1_000_000
env=ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0')
pool=[torch.rand(n,4,device='cuda')*2-1 for _ in range(8)]
env.reset()
for i in range(1500): env.step(pool[i%8])
for timed in (False, True, False, True):
    env._batch.set_timing(timed)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(200): env.step(pool[i%8])
    torch.cuda.synchronize(); el=time.perf_counter()-t0
    k = env._batch.get_timing()[0] if timed else float('nan')
    print('events' if timed else 'no events', 'wall ms/step %.4f' % (el/200*1e3), 'kernel %.4f' % k, flush=True)
    env._batch.set_timing(False)
# raw C-ABI loop without allocating outputs each step is what bench does? use batch.step directly
b=env._batch
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(200): b.step(pool[i%8])
torch.cuda.synchronize(); print('batch.step wall ms/step %.4f' % ((time.perf_counter()-t0)/200*1e3))
