"""numpy-in / numpy-out boundary (what the reference's runner calls): GPUVecEnv vs PinnedVecEnv vs DeviceVecEnv."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from neuralplane_amd.envs import env_wrappers as W
from neuralplane_amd.envs.control_env import ControlEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.RandomState(0)
acts = rng.uniform(-1, 1, (n, 1, 4)).astype(np.float32)
for name in ('GPUVecEnv', 'PinnedVecEnv', 'DeviceVecEnv'):
    cls = getattr(W, name, None)
    if cls is None:
        continue
    vec = cls([lambda: ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0')])
    vec.reset()
    a = torch.from_numpy(acts).cuda() if name == 'DeviceVecEnv' else acts
    for _ in range(3):
        vec.step(a)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 20
    for _ in range(K):
        out = vec.step(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f'{name}: N={n}: {dt*1e3:.2f} ms per step -> {n/dt:.3e} aircraft-steps/s')
