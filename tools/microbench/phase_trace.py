"""Per-wave phase times of the pair / throughput kernel in steady state (needs a -DNPF16_LAT_TRACE build, see lat_trace.py)."""
import ctypes as C, sys, torch, numpy as np
sys.path.insert(0, '.')
from neuralplane_amd import _lib
from neuralplane_amd.envs.control_env import ControlEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0')
b = env._batch
if len(sys.argv) > 2: b.set_kernel_variant(sys.argv[2])
pool = [torch.rand(n, 4, device='cuda') * 2 - 1 for _ in range(4)]
env.reset()
for i in range(400): env.step(pool[i % 4])
wgs = (n + 127) // 128
trace = torch.zeros((wgs * 2 * 8 + 64,), dtype=torch.int64, device='cuda')
_lib.check(b.lib.np_f16_set_trace(b._ctx, C.c_void_p(trace.data_ptr()), wgs * 4))
for i in range(3): env.step(pool[i % 4])
torch.cuda.synchronize()
t = trace.cpu().numpy()[:wgs * 2 * 8].reshape(wgs * 2, 8).astype(np.float64) / 100.0
names = ['load+update', 'REST nlplant', 'integrate+trig+obs+noise', 'FORCE2 nlplant', 'accel+done+stores', 'obs store']
d = np.diff(t[:, :7], axis=1)
start = t[:, 0] - t[:, 0].min()
order = np.argsort(start)
mid = order[len(order) // 4: 3 * len(order) // 4] if wgs > 2048 else order   # waves of the steady state (all of them for one generation)
print('waves', len(t), 'total per wave median us', np.median(t[mid, 6] - t[mid, 0]))
for k, nm in enumerate(names):
    print(f'{nm:28s} median {np.median(d[mid, k]):6.2f} us   p10 {np.percentile(d[mid, k], 10):6.2f}  p90 {np.percentile(d[mid, k], 90):6.2f}')
