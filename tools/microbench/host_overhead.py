"""Host-side cost of one env.step at a small batch (where the 16 us kernel makes the Python path the co-bottleneck)."""
import ctypes as C, sys, time, torch
sys.path.insert(0, '.')
from neuralplane_amd import _lib
from neuralplane_amd.envs.control_env import ControlEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0')
b = env._batch
a = torch.rand(n, 4, device='cuda') * 2 - 1
env.reset()
for _ in range(300): env.step(a)
torch.cuda.synchronize()
def timeit(f, K=3000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6
print('env.step           host %.2f us  wall %.2f us' % timeit(lambda: env.step(a)))
print('F16Batch.step      host %.2f us  wall %.2f us' % timeit(lambda: b.step(a)))
obs = torch.empty((n, 22), device='cuda'); rew = torch.empty(n, device='cuda'); fl = [torch.empty((3, n), dtype=torch.uint8, device='cuda') for _ in range(2)]
io = b._io(fl[0], a, obs, rew, None, None)
st = b._stream()
k = [0]
def raw():
    fi, fo = fl[k[0] & 1].data_ptr(), fl[(k[0] + 1) & 1].data_ptr()
    io.done_in, io.bad_in, io.timeout_in = fi, fi + n, fi + 2 * n
    io.done_out, io.bad_out, io.timeout_out = fo, fo + n, fo + 2 * n
    io.call_idx = k[0]; k[0] += 1
    b.lib.np_f16_step(b._ctx, n, C.byref(io), st)
print('raw ctypes launch  host %.2f us  wall %.2f us' % timeit(raw))
print('3 x torch.empty    host %.2f us' % timeit(lambda: (torch.empty((n, 22), device='cuda'), torch.empty(n, device='cuda'), torch.empty((3, n), dtype=torch.uint8, device='cuda')))[0])
print('1 x torch.empty    host %.2f us' % timeit(lambda: torch.empty(n * 26, device='cuda'))[0])
print('current_stream     host %.2f us' % timeit(lambda: torch.cuda.current_stream(b.device).cuda_stream)[0])
