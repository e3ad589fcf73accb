"""np_rollout_returns alone: time per call and the HBM rate of its algorithmic bytes (GAE: 12 B read + 4 B written per element;
with proper time limits 16 + 4), next to the reference's numpy loop on the host for the same shape."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from neuralplane_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for T, N in ((16, 1_000_000), (64, 1_000_000), (512, 131072), (100, 10000), (3000, 3000), (3000, 10000)):
    r = torch.randn(T, N, device='cuda'); v = torch.randn(T + 1, N, device='cuda'); m = (torch.rand(T + 1, N, device='cuda') > 0.1).float()
    b = (torch.rand(T + 1, N, device='cuda') > 0.1).float(); nv = torch.randn(N, device='cuda'); ret = torch.zeros(T + 1, N, device='cuda')
    for proper in (0, 1):
        call = lambda: lib.np_rollout_returns(T, N, 0.99, 0.95, 1, proper, r.data_ptr(), v.data_ptr(), m.data_ptr(), b.data_ptr(), nv.data_ptr(), ret.data_ptr(), 0, st)
        for _ in range(3): call()
        torch.cuda.synchronize(); t0 = time.perf_counter(); K = 20
        for _ in range(K): call()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        nbytes = T * N * (16 + 4 * proper)
        print(f'T={T} N={N} gae proper={proper}: {dt*1e3:.3f} ms per call, {nbytes/dt/1e12:.2f} TB/s algorithmic')
    if T * N <= 64_000_000:
        rn, vn, mn = r.cpu().numpy(), v.cpu().numpy(), m.cpu().numpy()
        t0 = time.perf_counter(); gae = 0; out = np.zeros_like(vn)
        for s in reversed(range(T)):
            td = rn[s] + 0.99 * vn[s + 1] * mn[s + 1] - vn[s]; gae = td + 0.99 * 0.95 * mn[s + 1] * gae; out[s] = gae + vn[s]
        print(f'   numpy loop of the reference on the host (same shape): {(time.perf_counter()-t0)*1e3:.1f} ms')
