"""Phase breakdown of actor_forward_mfma32_kernel: shader-clock stamps of the last workgroup's first wave (a -DNPACT_TRACE=1
build selected with NPF16_LIB).  Prints cycles per phase, averaged over a few calls."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
from neuralplane_amd import _lib
from neuralplane_amd.actor import FusedActor, NUM_FLOATS
NAMES = ['loads + obs LayerNorm', 'L1 (22 MFMAs) + transpose', 'LN1', 'L2 dense', 'transpose + LN2', 'h -> LDS + gi_r', 'gh_r', 'sigmoid r',
         'gi_z + gh_z', 'sigmoid z', 'gi_n + gh_n', 'gates + barrier', 'transpose + h store + LN3', 'A1 dense', 'transpose + LN4', 'A2 dense',
         'transpose + LN5', 'head']
fa = FusedActor(np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32), 'cuda:0')
lib = _lib.load()
buf = (C.c_longlong * 64)()
for n in [int(x) for x in sys.argv[1:]] or [64, 4096, 8192, 10000]:
    obs = torch.randn(n, 22, device='cuda'); h = torch.zeros(n, 1, 128, device='cuda'); m = torch.ones(n, 1, device='cuda')
    acc = np.zeros(18)
    K = 20
    for it in range(K + 3):
        _, _, h = fa(obs, h, m)
        torch.cuda.synchronize()
        assert lib.np_actor_trace_read(buf) == 0
        t = np.array(buf[:19], dtype=np.float64)
        if it >= 3: acc += np.diff(t)
    acc /= K
    print(f'n = {n}: {acc.sum():.0f} cycles from entry to end')
    for name, c in zip(NAMES, acc): print(f'   {name:28s} {c:8.0f}')
