"""In-kernel phase timing of the latency variants (4 or 8 waves per 64-aircraft tile): 100 MHz stamps at the phase boundaries of
every wave.  Needs an experiment build:
    NPF16_EXTRA_FLAGS="-DNPF16_LAT_TRACE" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
    cp neuralplane_amd/csrc/libneuralplane_hip.so tools/microbench/libs/lat_trace.so      (then rebuild the normal library)
    gpurun -- 'NPF16_LIB=$PWD/tools/microbench/libs/lat_trace.so python3 tools/microbench/lat_trace.py 256 latency8'
"""
import ctypes as C, sys, torch, numpy as np
sys.path.insert(0,'.')
from neuralplane_amd import _lib
from neuralplane_amd.envs.control_env import ControlEnv
n=int(sys.argv[1]) if len(sys.argv)>1 else 256
variant=sys.argv[2] if len(sys.argv)>2 else 'latency8'
W=8 if variant=='latency8' else 4
env=ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0')
b=env._batch
b.set_kernel_variant(variant)
a=torch.rand(n,4,device='cuda')*2-1
env.reset()
for _ in range(300): env.step(a)
tiles=(n+63)//64
trace=torch.zeros((tiles*W*8+64,),dtype=torch.int64,device='cuda')
_lib.check(b.lib.np_f16_set_trace(b._ctx, C.c_void_p(trace.data_ptr()), tiles*W*2+16))
for _ in range(5): env.step(a)
torch.cuda.synchronize()
t=trace.cpu().numpy()[:tiles*W*8].reshape(tiles,W,8)
names=['start->pre-REST','REST nlplant','integrate+noise share','FORCE2 nlplant','epilogue','obs store']
for tile in range(min(tiles,2)):
    t0=t[tile,:,0].min()
    for w in range(W):
        r=t[tile,w]
        print('tile',tile,'wave',w,' '.join(f'{names[k]}={(r[k+1]-r[k])/100:.2f}' for k in range(6)), 'total us', (r[6]-t0)/100, 'start offs', (r[0]-t0)/100)
