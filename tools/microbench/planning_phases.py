"""Phase breakdown of one iteration of the persistent PlanningEnv kernel (np_planning.hip): shader-clock stamps of the last workgroup
in its last-but-one iteration (waves 0, 1 and 4) plus the controller's own phase stamps (last iteration, wave 0).

    python tools/microbench/planning_phases.py build            # in the build container: tools/microbench/libs/plan_trace.so
    NPF16_LIB=tools/microbench/libs/plan_trace.so python tools/microbench/planning_phases.py 8192 [waves [mode]]     # on the GPU box
NUMERICS=fp32|i8 selects the controller.  For the block-fixed-point controller (i8) only the OUTER phases are meaningful and even those are
inflated: a stamp orders the memory operations around it, and the i8 call lives on its weight prefetch (a traced build ran an iteration in
152 K cycles against 67 K untraced) — its inner breakdown comes from timing-only builds with parts removed: tools/microbench/i8_actor_phases.sh.
"""
import ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIB = os.path.join(HERE, 'libs', 'plan_trace.so')

if len(sys.argv) > 1 and sys.argv[1] == 'build':
    from neuralplane_amd import build as nb
    nb.build_hip()
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    obj = os.path.join(HERE, 'libs', 'np_planning_trace.o')
    subprocess.run([nb._hipcc()] + nb.FLAGS + ['-DNP_PLAN_TRACE=1', '-DNPACT_TRACE=1'] + sys.argv[2:] + ['-c', '-o', obj, os.path.join(nb.CSRC, 'np_planning.hip')], check=True, cwd=nb.CSRC,
                   stderr=subprocess.DEVNULL)
    subprocess.run([nb._hipcc(), '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB] + [os.path.join(nb.OBJ_DIR, os.path.splitext(f)[0] + '.o') for f in nb.SOURCES if f != 'np_planning.hip'] + [obj], check=True)  # every shipped unit but the planning one
    print('built', LIB)
    sys.exit(0)

import numpy as np, torch
from neuralplane_amd import _lib
from neuralplane_amd.envs.planning_env import PlanningEnv
from neuralplane_amd.actor import FusedActor, NUM_FLOATS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else 'persistent'
w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cuda:0', controller=FusedActor(w, 'cuda:0', numerics=os.environ.get('NUMERICS', 'i8')))
env.loop_mode, env.loop_waves = mode, waves
lib = _lib.load()
lib.np_plan_trace_read.argtypes = [C.c_void_p, C.c_void_p]
a = torch.rand(n, 3, device='cuda') * 2 - 1
PN = ['entry -> controller', 'controller (waves 4..7: the previous step\'s back)', 'barrier', 'FDM: state + u lag', 'FDM: integrator evaluation (REST nets) + Euler',
      'FDM: Overload evaluation | pipelined: new-state chains + barrier', 'FDM: terminations / next observation', 'end barrier']
I8 = os.environ.get('NUMERICS', 'i8') == 'i8'
AN8 = ['(entry)', 'obs LayerNorm + quantise', 'L1 (1 k-step) + epilogue', 'LN1 + quantise + barrier', 'L2 + epilogue', 'LN2 + 2 x quantise + barrier', 'gi_z', 'gh_z', 'sigmoid z',
       'gi_r + gh_r', 'sigmoid r', 'gh_n + gi_n', 'tanh + blend', 'LN3 + quantise + barrier', 'A1 + epilogue', 'LN4 + quantise + barrier', 'A2 + epilogue', 'LN5', 'head']
AN = ['(entry)', 'obs LayerNorm + prefetch', 'L1 (22 MFMAs) + transpose', 'LN1', 'L2 dense', 'transpose + LN2', 'h -> LDS + gi_r', 'gh_r', 'sigmoid r',
      'gi_z + gh_z', 'sigmoid z', 'gi_n + gh_n', 'gates + barrier', 'transpose + LN3', 'A1 dense', 'transpose + LN4', 'A2 dense', 'transpose + LN5', 'head']
acc, acc_a, K = np.zeros((8, 8)), np.zeros(18 if os.environ.get('NUMERICS', 'i8') == 'i8' else 17), 10
buf, abuf = (C.c_ulonglong * 128)(), (C.c_longlong * 64)()
for it in range(K + 3):
    env.step(a)
    torch.cuda.synchronize()
    assert lib.np_plan_trace_read(buf, abuf) == 0
    t = np.array(buf[:], dtype=np.float64).reshape(8, 16)
    ta = np.array(abuf[:19], dtype=np.float64)
    if it >= 3:
        # a stamp a schedule never writes stays 0 (device globals are zero-initialised): differences next to it are not phases (r04 printed them)
        d, da = np.diff(t[:, :9], axis=1), np.diff(ta[0:19] if I8 else ta[1:19])
        d[(t[:, :8] == 0) | (t[:, 1:9] == 0)] = np.nan
        tb = ta[0:19] if I8 else ta[1:19]
        da[(tb[:-1] == 0) | (tb[1:] == 0)] = np.nan
        acc += d
        acc_a += da[:len(acc_a)] if len(da) >= len(acc_a) else np.pad(da, (0, len(acc_a) - len(da)), constant_values=np.nan)
acc /= K; acc_a /= K
print(f'controller numerics {os.environ.get("NUMERICS", "i8")}, n = {n}, {waves} waves per tile, mode {mode}: cycles per phase (wave 0 | wave 1 | wave {4 if waves == 8 else 3}), one iteration = {np.nansum(acc[0]):.0f} cycles')
for k, name in enumerate(PN):
    print(f'   {name:52s} {acc[0, k]:8.0f} {acc[1, k]:8.0f} {acc[4 if waves == 8 else 3, k]:8.0f}')
print(f"   inside the controller call (wave 0, last iteration): {np.nansum(acc_a):.0f} cycles from the obs LayerNorm to the head's end")
for name, c in zip((AN8 if I8 else AN)[1:], acc_a):
    print(f'      {name:32s} {c:8.0f}')
