"""Experiment: PlanningEnv's 50 inner iterations of G row groups enqueued from C on G plain streams (tools/microbench/replay/replay.c)
— does one group's controller call overlap another's FDM step when the host is not the limit?
    python tools/microbench/planning_replay.py 10000 [groups ...]"""
import ctypes as C, os, subprocess, sys, time, torch
import numpy as np
sys.path.insert(0, '.')
from neuralplane_amd import _lib
from neuralplane_amd.envs.planning_env import PlanningEnv, INNER_STEPS
from neuralplane_amd.actor import FusedActor, NUM_FLOATS, HID

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'replay')
so = os.path.join(here, 'replay.so')
if not os.path.exists(so):
    subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(here, 'replay.c')])
rp = C.CDLL(so)


class Item(C.Structure):
    _fields_ = [('kind', C.c_int), ('device', C.c_int), ('ctx', C.c_void_p), ('n', C.c_longlong), ('io', C.c_void_p), ('stream', C.c_void_p),
                ('w', C.c_void_p), ('nf', C.c_longlong), ('obs', C.c_void_p), ('hin', C.c_void_p), ('mask', C.c_void_p), ('act', C.c_void_p), ('hout', C.c_void_p)]


n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
groups_list = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
dev = 'cuda:0'
w = np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
lib = _lib.load()
step_p = C.cast(lib.np_f16_step, C.c_void_p).value
actor_p = C.cast(lib.np_actor_forward, C.c_void_p).value
rp.replay.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]

for G in groups_list:
    sizes = [n // G + (1 if g < n % G else 0) for g in range(G)]
    if os.environ.get('NP_REPLAY_SIZES'):      # explicit group sizes, e.g. 8192,1808
        sizes = [int(x) for x in os.environ['NP_REPLAY_SIZES'].split(',')]
        n, G = sum(sizes), len(sizes)
    envs = [PlanningEnv(num_envs=m, config='tracking', model='F16', random_seed=g, device=dev, controller=FusedActor(w, dev)) for g, m in enumerate(sizes)]
    streams = [torch.cuda.Stream(device=dev) for _ in envs]
    keep, per_group = [], []
    for e, s in zip(envs, streams):
        b, m = e._batch, e.n
        e.step(torch.rand(m, 3, device=dev) * 2 - 1)       # a valid state, cache filled
        torch.cuda.synchronize()
        tgt3 = torch.stack((b.s[4], b.s[5], b.s[6])).contiguous()
        ll = [b.lowlevel_obs(tgt3), torch.empty((m, 22), device=dev)]
        h = [e.ego_rnn_states.reshape(m, HID).contiguous().clone(), torch.empty((m, HID), device=dev)]
        act = torch.empty((m, 4), device=dev); mask = torch.ones(m, device=dev); rew = torch.empty(m, device=dev); obs = torch.empty((m, 22), device=dev)
        fl = [b.flags.contiguous().clone(), torch.empty((3, m), dtype=torch.uint8, device=dev)]
        items = []
        for k in range(INNER_STEPS):
            last = k == INNER_STEPS - 1
            items.append(Item(1, 0, None, m, None, s.cuda_stream, e.controller.weights.data_ptr(), NUM_FLOATS, ll[k % 2].data_ptr(), h[k % 2].data_ptr(),
                              mask.data_ptr(), act.data_ptr(), h[1 - k % 2].data_ptr()))
            b.flags = fl[k % 2]
            io = b._io(fl[1 - k % 2], act, obs if last else None, rew, None, None, inner=True, ll_tgt=None if last else tgt3, ll_obs=None if last else ll[1 - k % 2])
            io.cache_valid = 1
            cp = _lib.NpF16Io(); C.memmove(C.byref(cp), C.byref(io), C.sizeof(io)); keep.append(cp)
            items.append(Item(0, 0, b._ctx, m, C.addressof(cp), s.cuda_stream, None, 0, None, None, None, None, None))
        per_group.append(items)
        keep += [tgt3, ll, h, act, mask, rew, obs, fl]
    # interleave the groups iteration by iteration (actor g0, step g0, actor g1, step g1, ...), offset by half an iteration would be the
    # GPU's business: streams are independent
    order = []
    for k in range(INNER_STEPS):
        for items in per_group:
            order += items[2 * k:2 * k + 2]
    arr = (Item * len(order))(*order)

    offs = int(os.environ.get('NP_REPLAY_OFFSET_CYCLES', '0'))   # start group g late by g x this many shader cycles (torch.cuda._sleep)

    def run():
        if offs:
            for g, s in enumerate(streams):
                if g:
                    with torch.cuda.stream(s):
                        torch.cuda._sleep(g * offs)
        rc = rp.replay(step_p, actor_p, C.addressof(arr), len(order))
        assert rc == 0, (rc, lib.np_last_error())
    for _ in range(10):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        run()
    t_host = (time.perf_counter() - t0) / K
    torch.cuda.synchronize()
    print(f'offset={offs} n={n} groups={G} {sizes}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per 50 inner iterations (host enqueue {t_host * 1e3:.3f} ms)', flush=True)
    del envs
