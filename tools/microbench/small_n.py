import sys, time, torch
sys.path.insert(0, '.')
from neuralplane_amd.envs.control_env import ControlEnv
for n in (256, 4096, 65536):
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0')
    env.reset()
    a = torch.rand(n, 4, device='cuda') * 2 - 1
    for _ in range(50): env.step(a)
    env._batch.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 2000
    for _ in range(K): env.step(a)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, cnt = env._batch.get_timing()
    print(f'N={n}: {dt/K*1e6:.1f} us/step wall ({t_host/K*1e6:.1f} us host enqueue), kernel {ms*1e3:.1f} us -> {n*K/dt:.3e} aircraft-steps/s')
