import sys, time, os, torch
sys.path.insert(0, '.')
from neuralplane_amd.envs.control_env import ControlEnv
for n in (163840, 229376, 327680, 524288):
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=0, device='cuda:0', solver='rk4')
    env.reset()
    a = torch.rand(n, 4, device='cuda') * 2 - 1
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end: env.step(a)
    env._batch.set_timing(True)
    for _ in range(200): env.step(a)
    torch.cuda.synchronize()
    smp = sorted(env._batch.get_timing_samples()); env._batch.set_timing(False)
    print(f"rk4 PAIR_WAVES={os.environ.get('NPF16_PAIR_WAVES','auto')} N={n} kernel {smp[len(smp)//2]*1e3:.1f} us", flush=True)
    del env
