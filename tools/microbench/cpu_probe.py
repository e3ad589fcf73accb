import os, time, subprocess, sys
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
for f in ['/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us']:
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
print(open('/proc/loadavg').read())
code = '''
import os, time, numpy as np, sys
sys.path.insert(0,'.')
from oracle.f16_oracle import Oracle
n=32768
o = Oracle('heading'); st = Oracle.new_state(n)
a = np.random.RandomState(0).uniform(-1,1,(n,4)).astype(np.float32)
o.reset(st); o.step(st, a, call_idx=1)
t0=time.perf_counter(); k=0
while time.perf_counter()-t0 < 2: o.step(st, a, call_idx=2+k); k+=1
print(os.environ.get('OMP_NUM_THREADS'), 'threads', o.threads, 'rate %.3e'%(n*k/(time.perf_counter()-t0)))
'''
for t in [1, 4, 8, 16, 32, 64, 128]:
    env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND='false')
    print(subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True).stdout.strip())
