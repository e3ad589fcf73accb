mkdir -p gpurun_out/s9
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/s9/pytest.log 2>&1
tail -5 gpurun_out/s9/pytest.log
python3 tools/microbench/ab_libs.py --rounds 2 > gpurun_out/s9/ab.log 2>&1
tail -4 gpurun_out/s9/ab.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s9/bench_driver.json 2> gpurun_out/s9/bench_driver.err
python3 -c "
import json
d=json.loads(open('gpurun_out/s9/bench_driver.json').read().strip().splitlines()[-1])
print('value %.4e kern %.4f frac %.4f cold %.4e'%(d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['cold_start']['value']))
for k,v in d['optional_modes'].items(): print(k, v.get('value'), v.get('kernel_avg_ms', v.get('kernel_avg_us')))
"
