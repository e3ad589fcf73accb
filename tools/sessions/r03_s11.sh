#!/bin/bash
# round 3, GPU session 11: PlanningEnv with the low-level observation written by the inner step — parity tests, macro-step time at
# n = 1e4 and 262 144 (roofline block), parity report on the HIP engine, full bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s11; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_actor.py tests/test_gpu_edge_cases.py -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 600 python tools/parity_report.py --engine hip --out $out/parity_hip.json > /dev/null 2> $out/parity.err; tail -2 $out/parity.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench.err < /dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_s11/bench_driver_cmd.json') if l.startswith('{')][-1])
print('value %.4e ms %.4f kernel %.4f frac %.3f exec %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['executed_frac']))
for k,v in d['optional_modes'].items():
    print(k, v.get('value'), v.get('unit','')[:40], (v.get('roofline') or {}).get('frac'), v.get('env_kernels_ms_per_macro_step'))
PY
