#!/bin/bash
# round 3, final check of the committed tree: smoke(), the full GPU suite, the driver's bench command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_final; mkdir -p $out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench.err ) 2>&1 | grep real
python -c "
import json
d=json.loads([l for l in open('$out/bench_driver_cmd.json') if l.startswith('{')][-1])
print('value %.4e kernel %.4f frac %.3f' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac']))"
