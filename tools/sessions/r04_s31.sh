#!/bin/bash
# round 4, GPU session 31: final state — build() + smoke() as the driver runs them, the whole GPU suite, the driver's bench command timed
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s31; mkdir -p $out
( time python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" ) > $out/smoke.log 2>&1; tail -4 $out/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gputest.log 2>&1; echo "gpu suite rc=$?"; tail -3 $out/gputest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err ) 2>&1 | tail -3; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_s31/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')})
print(d['config']); print(d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
for k, v in d.get('optional_modes', {}).items():
    if isinstance(v, dict) and 'value' in v: print(k, v['value'], v.get('roofline', {}).get('frac'), v.get('inner_loop'))
PY
