#!/bin/bash
# round 4, GPU session 17: the library with the self-validating coefficient cache as shipped (key loads pinned in front of the coefficient
# loads, lanes beyond the batch exempt) — the whole GPU suite, the driver's bench command, the planning timings
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s17; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gputest.log 2>&1; echo "gpu suite rc=$?"; tail -3 $out/gputest.log; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -80
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_s17/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['frac'], d['roofline'].get('kernel_avg_ms'))
for k, v in d.get('optional_modes', {}).items():
    if isinstance(v, dict) and 'value' in v: print(k, v['value'], v.get('roofline', {}).get('frac'), v.get('launch_by_launch', {}).get('ms'))
PY
