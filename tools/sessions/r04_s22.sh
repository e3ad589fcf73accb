#!/bin/bash
# round 4, GPU session 22: the library with the dual workgroups as the automatic choice between 1.5 and 2 tiles per CU — whole GPU suite, the
# driver's bench command, the planning sizes around the switch-overs
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s22; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gputest.log 2>&1; echo "gpu suite rc=$?"; tail -3 $out/gputest.log; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -80
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_s22/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
for k, v in d.get('optional_modes', {}).items():
    if isinstance(v, dict) and 'value' in v: print(k, v['value'], v.get('roofline', {}).get('frac'), v.get('launch_by_launch', {}).get('ms'))
PY
for cfg in "8192 20 0 auto" "10000 20 0 auto" "12288 20 0 auto" "12320 20 0 auto" "14000 20 0 auto" "16384 20 0 auto" "16416 20 0 auto"; do
  timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
