#!/bin/bash
# round 4, GPU session 11: the library with the CU-count-aware dispatch (np_dispatch.h, a.cus in the kernels), the one-launch prelude of
# PlanningEnv.step and the guest schedule as the automatic choice — the whole GPU suite, the driver's bench command, planning timings
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s11; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gputest.log 2>&1; echo "gpu suite rc=$?"; tail -3 $out/gputest.log; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -80
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_s11/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['frac'])
for k, v in d.get('optional_modes', {}).items():
    if isinstance(v, dict) and 'value' in v: print(k, v['value'], v.get('roofline', {}).get('frac'))
PY
for cfg in "3000 20 0 auto" "8192 20 0 auto" "9000 20 0 auto" "10000 20 0 auto" "10000 20 0 launches" "12288 20 0 auto" "16384 20 0 auto"; do
  timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
