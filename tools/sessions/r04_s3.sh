#!/bin/bash
# round 4, GPU session 3: persistent kernel parity after the counter fix; phase stamps of one iteration (8 and 4 waves per tile)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s3; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for cfg in "8192 8 persistent" "8192 4 persistent" "10000 8 queue" "10000 4 queue"; do
  NPF16_LIB=tools/microbench/libs/plan_trace.so timeout 200 python tools/microbench/planning_phases.py $cfg 2>/dev/null
done | tee $out/phases.log
