#!/bin/bash
# round 3, GPU session 33: np_planning_inner_loop (the 50 iterations of PlanningEnv.step enqueued by one call, two row groups for
# 8 192 < n <= 16 384) — parity with the launch-by-launch path, then PlanningEnv.step through the public surface
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s33; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_actor.py tests/test_gpu_step_parity.py -x -q -m gpu -k "planning or Planning or actor or inner" > $out/gputest.log 2>&1; grep -E "passed|failed|Error" $out/gputest.log | tail -3; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for n in 3000 8192 10000 12288 16384; do timeout 200 python tools/microbench/planning_profile.py $n 20 2>/dev/null | grep "ms per"; done | tee $out/planning.log
