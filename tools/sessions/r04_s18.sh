#!/bin/bash
# round 4, GPU session 18: pipelined schedule with the 22 moment-side alpha/beta-only nets moved from an inner step's front to its back
# (front: the six el-dependent nets only; back: all 36 + Cx, Cz on waves 4..7 during the next controller call) — parity, timings, phase stamps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s18; mkdir -p $out
for k in "200-persistent-8" "10037-guests" "95-queue-8-1"; do
  timeout 120 python -m pytest tests/test_gpu_actor.py -x -q -m gpu -k "persistent_kernel and $k" > $out/t_$k.log 2>&1; echo "$k rc=$?"; grep -E "passed|failed" $out/t_$k.log | tail -1; grep -B2 -A12 "Error\|assert" $out/t_$k.log | head -30
done
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for cfg in "3000 20 0 auto" "8192 20 0 auto" "9000 20 0 auto" "10000 20 0 auto" "10000 20 0 queue 8 5" "12288 20 0 auto"; do
  timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
for cfg in "8192 8 persistent"; do
  NPF16_LIB=tools/microbench/libs/plan_trace.so timeout 200 python tools/microbench/planning_phases.py $cfg 2>/dev/null
done | tee $out/phases.log
