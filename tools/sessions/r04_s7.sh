#!/bin/bash
# round 4, GPU session 7 (after the container was re-created: the earlier sessions' logs are gone): persistent kernel parity, the
# controller / planning tests, timings per mode and size, phase stamps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s7; mkdir -p $out
for k in "200-persistent-8" "95-queue-8-1"; do
  timeout 180 python -m pytest tests/test_gpu_actor.py -x -q -m gpu -k "persistent_kernel and $k" > $out/t_$k.log 2>&1; echo "$k rc=$?"; grep -E "passed|failed" $out/t_$k.log | tail -1; grep -B2 -A12 "Error\|assert" $out/t_$k.log | head -30
done
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for n in 3000 8192 10000 16384; do
  for cfg in "0 launches 0" "0 persistent 4" "0 persistent 8" "0 queue 4 5" "0 queue 8 5" "0 queue 8 10"; do timeout 200 python tools/microbench/planning_profile.py $n 20 $cfg 2>/dev/null | grep "ms per"; done
done | tee $out/planning.log
for cfg in "8192 8 persistent" "8192 4 persistent" "10000 8 queue"; do
  NPF16_LIB=tools/microbench/libs/plan_trace.so timeout 200 python tools/microbench/planning_phases.py $cfg 2>/dev/null
done | tee $out/phases.log
