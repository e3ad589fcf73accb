#!/bin/bash
# round 4, GPU session 8: the static guest schedule (np_planning_loop.mode = guests), mask / controller targets in the tile context,
# three-launch prelude — parity (own timeouts: a schedule bug would hang), timings per mode and size, phase stamps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s8; mkdir -p $out
for k in "200-guests" "10037-guests" "9001-guests" "12288-guests" "16000-guests" "20011-guests" "10037-auto" "8192-auto"; do
  timeout 120 python -m pytest tests/test_gpu_actor.py -x -q -m gpu -k "persistent_kernel and $k" > $out/t_$k.log 2>&1; echo "$k rc=$?"; grep -E "passed|failed" $out/t_$k.log | tail -1; grep -B2 -A12 "Error\|assert" $out/t_$k.log | head -30
done
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for cfg in "3000 20 0 auto" "8192 20 0 auto" "8192 20 0 persistent 8" "8192 20 0 persistent 4" "9000 20 0 guests 8 1" "9000 20 0 launches" "10000 20 0 guests 8 1" "10000 20 0 guests 8 2" "10000 20 0 guests 8 3" "10000 20 0 queue 8 5" "10000 20 0 auto" \
    "11000 20 0 guests 8 1" "11000 20 0 launches" "12288 20 0 guests 8 1" "12288 20 0 guests 8 2" "12288 20 0 launches" "14000 20 0 guests 8 1" "14000 20 0 launches" "16384 20 0 guests 8 1" "16384 20 0 guests 4 1" "20000 20 0 guests 4 1" "20000 20 0 launches" "24576 20 0 guests 4 1" "24576 20 0 launches"; do
  timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
for cfg in "8192 8 persistent" "10000 8 guests"; do
  NPF16_LIB=tools/microbench/libs/plan_trace.so timeout 200 python tools/microbench/planning_phases.py $cfg 2>/dev/null
done | tee $out/phases.log
