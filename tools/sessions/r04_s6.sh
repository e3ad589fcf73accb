#!/bin/bash
# round 4, GPU session 6: (tile, block) queue parity + timings at n = 1e4 per block size; phase stamps of the pipelined schedule
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s6; mkdir -p $out
timeout 120 python -m pytest tests/test_gpu_actor.py -x -q -m gpu -k "persistent_kernel and 95-queue" > $out/t_q.log 2>&1; echo "queue small rc=$?"; grep -E "passed|failed" $out/t_q.log | tail -1; grep -B2 -A12 "Error\|assert" $out/t_q.log | head -30
timeout 600 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for cfg in "10000 20 0 queue 8 1" "10000 20 0 queue 8 2" "10000 20 0 queue 8 5" "10000 20 0 queue 8 10" "10000 20 0 queue 8 25" "10000 20 0 launches 0" "9000 20 0 queue 8 5" "12288 20 0 queue 8 5" "12288 20 0 launches 0" "16384 20 0 queue 8 5" "8192 20 0 persistent 8" "8192 20 0 queue 8 50"; do timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
for cfg in "8192 8 persistent"; do
  NPF16_LIB=tools/microbench/libs/plan_trace.so timeout 200 python tools/microbench/planning_phases.py $cfg 2>/dev/null
done | tee $out/phases.log
