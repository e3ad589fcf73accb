#!/bin/bash
# round 4, GPU session 5: the pipelined schedule of the persistent kernel (an inner step's Overload evaluation + terminations on waves 4..7
# during the next controller call) — parity first (a barrier mismatch would hang: own timeouts), then timing and phase stamps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s5; mkdir -p $out
export NPF16_LIB=tools/microbench/libs/plan_pipe.so
for k in "200-persistent-8" "33-persistent-8" "8192-persistent-8"; do
  timeout 120 python -m pytest tests/test_gpu_actor.py -x -q -m gpu -k "persistent_kernel and $k" > $out/t_$k.log 2>&1; echo "$k rc=$?"; grep -E "passed|failed" $out/t_$k.log | tail -1; grep -B2 -A12 "Error\|assert" $out/t_$k.log | head -30
done
timeout 600 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for cfg in "3000 20 0 persistent 8" "8192 20 0 persistent 8" "8192 20 0 launches 0" "3000 20 0 launches 0"; do timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
