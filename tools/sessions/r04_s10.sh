#!/bin/bash
# round 4, GPU session 10: coherent transfer of the recurrent state through the LDS stage (8-byte pieces, consecutive across lanes), queue words
# cleared once instead of per call — parity of the coherent schedules, then the timings
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s10; mkdir -p $out
for k in "95-queue-8-1" "10037-guests" "20011-guests"; do
  timeout 120 python -m pytest tests/test_gpu_actor.py -x -q -m gpu -k "persistent_kernel and $k" > $out/t_$k.log 2>&1; echo "$k rc=$?"; grep -E "passed|failed" $out/t_$k.log | tail -1; grep -B2 -A12 "Error\|assert" $out/t_$k.log | head -30
done
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for cfg in "8192 20 0 auto" "8192 20 0 queue 8 50" "8224 20 0 guests 8 1" "9000 20 0 guests 8 1" "10000 20 0 guests 8 1" "10000 20 0 queue 8 5" "10000 20 0 queue 8 2" "11000 20 0 guests 8 1" "12288 20 0 guests 8 1" "12288 20 0 queue 8 5" "16384 20 0 queue 4 5" "16384 20 0 guests 4 1" "20000 20 0 guests 4 1" "20000 20 0 queue 4 5"; do
  timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
