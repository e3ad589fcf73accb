#!/bin/bash
# round 4, GPU session 21: dual workgroups (np_planning_loop.mode = dual: two 32-row tiles per eight-wave workgroup, controller calls in
# lock-step on waves 0..3 / 4..7, one 64-lane FDM step for both) — parity (own timeouts), timings against the other schedules
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s21; mkdir -p $out
for k in "200-dual" "33-dual" "1-dual" "95-dual" "10037-dual" "16384-dual" "20011-dual"; do
  timeout 120 python -m pytest tests/test_gpu_actor.py -x -q -m gpu -k "persistent_kernel and $k" > $out/t_$k.log 2>&1; echo "$k rc=$?"; grep -E "passed|failed" $out/t_$k.log | tail -1; grep -B2 -A14 "Error\|assert" $out/t_$k.log | head -34
done
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -60
for cfg in "8192 20 0 dual 8" "8192 20 0 auto" "10000 20 0 dual 8" "10000 20 0 auto" "12288 20 0 dual 8" "12288 20 0 auto" "14000 20 0 dual 8" "14000 20 0 launches" "16384 20 0 dual 8" "16384 20 0 launches" "20000 10 0 dual 8" "20000 10 0 launches" "24576 10 0 dual 8" "24576 10 0 launches" "32768 10 0 dual 8" "32768 10 0 launches" "49152 6 0 dual 8" "49152 6 0 launches" "65536 6 0 dual 8" "65536 6 0 launches" "131072 4 0 dual 8" "131072 4 0 launches" "262144 4 0 dual 8" "262144 4 0 launches"; do
  timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
