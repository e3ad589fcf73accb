#!/bin/bash
# round 4, GPU session 9: where the guest schedule's ~75 us over (50 + block) iterations go — kernel trace of a PlanningEnv.step at n = 1e4
# (guests) and 8 192 (persistent; queue with one 50-iteration block = the coherent-access variant's own cost)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s9; mkdir -p $out
for cfg in "8192 20 0 persistent 8" "8192 20 0 queue 8 50" "8224 20 0 guests 8 1" "10000 20 0 guests 8 1"; do
  timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
for cfg in "10000 40 0 guests 8 1" "8192 40 0 persistent 8"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o p -- python tools/microbench/planning_profile.py $cfg > $out/prof_$tag.log 2>&1
  f=$(find $out/prof_$tag -name "p_kernel_stats.csv" | head -1); echo "== $cfg"; head -12 $f | cut -c1-200
done | tee $out/kernel_stats.log
