#!/bin/bash
# round 3, GPU session 30: rocprofv3 kernel statistics of PlanningEnv.step (fused controller) at n = 8 192 and 1e4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s30; mkdir -p $out
for n in 8192 10000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$n -o p -- python tools/microbench/planning_profile.py $n 20 > $out/stats_$n.log 2>&1
  f=$(find $out/stats_$n -name "*kernel_stats.csv" | head -1); cut -c1-160 $f | head -8; grep "ms per" $out/stats_$n.log
  cp $f $out/planning_kernel_stats_n$n.csv
done
