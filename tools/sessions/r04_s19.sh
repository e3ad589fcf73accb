#!/bin/bash
# round 4, GPU session 19: the round's evidence set on the library as committed (tools/profile_round.sh r04c) + the persistent planning
# kernel's own kernel stats at n = 8 192 / 1e4
bash tools/profile_round.sh r04c
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c
for n in 8192 10000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_planning_$n -o p -- python tools/microbench/planning_profile.py $n 40 0 auto > $out/stats_planning_$n.log 2>&1 < /dev/null
  f=$(find $out/stats_planning_$n -name "p_kernel_stats.csv" | head -1); cp $f $out/planning_kernel_stats_n$n.csv
done
NPF16_LIB=tools/microbench/libs/plan_trace.so timeout 200 python tools/microbench/planning_phases.py 8192 8 persistent > $out/planning_phases.log 2>/dev/null
ls $out | head -50
