#!/bin/bash
# round 4, session 14: two (then three) processes run the guest / queue schedules on the one GPU at the same time
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s14; mkdir -p $out
( timeout 240 python tools/microbench/planning_two_procs.py 10000 200 2; echo "rc $?"; timeout 240 python tools/microbench/planning_two_procs.py 9000 100 3; echo "rc $?" ) 2>&1 | grep -v "Warning\|amdgpu.ids\|warn" | tee $out/planning_two_procs.log
