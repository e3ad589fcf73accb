#!/bin/bash
# round 4, session 13: soak of the persistent kernel's schedules against the launch-by-launch path (400 macro-steps per case)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s13; mkdir -p $out
timeout 1500 python tools/microbench/planning_soak_modes.py ${SOAK_STEPS:-400} 2>&1 | grep -v Warning | tee $out/planning_soak_modes.log
