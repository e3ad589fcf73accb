#!/bin/bash
# round 4, GPU session 4: wave priority outside the controller's MFMA chains (s_setprio) — A/B of the persistent kernel where two tiles share a CU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s4; mkdir -p $out
for lib in plan_base plan_prio1 plan_prio3; do
  for cfg in "8192 20 0 persistent 8" "10000 20 0 persistent 4" "16384 20 0 persistent 4" "10000 20 0 queue 4" "10000 20 0 queue 8"; do
    echo -n "$lib: "; NPF16_LIB=tools/microbench/libs/$lib.so timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"
  done
done | tee $out/prio_ab.log
