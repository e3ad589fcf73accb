#!/bin/bash
# round 4, session 11: the static schedule with the moment-side nets in the controller call's windows (NP_PLAN_WIN=1, shipped) against the
# round-4 front (libs/win0.so), same session; then the controller / planning / rollout tests on the shipped library
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s11; mkdir -p $out
for rep in 1 2; do
for lib in "" tools/microbench/libs/win0.so; do
  for n in 4096 8192; do
    echo -n "lib=${lib:-shipped} "; NPF16_LIB=$lib timeout 200 python tools/microbench/planning_profile.py $n 40 0 persistent 8 2>/dev/null | grep "ms per"
  done
done; done | tee $out/planning_win_ab.log
timeout 900 python -m pytest tests/test_gpu_actor.py tests/test_gpu_rollout.py -m gpu -x -q 2>&1 | tail -3 | tee $out/tests.txt
