#!/bin/bash
# round 4, session 12: the window nets in the coherent kernels too (libs/winq.so, -DNP_PLAN_WIN_QUEUE=1) against the shipped library: guest schedule
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s12; mkdir -p $out
for rep in 1 2; do
for lib in "" tools/microbench/libs/winq.so; do
  for n in 9000 10000 12288; do
    echo -n "lib=${lib:-shipped} "; NPF16_LIB=$lib timeout 200 python tools/microbench/planning_profile.py $n 40 0 guests 8 1 2>/dev/null | grep "ms per"
  done
done; done | tee $out/planning_winq_ab.log
NPF16_LIB=tools/microbench/libs/winq.so timeout 900 python -m pytest tests/test_gpu_actor.py -m gpu -x -q 2>&1 | tail -3 | tee $out/tests_winq.txt
