#!/bin/bash
# round 3, GPU session 31: PlanningEnv's inner step on the three-wave pair build (de-phased) for n > 131 072 — parity tests, then the
# macro-step against the tree before (a_base)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s31; mkdir -p $out
NPF16_LIB=$PWD/tools/microbench/libs/h_inner3.so timeout 900 python -m pytest tests -x -q -m gpu -k "planning or Planning or inner" > $out/gputest.log 2>&1; grep -E "passed|failed" $out/gputest.log | tail -2
for lib in a_base h_inner3; do for n in 150000 262144; do
NPF16_LIB=$PWD/tools/microbench/libs/$lib.so timeout 300 python tools/microbench/planning_profile.py $n 4 2>/dev/null | grep "ms per" | sed "s/^/$lib /"
done; done | tee $out/planning.log
