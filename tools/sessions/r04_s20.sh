#!/bin/bash
# round 4, GPU session 20: LayerNorm in the MFMA accumulator layout (quad-tree summation order in the oracle and both controller kernels,
# v_permlane32_swap / v_permlane16_swap, three barriers instead of four, the recurrent state in the accumulator layout) — parity of the
# controller kernels against the oracle, the planning paths against each other, then timings and phase stamps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s20; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q -m gpu > $out/gputest.log 2>&1; echo "actor tests rc=$?"; grep -E "passed|failed" $out/gputest.log | tail -2; grep -B5 -A30 "Error\|FAILED" $out/gputest.log | head -80
for cfg in "3000 20 0 auto" "8192 20 0 auto" "8192 20 0 persistent 4" "8192 20 0 launches" "10000 20 0 auto" "10000 20 0 launches" "12288 20 0 auto" "16384 20 0 auto" "32768 10 0 auto" "262144 4 0 auto"; do
  timeout 200 python tools/microbench/planning_profile.py $cfg 2>/dev/null | grep "ms per"; done | tee $out/planning.log
NPF16_LIB=tools/microbench/libs/plan_trace.so timeout 200 python tools/microbench/planning_phases.py 8192 8 persistent 2>/dev/null | tee $out/phases.log
timeout 300 python tools/microbench/actor_bench.py 2>/dev/null | tail -12 | tee $out/actor.log
