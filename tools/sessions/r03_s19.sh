#!/bin/bash
# round 3, GPU session 19: recurrent-state store deferred to the end of the 32-row controller kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s19; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q > $out/pytest_actor.log 2>&1; tail -3 $out/pytest_actor.log
NPF16_LIB=tools/microbench/libs/act_trace.so timeout 300 python tools/microbench/actor_phases.py 4096 > $out/phases.log 2>&1; grep -v Warn $out/phases.log
timeout 600 python tools/microbench/actor_bench.py 64 1024 4096 8192 10000 16384 32768 65536 > $out/actor_bench.log 2>&1; grep "n=" $out/actor_bench.log
