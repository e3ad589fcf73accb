#!/bin/bash
# round 3, GPU session 18: 32-row controller kernel with the 3-deep weight ring, LDS-staged head; start offset for the second tile of a CU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s18; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q > $out/pytest_actor.log 2>&1; tail -3 $out/pytest_actor.log
NPF16_LIB=tools/microbench/libs/act_trace.so timeout 300 python tools/microbench/actor_phases.py 4096 10000 > $out/phases.log 2>&1; grep -v Warn $out/phases.log
for lib in "" act_d2500 act_d5000 act_d9000; do
  if [ -n "$lib" ]; then export NPF16_LIB=tools/microbench/libs/$lib.so; fi
  timeout 600 python tools/microbench/actor_bench.py 1024 8192 10000 12288 16384 32768 > $out/actor_bench_$lib.log 2>&1; echo "== ${lib:-head}"; grep "n=" $out/actor_bench_$lib.log
done
