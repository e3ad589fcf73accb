#!/bin/bash
# round 3, GPU session 7: combat kernel with the state pinned before the Overload phase (pair 214 -> 170 VGPRs; three-wave build
# 184 -> 16 B of scratch): parity, two vs three waves per SIMD per grid size; de-phasing of the env kernel re-tuned on the new build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s7; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_combat_parity.py -x -q -m gpu > $out/pytest_combat.log 2>&1; tail -2 $out/pytest_combat.log
for pw in 2 3; do NPF16_PAIR_WAVES=$pw COMBAT_VARIANTS=pair timeout 400 python tools/microbench/combat_bench.py 12500 25000 50000 70000 100000 150000 200000 500000 > $out/combat_pw$pw.log 2>&1; done
COMBAT_VARIANTS=latency timeout 200 python tools/microbench/combat_bench.py 12500 25000 > $out/combat_lat.log 2>&1
NPF16_LIB=tools/microbench/libs/a_base.so COMBAT_VARIANTS=pair timeout 300 python tools/microbench/combat_bench.py 12500 50000 100000 500000 > $out/combat_r02.log 2>&1
grep -h "E=" $out/combat_*.log
timeout 900 python tools/microbench/ab_libs.py --rounds 2 --steps 100 e_head e_7x7000 e_7x11000 e_5x9000 e_9x7000 > $out/ab_stagger.log 2>&1; tail -7 $out/ab_stagger.log
