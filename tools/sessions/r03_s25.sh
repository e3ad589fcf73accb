#!/bin/bash
# round 3, GPU session 25: first-generation de-phasing in the SingleCombat pair kernel (groups x cycles), one session
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s25; mkdir -p $out
sizes="98304 100000 150000 196608 300000 500000"
for lib in a_base e_9x7000 e_9x20000 e_5x40000 e_3x60000; do NPF16_LIB=tools/microbench/libs/$lib.so COMBAT_VARIANTS=auto timeout 300 python tools/microbench/combat_bench.py $sizes 2>/dev/null | grep "E=" | sed "s/^/$lib /"; done > $out/combat.log
cut -c1-75 $out/combat.log
