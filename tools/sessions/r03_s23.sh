#!/bin/bash
# round 3, GPU session 23: full GPU suite on the tree with the three-wave pair build de-phased from 1 025 workgroups; SingleCombat
# kernel time against the number of engagements around the one-generation boundary (98 304 engagements = 1 536 workgroups)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s23; mkdir -p $out
(time timeout 1100 python -m pytest tests -m gpu -x -q) > $out/gputest.log 2>&1; tail -3 $out/gputest.log
sizes="25000 50000 65536 80000 90000 98304 100000 110000 131072 150000 196608 250000 500000"
COMBAT_VARIANTS=auto timeout 300 python tools/microbench/combat_bench.py $sizes 2>/dev/null | grep "E=" | sed "s/^/auto  /" > $out/combat.log
for pw in 2 3; do NPF16_PAIR_WAVES=$pw COMBAT_VARIANTS=pair timeout 300 python tools/microbench/combat_bench.py $sizes 2>/dev/null | grep "E=" | sed "s/^/pair$pw /" >> $out/combat.log; done
cat $out/combat.log
