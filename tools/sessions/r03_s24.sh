#!/bin/bash
# round 3, GPU session 24: SingleCombat tail launch (the remainder of a grid beyond whole generations -> latency variant on a second
# stream): parity suite, then kernel time with the tail launch off (NPF16_COMBAT_TAIL=0) / on around the generation boundaries
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s24; mkdir -p $out
(time timeout 900 python -m pytest tests/test_gpu_combat_parity.py -x -q -m gpu) > $out/gputest_combat.log 2>&1; grep -E "passed|failed" $out/gputest_combat.log
sizes="98304 99000 100000 102000 105000 110000 150000 196608 200000 205000 300000 500000"
for tail in 0 192 384; do NPF16_COMBAT_TAIL=$tail COMBAT_VARIANTS=auto timeout 300 python tools/microbench/combat_bench.py $sizes 2>/dev/null | grep "E=" | sed "s/^/tail$tail /"; done > $out/combat.log
cat $out/combat.log
