#!/bin/bash
# round 4, GPU session 32: SingleCombat dual8 variant (np_combat_lat.hip) — parity first, then the A/B against the four-wave latency kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s32; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_combat_parity.py -x -q -m gpu > $out/combat_tests.log 2>&1; echo "combat tests rc=$?"; tail -3 $out/combat_tests.log
for E in 512 4096 8192 12500 16384 20000; do
  COMBAT_VARIANTS=auto,latency,pair timeout 200 python tools/microbench/combat_bench.py $E 2>&1 | grep "E=" | sed 's/^/dual8on  /' >> $out/combat_dual8_ab.log
  NPF16_COMBAT_DUAL8=0 COMBAT_VARIANTS=auto timeout 200 python tools/microbench/combat_bench.py $E 2>&1 | grep "E=" | sed 's/^/dual8off /' >> $out/combat_dual8_ab.log
done
cat $out/combat_dual8_ab.log
