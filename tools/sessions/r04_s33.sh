#!/bin/bash
# round 4, GPU session 33: SingleCombat dual family (np_combat_lat.hip: dual8 / dual4) — parity, then the variants side by side per batch size
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s33; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_combat_parity.py -x -q -m gpu > $out/combat_tests.log 2>&1; echo "combat tests rc=$?"; tail -3 $out/combat_tests.log
for E in 8192 10000 12500 16384 20000 25000 32768 36000 50000; do
  COMBAT_VARIANTS=auto,latency,dual8,dual4,pair timeout 300 python tools/microbench/combat_bench.py $E 2>&1 | grep "E=" >> $out/combat_dual_ab.log
done
cat $out/combat_dual_ab.log | cut -c1-80
for E in 12500 25000; do
timeout 300 python bench.py --task combat --engagements $E --steps 200 --warmup 20 > $out/bench_combat_e$E.json 2>> $out/bench.err < /dev/null
done
python - <<'PY'
import json
for E in (12500, 25000):
    d = json.loads(open(f'gpurun_out/r04_s33/bench_combat_e{E}.json').read().strip().splitlines()[-1])
    print(E, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_ms'])
PY
