#!/bin/bash
# pair kernel variant against the default throughput variant, same library, one session: NS="1000000 3000000" bash tools/microbench/ab_pair.sh
cd $GRAFT_REPO_ROOT
run() { NPF16_KERNEL=$1 timeout 200 python bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --aircraft $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${1:-default}', 'N=$2', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'steps/s', '%.3e'%d['value'])"; }
for rep in 1 2; do for n in ${NS:-1000000}; do run "" $n; run pair $n; done; done
