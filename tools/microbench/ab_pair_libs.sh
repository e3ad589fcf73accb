#!/bin/bash
# pair-variant builds against each other: bash tools/microbench/ab_pair_libs.sh libA libB ...   (NS = batch sizes)
cd $GRAFT_REPO_ROOT
run() { NPF16_KERNEL=pair timeout 200 python bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --aircraft $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'N=$2', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'steps/s', '%.3e'%d['value'])"; }
cp neuralplane_amd/csrc/libneuralplane_hip.so /tmp/keep.so
for rep in 1 2; do for v in "$@"; do cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so; for n in ${NS:-1000000}; do run $v $n; done; done; done
cp /tmp/keep.so neuralplane_amd/csrc/libneuralplane_hip.so
