#!/bin/bash
# A/B of SingleCombat kernel builds inside ONE gpurun session:  gpurun -- 'bash tools/microbench/combat_variants.sh a b'  (libs/<name>.so)
cd $GRAFT_REPO_ROOT
cp neuralplane_amd/csrc/libneuralplane_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so
    echo "== $v"; timeout 200 python tools/microbench/combat_bench.py ${ES:-12500 100000} 2>&1 | grep "E="
  done
done
cp /tmp/keep.so neuralplane_amd/csrc/libneuralplane_hip.so
