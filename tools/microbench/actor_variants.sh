#!/bin/bash
# A/B timing of the fused-actor kernel across builds inside ONE gpurun session (see variants.sh):
#     gpurun -- 'bash tools/microbench/actor_variants.sh act0 act1 ...'      (tools/microbench/libs/<name>.so, git-ignored)
# NPACT_EXP bit 1 = all waves stream the same weight slice, 2 = every feature re-reads one 64 B chunk (timing only; wrong results).
cd $GRAFT_REPO_ROOT
cp neuralplane_amd/csrc/libneuralplane_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so
    echo "== $v"; timeout 120 python tools/microbench/actor_bench.py 2>&1 | grep -E "n=(64|16384|262144):"
  done
done
cp /tmp/keep.so neuralplane_amd/csrc/libneuralplane_hip.so
