cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in cw2 cw3; do cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so; echo "== $v"; timeout 200 python tools/microbench/combat_bench.py 2>&1 | grep "E="; done
done
