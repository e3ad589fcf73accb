cd $GRAFT_REPO_ROOT
cp neuralplane_amd/csrc/libneuralplane_hip.so /tmp/keep.so
for v in ${LIBS:-head exp1 exp4 exp5}; do cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so; echo "== $v"; timeout 120 python tools/microbench/small_n.py 256 10000 2>&1 | grep "N="; done
cp /tmp/keep.so neuralplane_amd/csrc/libneuralplane_hip.so
