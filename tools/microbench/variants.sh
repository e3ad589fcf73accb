cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'steps/s', '%.3e'%d['value'])"; }
cp neuralplane_amd/csrc/libneuralplane_hip.so /tmp/keep.so
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
for v in head xv; do cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so; run "$v tables" "--aero-1d-tables 1"; run "$v mlp" ""; done
done
cp /tmp/keep.so neuralplane_amd/csrc/libneuralplane_hip.so
