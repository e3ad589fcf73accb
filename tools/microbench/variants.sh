cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'steps/s', '%.3e'%d['value'])"; }
for rep in 1 2; do
for v in old new; do cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so; run "$v"; done
done
cp tools/microbench/libs/new.so neuralplane_amd/csrc/libneuralplane_hip.so
timeout 300 python -m pytest tests/test_gpu_step_parity.py -m gpu -x -q 2>&1 | tail -2
