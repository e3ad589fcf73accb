cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'steps/s', '%.3e'%d['value'])"; }
timeout 600 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_combat_parity.py -m gpu -x -q 2>&1 | tail -3
cp neuralplane_amd/csrc/libneuralplane_hip.so /tmp/keep.so
for v in phase0 phase1 phase0 phase1; do cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so; run "$v"; echo "== $v combat"; timeout 100 python tools/microbench/combat_bench.py 2>&1 | grep "E=100000"; done
cp /tmp/keep.so neuralplane_amd/csrc/libneuralplane_hip.so
