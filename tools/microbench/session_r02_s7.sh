mkdir -p gpurun_out/s7
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/s7/pytest.log 2>&1
tail -5 gpurun_out/s7/pytest.log
python3 tools/microbench/combat_bench.py 12500 50000 100000 500000 > gpurun_out/s7/combat.log 2>&1
cat gpurun_out/s7/combat.log
python3 tools/microbench/ab_libs.py --rounds 2 > gpurun_out/s7/ab.log 2>&1
tail -4 gpurun_out/s7/ab.log
