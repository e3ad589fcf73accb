mkdir -p gpurun_out/s4
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/s4/pytest.log 2>&1
tail -5 gpurun_out/s4/pytest.log
python3 tools/microbench/ab_libs.py --rounds 2 > gpurun_out/s4/ab.log 2>&1
tail -8 gpurun_out/s4/ab.log
