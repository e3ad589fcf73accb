mkdir -p gpurun_out/s5
timeout 600 python3 -m pytest tests/test_gpu_step_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/s5/pytest.log 2>&1
tail -3 gpurun_out/s5/pytest.log
python3 tools/microbench/ab_libs.py --rounds 2 > gpurun_out/s5/ab.log 2>&1
tail -10 gpurun_out/s5/ab.log
