mkdir -p gpurun_out/s10
timeout 1200 python3 -m pytest tests -m gpu -x -q > gpurun_out/s10/pytest.log 2>&1
tail -5 gpurun_out/s10/pytest.log
python3 tools/microbench/small_n.py 256 4096 10000 16384 32768 > gpurun_out/s10/small.log 2>&1
cat gpurun_out/s10/small.log
