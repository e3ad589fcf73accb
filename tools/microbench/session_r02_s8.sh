mkdir -p gpurun_out/s8
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/s8/pytest.log 2>&1
tail -5 gpurun_out/s8/pytest.log
NPF16_LIB=$PWD/tools/microbench/libs/h_cur.so python3 tools/microbench/small_n.py 256 4096 16384 65536 > gpurun_out/s8/small_before.log 2>&1
python3 tools/microbench/small_n.py 256 4096 16384 32768 65536 98304 > gpurun_out/s8/small_after.log 2>&1
cat gpurun_out/s8/small_before.log gpurun_out/s8/small_after.log
