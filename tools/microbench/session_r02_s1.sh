mkdir -p gpurun_out/s1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/bench_driver.json 2> gpurun_out/s1/bench_driver.err
python3 tools/microbench/cold_start.py --steps 120 > gpurun_out/s1/cold_none.json 2>&1
python3 tools/microbench/cold_start.py --steps 120 --prelude spin > gpurun_out/s1/cold_spin.json 2>&1
python3 tools/microbench/cold_start.py --steps 120 --prelude sleep > gpurun_out/s1/cold_sleep.json 2>&1
python3 tools/microbench/cold_start.py --steps 120 --trace > gpurun_out/s1/cold_trace.json 2>&1
python3 tools/microbench/cold_start.py --steps 60 --n 10000000 --trace > gpurun_out/s1/cold_trace_1e7.json 2>&1
python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/s1/bench_200.json 2> gpurun_out/s1/bench_200.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s1/bench_driver2.json 2> gpurun_out/s1/bench_driver2.err
timeout 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1
tail -3 gpurun_out/s1/pytest.log
