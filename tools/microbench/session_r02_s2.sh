mkdir -p gpurun_out/s2
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1
tail -15 gpurun_out/s2/pytest.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s2/bench_driver.json 2> gpurun_out/s2/bench_driver.err
python3 tools/wg_timeline.py --n 1000000 --out gpurun_out/s2/timeline_1e6.json --raw gpurun_out/s2/trace_1e6.npy > /dev/null 2> gpurun_out/s2/timeline.err
python3 tools/wg_timeline.py --n 10000000 --prelude 40 --out gpurun_out/s2/timeline_1e7.json > /dev/null 2>> gpurun_out/s2/timeline.err
python3 bench.py --gpus 1 --task combat --steps 50 --warmup 5 --headline-only > gpurun_out/s2/bench_combat.json 2> gpurun_out/s2/bench_combat.err
