#!/bin/bash
# round 4, session 10: the multi-unit build (np_env_t*s*.hip) on the GPU: whole GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT; out=gpurun_out/${TAG:-r04f}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/gputest.txt 2>&1; echo "pytest rc $?" >> $out/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err < /dev/null
tail -3 $out/gputest.txt; tail -1 $out/smoke.txt; cut -c1-300 $out/bench_driver_cmd.json
