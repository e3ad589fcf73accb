#!/bin/bash
# round 3, GPU session 6: full GPU suite on the tree (pins for the pair variant only, ADVICE fixes), driver command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s6; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench.err < /dev/null
tail -c 600 $out/bench_driver_cmd.json
