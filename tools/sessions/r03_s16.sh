#!/bin/bash
# round 3, GPU session 16: the controller on 32-row tiles (v_mfma_f32_16x16x1_4b_f32): exactness of the instruction, parity, per-call time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s16; mkdir -p $out
timeout 120 tools/microbench/mfma16_exact > $out/mfma16_exact.log 2>&1; tail -12 $out/mfma16_exact.log
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q > $out/pytest_actor.log 2>&1; tail -5 $out/pytest_actor.log
timeout 600 python tools/microbench/actor_bench.py > $out/actor_bench.log 2>&1; cat $out/actor_bench.log
timeout 600 python - > $out/planning.log 2>&1 <<'PY'
import json, os, sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0'); g = torch.Generator(device=dev); g.manual_seed(0)
for tile in ('', '64'):
    if tile: os.environ['NP_ACTOR_TILE'] = tile
    else: os.environ.pop('NP_ACTOR_TILE', None)
    for n, k in ((4096, 20), (8192, 20), (10000, 20), (16384, 20), (32768, 10)):
        r = bench.planning_mode(dev, g, n, k)
        print(tile or 'auto', n, f"{r['value']:.3f} ms  env {r['env_kernels_ms_per_macro_step']:.3f} ms  frac {r['roofline']['frac']:.3f}", flush=True)
PY
cat $out/planning.log | grep -v Warn | tail -12
