#!/bin/bash
# round 3, GPU session 20: full GPU suite + planning / mid-N checks of the tree with the 32-row controller tiles and the one-generation offset
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s20; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
timeout 300 python tools/microbench/mid_n.py --variants auto --out $out/mid.json 114688 122880 131072 > $out/mid.log 2>&1; grep "N=" $out/mid.log
timeout 600 python - > $out/planning.log 2>&1 <<'PY'
import json, os, sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0'); g = torch.Generator(device=dev); g.manual_seed(0)
for n, k in ((1024, 20), (4096, 20), (8192, 20), (10000, 20), (16384, 20), (32768, 10), (262144, 4)):
    r = bench.planning_mode(dev, g, n, k)
    print(n, f"{r['value']:.3f} ms  env {r['env_kernels_ms_per_macro_step']:.3f} ms  frac {r['roofline']['frac']:.3f}", flush=True)
PY
grep -v Warn $out/planning.log | tail -8
