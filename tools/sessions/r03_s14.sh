#!/bin/bash
# round 3, GPU session 14: pair variant at two waves per SIMD with the observation noise generated behind the state loads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s14; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py -x -q -m gpu -k "pair or mid_sizes or full_size" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for r in 1 2; do for lib in i_notrig j_head; do NPF16_LIB=tools/microbench/libs/$lib.so timeout 400 python tools/microbench/mid_n.py --variants auto --out $out/mid_${lib}_$r.json 100000 114688 131072 229376 262144 327680 > $out/mid_${lib}_$r.log 2>&1; grep "N=" $out/mid_${lib}_$r.log | sed "s/^/$lib /"; done; done
