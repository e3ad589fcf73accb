#!/bin/bash
# round 3, GPU session 2: the two-waves-per-tile latency variant — parity, then the mid-size sweep against the other variants;
# a 128-VGPR build of the latency family (four waves per SIMD resident) beside it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s2; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "latency2" > $out/pytest_lat2.log 2>&1
tail -3 $out/pytest_lat2.log
timeout 600 python tools/microbench/mid_n.py --variants latency,latency2,pair --out $out/mid_lat2.json 30000 40000 49152 57344 65536 73728 81920 90112 98304 100000 114688 131072 > $out/mid_lat2.log 2>&1
NPF16_LIB=tools/microbench/libs/mw4.so timeout 600 python tools/microbench/mid_n.py --variants latency,latency2 --out $out/mid_mw4.json 49152 65536 81920 98304 100000 131072 > $out/mid_mw4.log 2>&1
for t in control tracking; do timeout 300 python tools/microbench/mid_n.py --task $t --variants latency,latency2,pair --out $out/mid_lat2_$t.json 65536 98304 > $out/mid_lat2_$t.log 2>&1; done
ls -la $out
