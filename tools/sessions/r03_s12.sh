#!/bin/bash
# round 3, GPU session 12: latency variants with the noise share at the top of the kernel; table lookups through 32-bit offsets;
# 128-VGPR build of the latency family (four waves per SIMD) now that it needs 132
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s12; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
for lib in g_head h_head; do NPF16_LIB=tools/microbench/libs/$lib.so timeout 400 python tools/microbench/mid_n.py --variants auto --out $out/mid_$lib.json 256 3000 10000 30000 49152 65536 81920 98304 > $out/mid_$lib.log 2>&1; grep "N=" $out/mid_$lib.log | sed "s/^/$lib /"; done
NPF16_LIB=tools/microbench/libs/h_mw4.so timeout 400 python tools/microbench/mid_n.py --variants latency,latency2 --out $out/mid_h_mw4.json 49152 65536 81920 98304 131072 > $out/mid_h_mw4.log 2>&1; grep "N=" $out/mid_h_mw4.log | sed "s/^/mw4 /"
for lib in g_head h_head; do
  NPF16_LIB=tools/microbench/libs/$lib.so timeout 300 python bench.py --headline-only --steps 100 --warmup 5 --aero-1d-tables 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib tables', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['kernel_median_ms'])"
done
