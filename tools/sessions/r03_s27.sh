#!/bin/bash
# round 3, GPU session 27: latency8 with the atmosphere power on a wave of its own (g_spread) against the tree before (a_base), small N;
# in-kernel phase stamps of the eight-wave variant (lat_trace = g_spread + -DNPF16_LAT_TRACE)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s27; mkdir -p $out
for rep in 1 2; do for lib in a_base g_spread; do
  NPF16_LIB=tools/microbench/libs/$lib.so timeout 300 python tools/microbench/mid_n.py --variants auto --steps 2000 256 3000 10000 16384 2>/dev/null | grep "N=" | sed "s/^/$lib /"
done; done > $out/table.log 2>&1
cut -c1-100 $out/table.log
NPF16_LIB=$PWD/tools/microbench/libs/lat_trace.so timeout 120 python tools/microbench/lat_trace.py 256 latency8 > $out/lat_trace_256.log 2>&1; grep "tile 0" $out/lat_trace_256.log | cut -c1-220
NPF16_LIB=$PWD/tools/microbench/libs/lat_trace.so timeout 120 python tools/microbench/lat_trace.py 10000 latency8 > $out/lat_trace_10000.log 2>&1; grep "tile 1" $out/lat_trace_10000.log | cut -c1-220
