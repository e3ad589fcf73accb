#!/bin/bash
# round 3, GPU session 26: de-phasing of the two-waves-per-tile latency variant on grids that fill the chip (65 536 < N <= 98 304), and
# that variant against the pair variant just above its range
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s26; mkdir -p $out
sizes="69632 73728 81920 90112 98304"
for lib in a_base f_3x4000 f_3x8000 f_5x3000 f_9x2000; do
  NPF16_LIB=tools/microbench/libs/$lib.so timeout 300 python tools/microbench/mid_n.py --variants latency2 --steps 300 $sizes 2>/dev/null | grep "N=" | sed "s/^/$lib /"
done > $out/table.log 2>&1
cut -c1-90 $out/table.log
